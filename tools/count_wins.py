#!/usr/bin/env python3
"""Useful bytes of a pipelined launch: how many pixels win the max-weight select, per level, in the steady state
bench.py times (cfg-A, keyframes W .. W+K).  Diagnostic build: PF_STAMP=1 selects the stamped instantiation of the block-form kernel, whose
stage D counts pixels seen / won.   usage: PF_STAMP=1 python tools/count_wins.py [--int16] [--frames 100] [--warm 20]"""
import argparse, ctypes as C, importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
import numpy as np
import torch
ap = argparse.ArgumentParser(); ap.add_argument("--int16", action="store_true"); ap.add_argument("--frames", type=int, default=100); ap.add_argument("--warm", type=int, default=20)
a = ap.parse_args()
assert os.environ.get("PF_STAMP"), "run with PF_STAMP=1"
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = bench.CAM
poses = wl.serpentine(cam, 100.0, a.frames + a.warm)
m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=0 if a.int16 else 1)
assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
fr = [torch.randint(0, 256, (cam[1], cam[0], 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
L = pf.lib(); L.pf_debug_select_counts.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(18, np.uint64)
for k in range(a.warm):
    m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k])
m.sync(); L.pf_debug_select_counts(buf.ctypes.data, 1)
for k in range(a.warm, a.warm + a.frames):
    m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k])
m.sync(); L.pf_debug_select_counts(buf.ctypes.data, 1)
es = 6 if a.int16 else 12
tot_seen = tot_won = 0
print("| level | pixels seen per keyframe | won | share |"); print("|---|---|---|---|")
for lv in range(9):
    seen, won = int(buf[2 * lv]), int(buf[2 * lv + 1])
    if not seen:
        continue
    tot_seen += seen; tot_won += won
    print("| %d | %.3f M | %.3f M | %.1f %% |" % (lv, seen / a.frames / 1e6, won / a.frames / 1e6, 100.0 * won / seen))
src = cam[0] * cam[1] * 3
useful = src + tot_seen / a.frames * 4 + tot_won / a.frames * (es + 4)
print("per keyframe: %.2f M pixel-levels seen, %.2f M won (%.1f %%)" % (tot_seen / a.frames / 1e6, tot_won / a.frames / 1e6, 100.0 * tot_won / max(tot_seen, 1)))
print("useful bytes per keyframe = source %.1f MB + stored weights read %.1f MB + winning pixels written %.1f MB = %.1f MB" %
      (src / 1e6, tot_seen / a.frames * 4 / 1e6, tot_won / a.frames * (es + 4) / 1e6, useful / 1e6))
