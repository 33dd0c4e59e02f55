#!/usr/bin/env python3
"""Per-role timing of the rolling-strip level kernel (diagnostic build: PF_STAMP=1 PF_STRIPS=1 selects a stamped instantiation
of k_strips; the product kernel carries no stamps): for every wave of a workgroup, the cycles it waited at the period barriers
and its lifetime -> which role sets the pace of a period.
usage: PF_STAMP=1 PF_STRIPS=1 python tools/strip_roles.py [--int16] [--frames N]"""
import argparse, ctypes as C, importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
import numpy as np
ap = argparse.ArgumentParser(); ap.add_argument("--int16", action="store_true"); ap.add_argument("--frames", type=int, default=60)
a = ap.parse_args()
assert os.environ.get("PF_STAMP"), "run with PF_STAMP=1"
import torch
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = bench.CAM
poses = wl.serpentine(cam, 100.0, a.frames + 20)
m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=0 if a.int16 else 1)
assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
fr = [torch.randint(0, 256, (cam[1], cam[0], 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
for k in range(a.frames + 20):
    m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k])
torch.cuda.synchronize()
L = pf.lib(); L.pf_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int]
n = L.pf_debug_phase_stamps(None, 1 << 16)
buf = np.zeros((n, 8), np.uint64)
assert L.pf_debug_phase_stamps(buf.ctypes.data, n) == n
wg = buf.reshape(-1, 32).astype(np.int64)            # 32 u64 per workgroup
wg = wg[wg[:, 1] > 0]
nw = int((wg[:, 0:24:2].sum(axis=0) + wg[:, 1:24:2].sum(axis=0) > 0).sum())
roles = {4: ["P0", "P1", "P2", "P3", "B0", "D0", "D1", "B1"], 8: ["P%d" % i for i in range(8)] + ["B0", "D0", "D1", "B1"]}.get(nw - 4, ["w%d" % i for i in range(nw)])
print("%d stamped workgroups, %d waves each (s_memtime ticks: 100 MHz constant clock -> 1 tick = 10 ns)" % (len(wg), nw))
for job in sorted(set(wg[:, 30])):
    b = wg[wg[:, 30] == job]
    life = b[:, 1]; per = b[:, 31]
    print("job %d: %5d workgroups, %.1f periods each, lifetime of wave 0: mean %.0f ticks = %.1f ticks per period" %
          (job, len(b), per.mean(), life.mean(), (life / np.maximum(per, 1)).mean()))
    for w in range(nw):
        wait, tot = b[:, 2 * w], b[:, 2 * w + 1]
        print("    %-3s waits at the barriers %5.1f %% of its life (busy %6.1f ticks per period)" %
              (roles[w], 100.0 * wait.sum() / max(tot.sum(), 1), ((tot - wait) / np.maximum(per, 1)).mean()))
