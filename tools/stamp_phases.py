#!/usr/bin/env python3
"""Where a workgroup of the level kernel spends its life, measured in the real mix (diagnostic build: PF_STAMP=1
selects a stamped instantiation of k_levels; the product kernel carries no stamps).
usage: PF_STAMP=1 python tools/stamp_phases.py [--int16] [--frames N]"""
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
    if k == a.frames + 18:
        pass
# the last full launch's stamps are overwritten by the flush launches of sync(): read before syncing the pipeline
torch.cuda.synchronize()
L = pf.lib(); L.pf_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int]
import time; time.sleep(0.2)
n = L.pf_debug_phase_stamps(None, 1 << 16)
buf = np.zeros((n, 8), np.uint64)
assert L.pf_debug_phase_stamps(buf.ctypes.data, n) == n
buf = buf[buf[:, 0] > 0]
names = ["A (table entry, stage + warp)", "wait at barrier 1", "B (pyrDown)", "wait at barrier 2", "D (Laplacian, select, stores drained)"]
# s_memtime counters of different XCDs do not share a base: spans are formed per XCD (stamp slot 7 = XCC id)
spans = []
for x in sorted(set(buf[:, 7].astype(int))):
    bx = buf[buf[:, 7] == x].astype(np.int64)
    spans.append(int(bx[:, 5].max() - bx[:, 0].min()))
print("%d stamped workgroups on %d XCDs; launch span per XCD: median %d ticks (min %d, max %d)" %
      (len(buf), len(spans), int(np.median(spans)), min(spans), max(spans)))
entry = (buf[:, 6] >> np.uint64(4)).astype(np.int64)          # s_memtime at kernel entry (slot 6 = job | entry << 4)
buf[:, 6] &= np.uint64(15)
for job in sorted(set(buf[:, 6].astype(int))):
    sel = buf[:, 6] == job
    b = buf[sel].astype(np.int64)
    pro = b[:, 0] - entry[sel]
    print("job %d: prologue (kernel entry -> block start: job pick, need test, scalar loads) mean %7.0f  median %7.0f  p90 %7.0f ticks" % (job, pro.mean(), np.median(pro), np.percentile(pro, 90)))
    d = np.diff(b[:, :6], axis=1)
    life = b[:, 5] - b[:, 0]
    print("job %d: %5d workgroups, lifetime mean %7.0f ticks (min %d, max %d)" % (job, len(b), life.mean(), life.min(), life.max()))
    for i, nm in enumerate(names):
        print("    %-40s mean %7.0f  median %7.0f  p90 %7.0f  = %4.1f %% of lifetime" % (nm, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90), 100 * d[:, i].sum() / life.sum()))
    # start / end relative to the first start on the same XCD
    rel0, rel5 = [], []
    for x in sorted(set(b[:, 7])):
        bx = b[b[:, 7] == x]; base = buf[buf[:, 7] == x].astype(np.int64)[:, 0].min()
        rel0 += list(bx[:, 0] - base); rel5 += list(bx[:, 5] - base)
    print("    starts (per-XCD base): first %d, median %d, last %d; ends: median %d, last %d" % (min(rel0), np.median(rel0), max(rel0), np.median(rel5), max(rel5)))
