#!/usr/bin/env python3
"""Map2DCPU (single band) GPU path on the bench workload (dev tool): keyframes/s and the kernel's event time.
PF_SINGLE_OLD=1 selects the one-pixel-per-thread kernel."""
import importlib, os, sys, time
if os.environ.get("PF_SINGLE_OLD") or os.environ.get("PF_FORCE_GENERAL"):      # these switches exist in the experiments build only (csrc/env.hpp)
    os.environ.setdefault("PF_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pi-slam-fusion_amd", "libpifusion_exp.so"))
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
import torch
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = [4000, 3000, 3000, 3000, 2000, 1500]
n = 220
poses = wl.serpentine(cam, 100.0, n)
m = pf.Map2D.create(pf.TypeCPU, False)
assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
fr = [torch.randint(0, 256, (3000, 4000, 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
for k in range(20): m.feed_device(fr[k % 4].data_ptr(), 3000, 4000, poses[k])
m.sync(); m.profile_enable(1)
t0 = time.perf_counter()
for k in range(20, n): m.feed_device(fr[k % 4].data_ptr(), 3000, 4000, poses[k])
m.sync(); dt = time.perf_counter() - t0
for name, v in m.profile_read().items():
    if v["launches"]:
        print("%-10s launches %4d avg %7.2f us  alg %7.1f GB/s" % (name, v["launches"], v["ms"] * 1e3 / v["launches"], v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9))
print("%.0f keyframes/s (events on every launch)" % ((n - 20) / dt))
