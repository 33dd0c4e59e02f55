#!/usr/bin/env python3
"""Per-kernel HIP-event times of the fused path on the bench workload (dev tool).
usage: tools/kprof.py [--int16] [--frames N]   (PF_ABLATE=<bits> for stage ablation)"""
import argparse, importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--int16", action="store_true"); ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--fused", type=int, default=1); ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--mul", type=int, default=1, help="frame edge multiplier (2: 8000x6000 frames)")
ap.add_argument("--no-events", action="store_true", help="no HIP events around the launches (under rocprofv3 --kernel-trace: the trace's own durations)")
a = ap.parse_args()
import torch
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = [4000 * a.mul, 3000 * a.mul, 3000 * a.mul, 3000 * a.mul, 2000 * a.mul, 1500 * a.mul]
poses = wl.serpentine(cam, 100.0, a.frames + 20)
m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=0 if a.int16 else 1, fused=a.fused, scale=a.scale)
assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
fr = [torch.randint(0, 256, (cam[1], cam[0], 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
for k in range(20): m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k])
m.sync(); m.profile_enable(0 if a.no_events else 1)
import time
t0 = time.perf_counter()
for k in range(20, 20 + a.frames): m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k])
m.sync(); dt = time.perf_counter() - t0
for n, v in m.profile_read().items():
    if v["launches"]:
        print("%-14s launches %5d  avg %8.2f us  per-frame %8.2f us  alg %7.1f GB/s" % (n, v["launches"], v["ms"] * 1e3 / v["launches"], v["ms"] * 1e3 / a.frames, v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9))
print("wall per frame (with events): %.1f us" % (dt / a.frames * 1e6))
