#!/usr/bin/env python3
"""PCIe-inclusive rate: frames handed over as host buffers through pf_feed.
usage: tools/host_feed_rate.py [--int16] [--frames N] [--pinned] [--thread]"""
import argparse, importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--int16", action="store_true"); ap.add_argument("--frames", type=int, default=100)
ap.add_argument("--pinned", action="store_true"); ap.add_argument("--thread", action="store_true")
a = ap.parse_args()
import numpy as np, torch
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = [4000, 3000, 3000, 3000, 2000, 1500]
poses = wl.serpentine(cam, 100.0, a.frames + 20)
m = pf.Map2D.create(pf.TypeMultiBandCPU, a.thread, force_float=0 if a.int16 else 1)
assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
fr = []
for k in range(4):
    t = torch.randint(0, 256, (3000, 4000, 3), dtype=torch.uint8)
    if a.pinned: t = t.pin_memory()
    fr.append(t.numpy())
for k in range(10): m.feed(fr[k % 4], poses[k])
m.sync()
t0 = time.perf_counter()
for k in range(10, 10 + a.frames): m.feed(fr[k % 4], poses[k])
m.sync(); dt = time.perf_counter() - t0
print("host-fed %s frames, pinned=%s thread=%s: %.1f keyframes/s (%.2f ms/frame, %.1f GB/s H2D)" %
      (a.frames, a.pinned, a.thread, a.frames / dt, dt / a.frames * 1e3, 36e6 * a.frames / dt / 1e9), m.stats())
