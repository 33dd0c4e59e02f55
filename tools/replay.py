#!/usr/bin/env python3
"""Replay a DroneMap-style dataset through the GPU fusion engine the way the reference's
file driver does (backup/map2dfusion.cpp testMap2D): PrepareFrameNum frames size the grid,
the rest are fed while queueSize() < 2, save() at the end.

    python tools/replay.py <datapath> [--prepare 10] [--thread 1] [--fps 0] [--scale 1] [--float] [--out result.png]
"""
import argparse
import importlib
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402  (package loader)

ap = argparse.ArgumentParser()
ap.add_argument("datapath"); ap.add_argument("--prepare", type=int, default=10)       # PrepareFrameNum
ap.add_argument("--thread", type=int, default=1); ap.add_argument("--fps", type=float, default=0)
ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--float", action="store_true")
ap.add_argument("--out", default="result.png")                                         # Map.File2Save
a = ap.parse_args()
pf = bench.load_package()
ds = importlib.import_module("pi_slam_fusion_amd.dataset").DroneMapDataset(a.datapath)
dt = importlib.import_module("pi_slam_fusion_amd.datatrans")
m = pf.Map2D.create(pf.TypeMultiBandCPU, bool(a.thread), force_float=1 if a.float else 0, scale=a.scale)
n0 = min(a.prepare, len(ds))
first = [ds.load(k) for k in range(n0)]
print("Loaded %d frames." % n0)
assert m.prepare(ds.plane, ds.camera, [p for _, p in first], images=[i for i, _ in first] if a.thread else None)
it = iter(range(n0, len(ds)))          # obtainFrame consumed the prepare frames (a thread=0 map never renders them, Map2D.cpp:42)
t0 = time.perf_counter()
fed = dt.feed_loop(m, lambda: (lambda k: None if k is None else ds.load(k, encoded=True))(next(it, None)), a.fps)   # .jpg frames: decoded by the map
m.sync()
print("fed %d frames in %.2f s; stats %s" % (fed, time.perf_counter() - t0, m.stats()))
print("save ->", a.out, m.save(a.out))
