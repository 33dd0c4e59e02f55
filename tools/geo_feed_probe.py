#!/usr/bin/env python3
"""dev probe: where does a geometry-only feed (pose without pixels) spend its time when called from Python?"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
import ctypes as C
cam = [4000, 3000, 3000, 3000, 2000, 1500]
poses = wl.serpentine(cam, 100.0, 320, max_rows=16)
m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1)
assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
L = pf.lib()
def t(name, f, n=300):
    t0 = time.perf_counter()
    for i in range(n): f(i)
    print("%-40s %8.2f us" % (name, (time.perf_counter() - t0) / n * 1e6))
t("m.feed(None, p)", lambda i: m.feed(None, poses[20 + i % 300]))
t("m.feed(None, p) again", lambda i: m.feed(None, poses[20 + i % 300]))
pp = [pf._pose(p) for p in poses]
t("L.pf_feed(h, None, pp)", lambda i: L.pf_feed(m._h, None, pp[20 + i % 300][1]))
t("_pose", lambda i: pf._pose(poses[20 + i % 300]))
print(m.timers())
