#!/usr/bin/env python3
"""dev probe: do the tails of one map's launches leave room that a second, independent stream can fill?  Two maps (own
streams, own sorties) fed alternately against one map fed alone: aggregate keyframes/s.  If two interleaved streams
are no faster than one, splitting a keyframe's level-0 job and its upper-level jobs over two streams cannot pay either."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
import torch
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = bench.CAM
ff = 0 if "--int16" in sys.argv else 1
N = 220
fr = [torch.randint(0, 256, (3000, 4000, 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()


def make(origin):
    poses = wl.serpentine(cam, 100.0, N, max_rows=16, origin=origin)
    m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=ff)
    assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
    m.reserve_tiles(3200)
    return m, poses


def run(maps):
    for k in range(20):
        for m, p in maps:
            m.feed_device(fr[k % 4].data_ptr(), 3000, 4000, p[k])
    for m, _ in maps:
        m.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20, N):
        for m, p in maps:
            m.feed_device(fr[k % 4].data_ptr(), 3000, 4000, p[k])
    for m, _ in maps:
        m.sync()
    torch.cuda.synchronize()
    return len(maps) * (N - 20) / (time.perf_counter() - t0)


a = make((0.0, 0.0))
print("one stream : %.0f keyframes/s" % run([a]))
a[0].close()
a, b = make((0.0, 0.0)), make((5000.0, 0.0))
print("two streams: %.0f keyframes/s aggregate" % run([a, b]))
