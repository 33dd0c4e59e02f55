#!/usr/bin/env python3
"""Output side of the path on the bench mosaic (dev tool): Ele::blend of every changed tile and save().
usage: tools/blend_save_rate.py [--int16] [--frames N] [--reps R]
Prints wall times into a fresh pageable buffer (first touch included), a touched pageable buffer and a page-locked one, and the
kernels' own time and algorithmic rate from the profile table."""
import argparse, importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--int16", action="store_true"); ap.add_argument("--frames", type=int, default=120)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
import numpy as np, torch
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = [4000, 3000, 3000, 3000, 2000, 1500]
poses = wl.serpentine(cam, 100.0, a.frames)
m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=0 if a.int16 else 1)
assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
fr = [torch.randint(0, 256, (3000, 4000, 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
for k in range(a.frames):
    m.feed_device(fr[k % 4].data_ptr(), 3000, 4000, poses[k])
m.sync()
nt = len(m.tiles())
m.profile_reset(); m.profile_enable(1)
def dump():
    for n, v in m.profile_read().items():
        if v["launches"]:
            print("    %-14s launches %5d  total %8.2f ms  alg %7.1f GB/s" % (n, v["launches"], v["ms"], v["alg_bytes"] / max(v["ms"], 1e-9) / 1e6))
    m.profile_reset()
t0 = time.perf_counter(); xy, out = m.blend_changed(cap=max(nt, 1)); t1 = time.perf_counter()
print("blend_changed: %d of %d tiles in %.1f ms = %.0f tiles/s (fresh pageable buffer, first touch included)" % (len(xy), nt, (t1 - t0) * 1e3, len(xy) / (t1 - t0)))
dump()
if hasattr(m, "blend_tiles"):
    tiles = list(xy)
    pinned = pf.host_array((len(tiles), 256, 256, 3))
    for name, buf in (("touched pageable", out), ("page-locked", pinned)):
        best = 1e9
        for _ in range(a.reps):
            t0 = time.perf_counter(); r = m.blend_tiles(tiles, out=buf); best = min(best, time.perf_counter() - t0)
        assert r is not None and (name == "touched pageable" or np.array_equal(pinned, out))
        print("blend_tiles  : %d tiles into a %s buffer: best of %d %.1f ms = %.0f tiles/s, %.1f GB/s of BGR8" %
              (len(tiles), name, a.reps, best * 1e3, len(tiles) / best, len(tiles) * 196608 / best / 1e9))
    dump()
t0 = time.perf_counter(); img = m.save_to_memory(); t1 = time.perf_counter()
print("save_to_memory: mosaic %dx%d (%d tiles) in %.1f ms (fresh pageable buffer)" % (img[0].shape[1], img[0].shape[0], nt, (t1 - t0) * 1e3))
dump()
if hasattr(pf, "host_array"):
    keep = {}
    def alloc_pinned(shape):
        if "p" not in keep: keep["p"] = pf.host_array(shape)
        return keep["p"]
    def alloc_touched(shape):
        return img[0]
    for name, al in (("touched pageable", alloc_touched), ("page-locked", alloc_pinned)):
        best = 1e9
        for _ in range(a.reps + 1):
            t0 = time.perf_counter(); m.save_to_memory(alloc=al); best = min(best, time.perf_counter() - t0)
        print("save_to_memory: into a %s buffer: best %.1f ms = %.1f GB/s of BGR8" % (name, best * 1e3, img[0].nbytes / best / 1e9))
    dump()
import tempfile
with tempfile.TemporaryDirectory() as d:
    for ext in ("png", "ppm"):
        p = os.path.join(d, "mosaic." + ext)
        t0 = time.perf_counter(); ok = m.save(p); t1 = time.perf_counter()
        print("save(%s): %s in %.2f s, file %.0f MB (collapse + D2H + encode on the host: PNG deflate level 1 in 256-row bands on up to 8 threads)" %
              (ext, "ok" if ok else "FAILED", t1 - t0, os.path.getsize(p) / 1e6))
