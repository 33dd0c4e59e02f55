#!/usr/bin/env python3
"""Host-side cost per Map2D::feed: is a tile shard submission-bound?

Run with PF_ABLATE=3 in the environment (the level kernel then does next to nothing, so the stream never backs up and
what is timed is the host): per case the wall time of a Python feed call and, from the library's own section timers
(pf_timer_read, the reference's pi::timer section names), the C++ host time inside it:
    Map2D::feed                     whole pf_feed / pf_feed_device
    MultiBandMap2DCPU::renderFrame  footprint, canvas, homography, table, need rectangles, launch
    MultiBandMap2DCPU::Apply        the part from the tile pass on (table + rectangles + kernel launch)
Cases: unsharded; rank 0 of 8 with hash cells of 8 and of 2 tiles; geometry-only feeds (no pixels: another rank's frame)."""
import importlib
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench          # noqa: E402
import torch          # noqa: E402

pf = bench.load_package()
wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = [4000, 3000, 3000, 3000, 2000, 1500]
poses = wl.serpentine(cam, 100.0, 320, max_rows=16)
fr = torch.randint(0, 256, (3000, 4000, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def case(name, geometry_only=False, **opt):
    m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1, lookahead=int(os.environ.get("LA", "48")), **opt)
    assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
    m.reserve_tiles(3000 // max(1, opt.get("shard_count", 1)) + 400)
    feed = (lambda p: m.feed(None, p)) if geometry_only else (lambda p: m.feed_device(fr.data_ptr(), 3000, 4000, p))
    for p in poses[:20]:
        assert feed(p)
    m.sync(); m.timer_reset()
    t0 = time.perf_counter()
    for p in poses[20:]:
        assert feed(p)
    t1 = time.perf_counter()
    m.sync()
    t = m.timers()
    rec = {"case": name, "python_call_us": round((t1 - t0) / 300 * 1e6, 2), "rendered": m.stats()["rendered"] - 20,
           "sections_us": {k: {"mean": round(v["mean_s"] * 1e6, 2), "min": round(v["min_s"] * 1e6, 2), "max": round(v["max_s"] * 1e6, 2),
                               "calls": v["calls"]} for k, v in t.items() if v["calls"]}}
    m.close()
    print(json.dumps(rec))
    return rec


print("PF_ABLATE =", os.environ.get("PF_ABLATE"))
case("unsharded, device frames")
case("rank 0 of 8, cell 8 tiles", shard_rank=0, shard_count=8, shard_block=8)
case("rank 0 of 8, cell 2 tiles", shard_rank=0, shard_count=8, shard_block=2)
# bench.py's weak mode: a rank of 8 whose 128-tile hash cell holds the whole sortie -- it owns every tile of every canvas
own = pf.tile_owner(pf.default_options(shard_count=8, shard_block=128), 5, 5)
case("rank of 8 that owns the whole sortie (cell 128 tiles)", shard_rank=own, shard_count=8, shard_block=128)
case("geometry-only feeds", geometry_only=True)
