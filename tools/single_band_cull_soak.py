#!/usr/bin/env python3
"""Soak of the Map2DCPU path's cull (round 6) against the oracle, which renders everything: random overlapping sorties (scale, yaw / tilt
jitter, weight type, overlaps, image content), every tile compared byte for byte.   usage: tools/single_band_cull_soak.py [first seed] [count]
The eight seeds of tests/test_single_band.py::test_single_band_cull_leaves_the_tiles_alone are the first eight of this sweep."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from conftest import load_package
from helpers import workloads
from oracle import orc
import test_single_band as T
pf = load_package(); wl = workloads()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bad = tiles_all = culled_t = culled_c = frames_all = 0
for seed in range(first, first + count):
    rs = np.random.RandomState(7300 + seed)
    cam = [640, 480, 500, 500, 320, 240]
    wt = seed & 1; scale = float(rs.choice([1.0, 1.5, 2.0, 3.0]))
    yaw = float(rs.choice([3.0, 12.0, 45.0, 180.0])); tilt = float(rs.choice([0.5, 4.0, 10.0]))
    nfr = int(rs.randint(12, 22))
    poses = wl.serpentine(cam, float(rs.uniform(60, 140)), nfr, per_row=int(rs.randint(3, 7)), fwd_overlap=float(rs.uniform(0.5, 0.9)),
                          side_overlap=float(rs.uniform(0.3, 0.8)), seed=seed, yaw_jitter_deg=yaw, tilt_jitter_deg=tilt, max_rows=3)
    poses = poses + [list(p) for p in poses[:3]]
    frames = [wl.noise_frame(480, 640, 50 * seed + k) if k % 3 else wl.smooth_frame(480, 640, k) for k in range(len(poses))]
    g, o = T.run_pair(pf, orc, cam, poses, frames, pf.TypeCPU if seed & 2 else pf.TypeGPU, n_prepare=6, weight_type=wt, scale=scale)
    miss = [t for t in o.tiles() if not np.array_equal(g.tile_bgra(*t), o.tile_bgra(*t))]
    bad += bool(miss); tiles_all += len(o.tiles()); culled_t += g.culled_tiles(); culled_c += g.culled_cells(); frames_all += len(poses)
    print("seed %d wt %d scale %.1f yaw %.0f tilt %.1f frames %d tiles %d culled tiles %d cells %d %s" %
          (seed, wt, scale, yaw, tilt, len(poses), len(o.tiles()), g.culled_tiles(), g.culled_cells(), "MISMATCH %s" % miss[:3] if miss else "ok"), flush=True)
    g.close()
print("sorties %d keyframes %d tiles %d culled tiles %d culled cells %d sorties with a mismatch %d" % (count, frames_all, tiles_all, culled_t, culled_c, bad))
sys.exit(1 if bad else 0)
