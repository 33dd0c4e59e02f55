#!/usr/bin/env python3
"""Longer run of tests/test_gpu_fuzz.py's generator at larger frame sizes (dev tool).  usage: fuzz_soak.py [cases] [scale]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from conftest import load_package
from helpers import compare_maps, workloads
import numpy as np
import test_gpu_fuzz as F
pf = load_package(); wl = workloads()
from oracle import orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bad = rendered = 0
for seed in range(n):
    rs = np.random.RandomState(5000 + seed)
    cam, poses, frames = F.random_case(rs, wl)
    cam = [c * scale for c in cam]
    frames = [wl.noise_frame(cam[1], cam[0], int(rs.randint(1 << 20))) for _ in frames]
    ff = seed & 1
    single = seed % 5 == 4
    if single:
        g = pf.Map2D.create(pf.TypeCPU, False); o = orc.OracleMap(single_band=True)
    else:
        g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=ff); o = orc.OracleMap(force_float=ff)
    assert g.prepare(wl.IDENTITY_PLANE, cam, poses[:2]) == o.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    for img, p in zip(frames, poses):
        a, b = g.feed(img, p), o.feed(img, p)
        assert a == b, (seed, a, b)
        rendered += bool(a)
    g.sync()
    if single:
        miss = [t for t in o.tiles() if not np.array_equal(g.tile_bgra(*t), o.tile_bgra(*t))]
    else:
        miss = compare_maps(g, o)
    if miss:
        bad += 1
        print("seed", seed, "MISMATCH", miss[:3], flush=True)
print("cases", n, "frames rendered", rendered, "cases with mismatches", bad)
