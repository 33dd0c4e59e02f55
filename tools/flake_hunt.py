#!/usr/bin/env python3
"""Diagnostics: run the GPU parity tests once (dirty state), then repeat test_row_padded_frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import load_package
import numpy as np
import test_gpu_parity as T
from helpers import workloads, jitter_poses, map_digest
pf = load_package()
from oracle import orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
T.test_stress_geometry_8000x6000_7band(pf, orc, 0); T.test_stress_geometry_8000x6000_7band(pf, orc, 1)
wl = workloads()
cam = [640, 480, 500, 500, 320, 240]
poses = jitter_poses(3, seed=4)
bad = 0
for it in range(n):
    a = pf.Map2D.create(pf.TypeMultiBandCPU, False); b = pf.Map2D.create(pf.TypeMultiBandCPU, False)
    assert a.prepare(wl.IDENTITY_PLANE, cam, poses) and b.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        wide = wl.noise_frame(480, 700, 70 + k)
        view = wide[:, 30:670]
        assert a.feed(view, p) and b.feed(np.ascontiguousarray(view), p)
    a.sync(); b.sync()
    da, db = map_digest(a), map_digest(b)
    if da != db:
        bad += 1
        print("it", it, "differ:", sorted(k for k in da if da[k] != db[k]), flush=True)
print("iterations", n, "bad", bad)
