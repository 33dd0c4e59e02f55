#!/usr/bin/env python3
"""Soak of the cull against the oracle: tests/test_gpu_cull.py's generator over many seeds (dev tool).
usage: cull_soak.py [cases] [first seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from conftest import load_package
import test_gpu_cull as T
pf = load_package()
from oracle import orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = frames = cells = tiles = 0
for seed in range(first, first + n):
    miss, fr, t, c, what = T.run_case(pf, orc, seed)
    frames += fr; tiles += t; cells += c; bad += bool(miss)
    print("seed %d %s culled tiles %d cells %d %s" % (seed, what, t, c, "MISMATCH " + str(miss[:3]) if miss else "ok"), flush=True)
print("cases", n, "frames rendered", frames, "culled tiles", tiles, "cells", cells, "cases with mismatches", bad)
