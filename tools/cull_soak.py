#!/usr/bin/env python3
"""Soak of the cull against the oracle: tests/test_gpu_cull.py's generators over many seeds (dev tool).

usage: cull_soak.py [cases] [first seed] [--margins "px,w;px,w;..."] [--steep N]

--margins: the cull's margins (FusionMap::cell_out: source pixels on a distance, a term on a weight; defaults 2, 1e-5) are set through
PF_CULL_MARGIN_PX / PF_CULL_MARGIN_W per setting, and every case is run once per setting: the table at the end -- cases with a mismatch
against the oracle per setting -- is the measured safety factor of the defaults (profiles/r05_cull_margins.md).
--steep N: N more cases from the steep-tilt generator (corner rays up to the 0.4 obliqueness gate)."""
import os, sys
os.environ.setdefault("PF_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pi-slam-fusion_amd", "libpifusion_exp.so"))   # the switches below exist in the experiments build only (csrc/env.hpp)
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from conftest import load_package
import test_gpu_cull as T
from helpers import compare_maps, workloads
pf = load_package()
from oracle import orc

args = [a for a in sys.argv[1:]]
margins, steep = [None], 0
if "--margins" in args:
    i = args.index("--margins"); margins = [tuple(s.split(",")) for s in args[i + 1].split(";")]; del args[i:i + 2]
if "--steep" in args:
    i = args.index("--steep"); steep = int(args[i + 1]); del args[i:i + 2]
n = int(args[0]) if len(args) > 0 else 40
first = int(args[1]) if len(args) > 1 else 0


def steep_case(seed):
    wl = workloads()
    poses = T.tilted_poses(wl, 14, seed, 80.0, 16.0, 27.5, 9.0)
    g, o, frames = T.feed_both(pf, orc, poses, poses[:6], seed, force_float=seed & 1, weight_type=(seed >> 1) & 1, scale=1.5)
    miss = compare_maps(g, o)
    res = (miss, frames, g.culled_tiles(), g.culled_cells(), "steep ff=%d wt=%d" % (seed & 1, (seed >> 1) & 1))
    g.close()
    return res


table = []
for mg in margins:
    if mg is not None:
        os.environ["PF_CULL_MARGIN_PX"], os.environ["PF_CULL_MARGIN_W"] = mg
    bad = frames = cells = tiles = px_bad = 0
    for seed in range(first, first + n + steep):
        miss, fr, t, c, what = T.run_case(pf, orc, seed) if seed < first + n else steep_case(seed - n)
        frames += fr; tiles += t; cells += c; bad += bool(miss); px_bad += len(miss)
        print("margins %s seed %d %s culled tiles %d cells %d %s" % (mg, seed, what, t, c, "MISMATCH " + str(miss[:3]) if miss else "ok"), flush=True)
    table.append((mg, n + steep, frames, tiles, cells, bad, px_bad))
    print("margins", mg, "cases", n + steep, "frames rendered", frames, "culled tiles", tiles, "cells", cells, "cases with mismatches", bad, flush=True)
print("| margin px | margin w | cases | keyframes | culled tiles | culled cells | cases with a mismatch | mismatching tile levels |")
print("|---|---|---|---|---|---|---|---|")
for mg, nc, fr, t, c, bad, pxb in table:
    print("| %s | %s | %d | %d | %d | %d | %d | %d |" % ((mg or ("2 (default)", "1e-5 (default)")) + (nc, fr, t, c, bad, pxb)))
