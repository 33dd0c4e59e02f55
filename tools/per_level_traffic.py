#!/usr/bin/env python3
"""HBM bytes per pyramid level (how much of the launch's traffic is the GW_1..4 round trip?).
Input: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --fused 3`, where every level of a keyframe is its
own k_level3 dispatch; dispatches are told apart by grid size.  usage: per_level_traffic.py <fetch dir> <write dir>"""
import csv, glob, os, sys
from collections import defaultdict


def rows(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    return [r for r in csv.DictReader(open(f)) if "k_level3<" in r["Kernel_Name"]]


def level_of(g, gmax):
    """dispatches of one pyramid level differ a little in grid size (the canvas is 17x13 or 18x13 tiles ...); levels differ
    by a factor of four"""
    import math
    return int(round(math.log(gmax / g, 4)))


def means(d, counter):
    rs = [r for r in rows(d) if r["Counter_Name"] == counter]
    gmax = max(int(r["Grid_Size"]) for r in rs)
    acc = defaultdict(list)
    for r in rs:
        acc[level_of(int(r["Grid_Size"]), gmax)].append((float(r["Counter_Value"]), int(r["Grid_Size"]) // 512))
    return ({l: sum(v for v, _ in a) / len(a) for l, a in acc.items()}, {l: len(a) for l, a in acc.items()},
            {l: sum(w for _, w in a) / len(a) for l, a in acc.items()})


f, nf, wg = means(sys.argv[1], "FETCH_SIZE")
w, _, _ = means(sys.argv[2], "WRITE_SIZE")
print("| level | workgroups (mean) | dispatches | fetch MB (x2: gfx950 correction) | write MB | sum MB |")
print("|---|---|---|---|---|---|")
tot_f = tot_w = 0
for lvl in sorted(f):
    fb, wb = 2 * f[lvl] * 1024 / 1e6, w.get(lvl, 0) * 1024 / 1e6
    tot_f += fb; tot_w += wb
    print("| %d | %.0f | %d | %.1f | %.1f | %.1f |" % (lvl, wg[lvl], nf[lvl], fb, wb, fb + wb))
print("| all | | | %.1f | %.1f | %.1f |" % (tot_f, tot_w, tot_f + tot_w))
