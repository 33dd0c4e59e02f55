#!/usr/bin/env python3
"""Gaps between consecutive launches of the level kernel in a rocprofv3 --kernel-trace CSV:
   tools/launch_gaps.py <dir with *_kernel_trace.csv> [kernel substring = k_levels<]"""
import csv, glob, os, sys
import numpy as np
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "k_levels<"
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
size = lambda r: int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])
k = [r for r in rows if pat in r["Kernel_Name"]]
full = max(size(r) for r in k)
gaps, durs, between = [], [], {}
fulls = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"] and size(r) * 2 > full]
for i, j in zip(fulls[:-1], fulls[1:]):
    a, b = rows[i], rows[j]
    if any(pat in rows[t]["Kernel_Name"] for t in range(i + 1, j)):
        continue                                       # a flush launch in between: not back to back
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    durs.append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    for t in range(i + 1, j):
        n = rows[t]["Kernel_Name"][:40]; between[n] = between.get(n, 0) + 1
g = np.array(gaps); du = np.array(durs)
print("%d back-to-back full-size `%s` launches: duration mean %.2f us median %.2f; gap to the next launch: median %.2f us, mean %.2f, p10 %.2f, p90 %.2f" %
      (len(g), pat, du.mean(), np.median(du), np.median(g), g.mean(), np.percentile(g, 10), np.percentile(g, 90)))
other = sorted(set(r["Kernel_Name"][:60] for r in rows if pat not in r["Kernel_Name"]))
print("dispatches between two such launches:", between if between else "none")
