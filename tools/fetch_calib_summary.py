#!/usr/bin/env python3
"""Per-kernel means of FETCH_SIZE / WRITE_SIZE of tools/cpp/fetch_calib.bin against its known byte counts.
usage: tools/fetch_calib_summary.py <fetch pmc dir> <write pmc dir>"""
import csv, glob, os, re, sys
from collections import defaultdict

FRAME = 4000 * 3000 * 3
NPX = 12 << 20
KNOWN = {"k_stream_rd<16>": ("read", FRAME), "k_stream_rd<8>": ("read", FRAME), "k_stream_rd<4>": ("read", FRAME),
         "k_gather8_blocks": ("read", FRAME), "k_gather8": ("read", FRAME),
         "k_store12<8>": ("write", NPX * 16), "k_store12<4>": ("write", NPX * 16 // 2), "k_store12<2>": ("write", NPX * 16 // 4),
         "k_stream_wr": ("write", NPX * 12 // 16 * 16)}


def means(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
print("| kernel | known bytes | FETCH_SIZE x 1024 | WRITE_SIZE x 1024 | bytes per counted byte |")
print("|---|---|---|---|---|")
for k, (kind, b) in KNOWN.items():
    fk = [v for n, v in fetch.items() if n.startswith(k)]; wk = [v for n, v in write.items() if n.startswith(k)]
    fv = fk[0] * 1024 if fk else 0.0; wv = wk[0] * 1024 if wk else 0.0
    cnt = fv if kind == "read" else wv
    print("| %s | %.1f MB (%s) | %.1f MB | %.1f MB | %.3f |" % (k, b / 1e6, kind, fv / 1e6, wv / 1e6, b / cnt if cnt else float("nan")))
