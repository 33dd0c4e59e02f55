#!/bin/bash
# GPU box: the JPEG tests, the replay tests, the rate table, and kernel stats + HBM traffic of the device back end
set -e
mkdir -p gpurun_out/r05
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_jpeg.py tests/test_replay.py tests/test_jpeg.py -x -q -m "gpu or not gpu" > gpurun_out/r05/jpeg_tests.txt 2>&1 || { tail -30 gpurun_out/r05/jpeg_tests.txt; exit 1; }
tail -3 gpurun_out/r05/jpeg_tests.txt
python tools/jpeg_rate.py --frames 12 --md gpurun_out/r05/jpeg_rate.md > gpurun_out/r05/jpeg_rate.txt 2>&1 || { tail -30 gpurun_out/r05/jpeg_rate.txt; exit 1; }
cat gpurun_out/r05/jpeg_rate.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/prof_jpeg -o s -- python3 $R/tools/jpeg_rate.py --decode-only --frames 20 > $R/gpurun_out/r05/prof_jpeg.log 2>&1 || { tail -20 $R/gpurun_out/r05/prof_jpeg.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r05/prof_jpeg_f -o f -- python3 $R/tools/jpeg_rate.py --decode-only --frames 20 > $R/gpurun_out/r05/prof_jpeg_f.log 2>&1 || { tail -20 $R/gpurun_out/r05/prof_jpeg_f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r05/prof_jpeg_w -o w -- python3 $R/tools/jpeg_rate.py --decode-only --frames 20 > $R/gpurun_out/r05/prof_jpeg_w.log 2>&1 || { tail -20 $R/gpurun_out/r05/prof_jpeg_w.log; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob
def one(pat): return glob.glob(pat, recursive=True)[0]
st = list(csv.DictReader(open(one("gpurun_out/r05/prof_jpeg/**/s_kernel_stats.csv"))))
for r in st:
    if "jpeg" in r["Name"] or "copy" in r["Name"].lower(): print(r["Name"][:60], r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
for tag, f in (("FETCH_SIZE", one("gpurun_out/r05/prof_jpeg_f/**/f_counter_collection.csv")), ("WRITE_SIZE", one("gpurun_out/r05/prof_jpeg_w/**/w_counter_collection.csv"))):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "jpeg" in r["Kernel_Name"]:
            acc.setdefault(r["Kernel_Name"][:40], []).append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(tag, k, "n", len(v), "mean", sum(v) / len(v))
PY
