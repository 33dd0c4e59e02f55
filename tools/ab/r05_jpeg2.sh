#!/bin/bash
# GPU box: kernel stats of the decode with the Huffman pass on the GPU
set -e
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/prof_jpeg2 -o s -- python3 $R/tools/jpeg_rate.py --decode-only --frames 20 > $R/gpurun_out/r05/prof_jpeg2.log 2>&1 || { tail -20 $R/gpurun_out/r05/prof_jpeg2.log; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob("gpurun_out/r05/prof_jpeg2/**/s_kernel_stats.csv", recursive=True)[0])):
    print(r["Name"][:70], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3), "total ms %.2f" % (float(r["TotalDurationNs"]) / 1e6))
PY
