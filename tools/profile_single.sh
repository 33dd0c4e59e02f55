#!/bin/bash
# rocprofv3 passes of the Map2DCPU (single band) path on the bench workload: bash tools/profile_single.sh <tag> -> gpurun_out/single_<tag>/
set -e
tag=$1
export TMPDIR=/tmp
out=gpurun_out/single_$tag
mkdir -p $out
B="python3 tools/single_band_rate.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $B > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $B > $out/write.log 2>&1
python3 tools/pmc_summary.py stats $out/stats > $out/kernel_stats.md
python3 tools/pmc_summary.py traffic $out/fetch $out/write f32 2> $out/traffic.txt > /dev/null
cat $out/kernel_stats.md | head -5; cat $out/traffic.txt
