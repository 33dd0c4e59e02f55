#!/bin/bash
# HBM traffic and VALU instruction counts of the TIMED launches of given bench.py windows (the counters behind a
# bench line must be those of the launches that line times).  Run on the GPU box from the repo root:
#   bash tools/profile_windows.sh <tag> "<K> <W> <PRE> [nocull]" ...        e.g.  "20 5 15" "200 20 0" "20 5 15 nocull"
# -> gpurun_out/summary_<tag>/pmc_traffic.json (records under windows[bench.window_key(K, W, PRE, nocull)]) + one text file per pass.
# Separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU); python3 directly after "--".
set -e
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out/summary_$tag
for win in "$@"; do
    set -- $win
    K=$1; W=$2; PRE=$3; NC=$4
    key="k${K}_w${W}_pre${PRE}${NC:+_nocull}"
    for dt in ${PF_DTYPES:-f32 int16}; do
        fl=""; [ $dt = int16 ] && fl="--int16"
        [ -n "$NC" ] && fl="$fl --no-cull"
        B="python3 bench.py --no-cpu --steps $K --warmup $W --pre $PRE $fl"
        rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/win_${tag}_${dt}_${key}_fetch -o f -- $B > $out/win_${tag}_${dt}_${key}_fetch.log 2>&1
        rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/win_${tag}_${dt}_${key}_write -o w -- $B > $out/win_${tag}_${dt}_${key}_write.log 2>&1
        rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/win_${tag}_${dt}_${key}_sq -o q -- $B > $out/win_${tag}_${dt}_${key}_sq.log 2>&1
        PF_WINDOW="$win" python3 tools/pmc_summary.py traffic $out/win_${tag}_${dt}_${key}_fetch $out/win_${tag}_${dt}_${key}_write $dt \
            $out/summary_$tag/pmc_traffic.json $out/win_${tag}_${dt}_${key}_sq 2> $out/summary_$tag/${tag}_${dt}_${key}_traffic.txt
        echo "$dt $key done: $(cat $out/summary_$tag/${tag}_${dt}_${key}_traffic.txt | tr '\n' ' ')"
    done
done
