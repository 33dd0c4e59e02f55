#!/bin/bash
# Collect the rocprofv3 evidence behind profiles/: kernel stats, HBM traffic counters and SQ
# instruction counters of bench.py's workload.  Run on the GPU box from the repo root:
#   bash tools/profile_round.sh <tag>        -> gpurun_out/prof_<tag>_*/
# (kernel trace and --pmc passes are separate runs; python3 directly after "--")
set -e
tag=${1:-rXX}
mode=${2:-all}        # all | sq (instruction counters only)
export TMPDIR=/tmp
out=gpurun_out
B="python3 bench.py --no-cpu --steps 100 --warmup 10"
for dt in f32 int16; do
    [ $mode = sq ] && break
    fl=""; [ $dt = int16 ] && fl="--int16"
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_${dt}_stats -o s -- $B $fl > $out/prof_${tag}_${dt}_stats.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/prof_${tag}_${dt}_fetch -o f -- $B $fl > $out/prof_${tag}_${dt}_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/prof_${tag}_${dt}_write -o w -- $B $fl > $out/prof_${tag}_${dt}_write.log 2>&1
    echo "$dt stats+traffic done"
done
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d $out/prof_${tag}_f32_sq$i -o q -- $B > $out/prof_${tag}_f32_sq$i.log 2>&1 || echo "set $i failed"
    echo "sq set $i done"
done
# Map2DCPU (single band) path
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_sb_stats -o s -- python3 tools/single_band_rate.py > $out/prof_${tag}_sb_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/prof_${tag}_sb_fetch -o f -- python3 tools/single_band_rate.py > $out/prof_${tag}_sb_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/prof_${tag}_sb_write -o w -- python3 tools/single_band_rate.py > $out/prof_${tag}_sb_write.log 2>&1
echo "single band done"
