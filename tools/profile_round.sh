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
# VALU instruction count of the int16 build too (bench.py's roofline.valu)
[ $mode = sq ] || rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/prof_${tag}_int16_sq1 -o q -- $B --int16 > $out/prof_${tag}_int16_sq1.log 2>&1
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

# summaries (what gets committed under profiles/): PF_GIT_SHA=<sha> bash tools/profile_round.sh <tag>
mkdir -p $out/summary_$tag
for dt in f32 int16; do
    python3 tools/pmc_summary.py stats $out/prof_${tag}_${dt}_stats > $out/summary_$tag/${tag}_${dt}_kernel_stats.md
    python3 tools/launch_gaps.py $out/prof_${tag}_${dt}_stats >> $out/summary_$tag/${tag}_${dt}_kernel_stats.md
    python3 tools/pmc_summary.py traffic $out/prof_${tag}_${dt}_fetch $out/prof_${tag}_${dt}_write $dt $out/summary_$tag/pmc_traffic.json $out/prof_${tag}_${dt}_sq1 2> $out/summary_$tag/${tag}_${dt}_traffic.txt
done
python3 tools/pmc_summary.py counters $out/prof_${tag}_f32_sq1 $out/prof_${tag}_f32_sq2 $out/prof_${tag}_f32_sq3 $out/prof_${tag}_f32_sq4 > $out/summary_$tag/${tag}_f32_sq_counters.md
python3 tools/pmc_summary.py stats $out/prof_${tag}_sb_stats > $out/summary_$tag/${tag}_single_band_kernel_stats.md
python3 tools/pmc_summary.py traffic $out/prof_${tag}_sb_fetch $out/prof_${tag}_sb_write single_band $out/summary_$tag/pmc_traffic.json 2> $out/summary_$tag/${tag}_single_band_traffic.txt
echo "summaries in $out/summary_$tag"
