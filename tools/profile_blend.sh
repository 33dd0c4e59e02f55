#!/bin/bash
# rocprofv3 passes of the output side (Ele::blend of the bench mosaic's 2869 tiles, save of the 12800 x 15104 mosaic):
#   bash tools/profile_blend.sh <tag>          -> gpurun_out/blend_<tag>/{f32,int16}_{stats,fetch,write,sq*}, summaries in gpurun_out/blend_<tag>/*.md
# kernel trace and every --pmc set in its own pass; python3 directly after "--".
set -e
tag=$1
export TMPDIR=/tmp
out=gpurun_out/blend_$tag
mkdir -p $out
for dt in f32 int16; do
    fl=""; [ $dt = int16 ] && fl="--int16"
    B="python3 tools/blend_save_rate.py $fl --reps 3"
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/${dt}_stats -o s -- $B > $out/${dt}_stats.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${dt}_fetch -o f -- $B > $out/${dt}_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${dt}_write -o w -- $B > $out/${dt}_write.log 2>&1
    i=0
    for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
               "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
        i=$((i + 1))
        rocprofv3 --pmc $set --output-format csv -d $out/${dt}_sq$i -o q -- $B > $out/${dt}_sq$i.log 2>&1 || echo "set $i failed"
    done
    python3 tools/pmc_summary.py stats $out/${dt}_stats > $out/${dt}_kernel_stats.md
    python3 tools/pmc_summary.py traffic $out/${dt}_fetch $out/${dt}_write $dt 2> $out/${dt}_traffic.txt > /dev/null
    python3 tools/pmc_summary.py counters $out/${dt}_sq1 $out/${dt}_sq2 $out/${dt}_sq3 $out/${dt}_sq4 > $out/${dt}_counters.md
    grep -E "fused" $out/${dt}_kernel_stats.md $out/${dt}_traffic.txt || true
done
