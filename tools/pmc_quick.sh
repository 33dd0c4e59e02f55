#!/bin/bash
# quick SQ counter passes of bench.py's workload for the current environment: tools/pmc_quick.sh <outdir> [bench args]
# (python3 directly after "--"; one --pmc set per run)
export TMPDIR=/tmp
out=$1; shift
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d $out/sq$i -o q -- python3 bench.py --no-cpu --steps 40 --warmup 10 "$@" > $out/sq$i.log 2>&1 || echo "set $i failed"
done
python3 tools/pmc_summary.py counters $out/sq1 $out/sq2 $out/sq3 $out/sq4 > $out/counters.md
cat $out/counters.md
