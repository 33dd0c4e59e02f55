#!/bin/bash
# Where do the LDS bank-conflict cycles and the texture-address stalls of the level kernel come from?  
#   bash tools/profile_lds_ta.sh <tag>    -> gpurun_out/lds_ta_<tag>/*.md
# LDS counters per stage via the stage ablation (PF_ABLATE: 0 = all stages, 2 = stage A only, 1 = stages B+D only with A
# filled with constants), ONE --pmc pass per setting; TA stall counters ONE COUNTER PER PASS (a wider TA set makes
# rocprofv3 fail with "error code 38: Request exceeds the capabilities of the hardware to collect").
export TMPDIR=/tmp
tag=${1:-r03}
out=gpurun_out/lds_ta_$tag
mkdir -p $out
B="python3 bench.py --no-cpu --steps 40 --warmup 5"
for ab in 0 2 1; do
    PF_ABLATE=$ab rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
        --output-format csv -d $out/lds_ablate$ab -o q -- $B > $out/lds_ablate$ab.log 2>&1 || echo "lds pass (ablate $ab) failed"
    echo "### PF_ABLATE=$ab" >> $out/lds_counters.md
    python3 tools/pmc_summary.py counters $out/lds_ablate$ab >> $out/lds_counters.md
    echo "lds ablate $ab done"
done
for c in TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum; do
    rocprofv3 --pmc $c --output-format csv -d $out/ta_$c -o q -- $B > $out/ta_$c.log 2>&1 || echo "pass $c failed (see $out/ta_$c.log)"
    python3 tools/pmc_summary.py counters $out/ta_$c >> $out/ta_counters.md 2>> $out/ta_errors.txt
    echo "$c done"
done
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm -o q -- $B > $out/grbm.log 2>&1
python3 tools/pmc_summary.py counters $out/grbm >> $out/ta_counters.md
