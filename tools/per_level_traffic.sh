#!/bin/bash
export PF_LIB=${PF_LIB:-pi-slam-fusion_amd/libpifusion_exp.so}   # PF_SINGLE_STREAM exists in the experiments build only (csrc/env.hpp)
# bash tools/per_level_traffic.sh <tag>  -> gpurun_out/per_level_<tag>.md   (one kernel per level: pf_options.fused = 3)
export TMPDIR=/tmp
tag=${1:-r03}; out=gpurun_out/per_level_$tag
for dt in f32 int16; do
    fl=""; [ $dt = int16 ] && fl="--int16"
    B="python3 bench.py --no-cpu --fused 3 --steps 40 --warmup 10 $fl"
    PF_SINGLE_STREAM=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${out}_${dt}_f -o f -- $B > ${out}_${dt}_f.log 2>&1
    PF_SINGLE_STREAM=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${out}_${dt}_w -o w -- $B > ${out}_${dt}_w.log 2>&1
    echo "## $dt" >> $out.md
    python3 tools/per_level_traffic.py ${out}_${dt}_f ${out}_${dt}_w >> $out.md
done
