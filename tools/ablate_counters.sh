#!/bin/bash
export TMPDIR=/tmp
for a in 0 2 4 1 3; do
  PF_ABLATE=$a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/abl_$a -o q -- python3 bench.py --no-cpu --steps 40 --warmup 4 > gpurun_out/abl_$a.log 2>&1 || echo fail $a
  echo "ablate $a"; python3 tools/pmc_summary.py counters gpurun_out/abl_$a | head -6
done
