#!/bin/bash
# GPU box: the whole GPU suite on the tree as it stands, the JPEG rates, the bench lines of record
set -e
mkdir -p gpurun_out/r05
python -m pytest tests -x -q -m gpu > gpurun_out/r05/gpu_tests_final.txt 2>&1 || { tail -40 gpurun_out/r05/gpu_tests_final.txt; exit 1; }
tail -2 gpurun_out/r05/gpu_tests_final.txt
python tools/jpeg_rate.py --frames 12 --md gpurun_out/r05/jpeg_rate.md 2>&1 | grep -v Resolution | tail -7
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
