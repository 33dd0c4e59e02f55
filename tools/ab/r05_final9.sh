#!/bin/bash
# GPU box: bench lines of record on the final tree + one more kernel trace of the profiled sortie (gaps between launches)
set -e
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
python bench.py > $O/bench_default_b.json 2> $O/bench_default_b.err; python -c "
import json; d=json.loads(open('$O/bench_default_b.json').read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))"
python bench.py --int16 --no-cpu > $O/bench_int16_b.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_int16_b.json').read().strip().splitlines()[-1]); print('int16', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_b.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_driver_b.json').read().strip().splitlines()[-1]); print('driver', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32_stats_c -o s -- python3 bench.py --no-cpu --steps 100 --warmup 10 > $O/prof_f32_stats_c.log 2>&1
python3 tools/pmc_summary.py stats $O/prof_f32_stats_c 2>&1 | tail -4
