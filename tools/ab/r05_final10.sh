#!/bin/bash
# GPU box: the whole GPU suite on the last tree, smoke, the bench lines of record
set -e
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests_last.txt 2>&1 || { tail -40 $O/gpu_tests_last.txt; exit 1; }
tail -2 $O/gpu_tests_last.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/bench_last.json 2> $O/bench_last.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_last.json 2>/dev/null
python bench.py --int16 --no-cpu > $O/bench_int16_last.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_last", "bench_driver_last", "bench_int16_last"):
    d = json.loads(open("gpurun_out/r05/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("jpeg_feed", {}).get("value"), d.get("host_feed", {}).get("value"))
PY
