#!/bin/bash
# interleaved A/B/C... on ONE box: tools/abn.sh <rounds> "<bench args>" "SETTING1" "SETTING2" ...
# a setting is a space-separated list of VAR=value (PF_LIB=<path> selects another build of the library)
n=$1; args="$2"; shift 2
for i in $(seq $n); do
  for v in "$@"; do
    r=$(env $v python bench.py --no-cpu $args 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'])")
    echo "[$v] $r"
  done
done
