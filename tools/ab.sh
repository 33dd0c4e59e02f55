#!/bin/bash
# A/B of two settings on ONE box, interleaved rounds (cdna guide rule 24):
#   tools/ab.sh "ENVA=1" "ENVB=1" [rounds] [bench args]
# a setting is a space-separated list of VAR=value; PF_LIB=<path> selects another build of the library (tools/ab/*.so)
a="$1"; b="$2"; n=${3:-3}; shift 3
for i in $(seq $n); do
  for v in "$a" "$b"; do
    r=$(env $v python bench.py --no-cpu "$@" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'])")
    echo "[$v] $r"
  done
done
