#!/bin/bash
# round 6, after the lookahead: the order of a launch's jobs again (upper levels last = product / first / dealt between the level-0 groups), experiments library, interleaved
out=gpurun_out/r06_job_order; mkdir -p $out
export PF_LIB=pi-slam-fusion_amd/libpifusion_exp.so
for rep in 1 2 3; do
  python3 bench.py --no-cpu > $out/last_f32_r$rep.json 2>/dev/null
  PF_UPPER_FIRST=1 python3 bench.py --no-cpu > $out/first_f32_r$rep.json 2>/dev/null
  PF_INTERLEAVE_JOBS=1 python3 bench.py --no-cpu > $out/mixed_f32_r$rep.json 2>/dev/null
  python3 bench.py --no-cpu --int16 > $out/last_i16_r$rep.json 2>/dev/null
  PF_UPPER_FIRST=1 python3 bench.py --no-cpu --int16 > $out/first_i16_r$rep.json 2>/dev/null
  PF_INTERLEAVE_JOBS=1 python3 bench.py --no-cpu --int16 > $out/mixed_i16_r$rep.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for k in ("last_f32","first_f32","mixed_f32","last_i16","first_i16","mixed_i16"):
    v=[]
    for f in sorted(glob.glob("gpurun_out/r06_job_order/%s_r*.json"%k)):
        try:
            j=json.loads(open(f).read().strip().splitlines()[-1]); v.append((j["value"], round(j["ms_per_step"]*1e3,1), j["roofline"]["avg_launch_us"]))
        except Exception as e: v.append(str(e))
    print(k, v)
PY
