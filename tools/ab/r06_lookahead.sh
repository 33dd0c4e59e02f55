#!/bin/bash
# round 6: the cull's lookahead -- bench.py's cfg-A window (200 after 20, and the driver's 20 after 5) at lookahead 0 / 1 / 2 / 4 / 8 / 20, interleaved on one box
out=gpurun_out/r06_lookahead; mkdir -p $out
for rep in 1 2; do
  for la in 0 48 4 8 20 32; do
    python3 bench.py --no-cpu --lookahead $la --steps 200 --warmup 20 > $out/f32_la${la}_r$rep.json 2>/dev/null
    python3 bench.py --no-cpu --lookahead $la --steps 200 --warmup 20 --int16 > $out/i16_la${la}_r$rep.json 2>/dev/null
    python3 bench.py --no-cpu --lookahead $la --steps 20 --warmup 5 > $out/drv_la${la}_r$rep.json 2>/dev/null
    echo "rep $rep la $la done"
  done
done
python3 - <<'PY'
import json,glob
for kind in ("f32","i16","drv"):
    for la in (0,4,8,20,32,48):
        v=[]
        for f in sorted(glob.glob("gpurun_out/r06_lookahead/%s_la%d_r*.json"%(kind,la))):
            try:
                j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
                v.append((j["value"], r["avg_launch_us"], r["frac"], j["config"]["rendered_share_rank0"], j["config"]["level0_run_share_rank0"], r["alg_bytes_run_per_launch"]/1e6))
            except Exception as e: v.append(("err",str(e)))
        print(kind, la, v)
PY
