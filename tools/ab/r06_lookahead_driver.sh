out=gpurun_out/r06_lookahead2; mkdir -p $out
for rep in 1 2 3; do for la in 0 8 4 2 12; do python3 bench.py --no-cpu --lookahead $la --steps 20 --warmup 5 > $out/drv_la${la}_r$rep.json 2>/dev/null; done; done
python3 - <<'PY'
import json,glob
for la in (0,2,4,8,12):
    v=[]
    for f in sorted(glob.glob("gpurun_out/r06_lookahead2/drv_la%d_r*.json"%la)):
        j=json.loads(open(f).read().strip().splitlines()[-1]); v.append((j["value"], j["roofline"]["avg_launch_us"]))
    print("drv", la, v)
PY
