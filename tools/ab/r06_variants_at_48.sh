out=gpurun_out/r06_variants48; mkdir -p $out
export PF_LIB=pi-slam-fusion_amd/libpifusion_exp.so
for rep in 1 2; do
  python3 bench.py --no-cpu > $out/base_r$rep.json 2>/dev/null
  PF_BLOCK24=1 python3 bench.py --no-cpu > $out/b24_r$rep.json 2>/dev/null
  PF_A_ILP=3 python3 bench.py --no-cpu > $out/ilp3_r$rep.json 2>/dev/null
  PF_BLOCK24=1 python3 bench.py --no-cpu --int16 > $out/b24i16_r$rep.json 2>/dev/null
  python3 bench.py --no-cpu --int16 > $out/basei16_r$rep.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for k in ("base","b24","ilp3","basei16","b24i16"):
    v=[]
    for f in sorted(glob.glob("gpurun_out/r06_variants48/%s_r*.json"%k)):
        try:
            j=json.loads(open(f).read().strip().splitlines()[-1]); v.append((j["value"], round(j["ms_per_step"]*1e3,1), j["roofline"]["avg_launch_us"]))
        except Exception as e: v.append(str(e))
    print(k, v)
PY
