out=gpurun_out/r06_abl; mkdir -p $out
export PF_LIB=pi-slam-fusion_amd/libpifusion_exp.so
for rep in 1 2; do
 python3 bench.py --no-cpu > $out/base_r$rep.json 2>/dev/null
 PF_NO_UPPER=1 python3 bench.py --no-cpu > $out/noupper_r$rep.json 2>/dev/null
 PF_ABLATE=3 python3 bench.py --no-cpu > $out/skel_r$rep.json 2>/dev/null
 PF_ABLATE=2 python3 bench.py --no-cpu > $out/aonly_r$rep.json 2>/dev/null
 PF_ABLATE=1 python3 bench.py --no-cpu > $out/bd_r$rep.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for k in ("base","noupper","skel","aonly","bd"):
    v=[]
    for f in sorted(glob.glob("gpurun_out/r06_abl/%s_r*.json"%k)):
        try:
            j=json.loads(open(f).read().strip().splitlines()[-1]); v.append((j["value"], j["ms_per_step"]*1e3, j["roofline"]["avg_launch_us"]))
        except Exception as e: v.append(str(e))
    print(k, v)
PY
