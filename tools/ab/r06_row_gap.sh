#!/bin/bash
# round 6: level-0 block rows nothing rendered depends on are not launched (LevelArgs::row_gap) -- A/B in the experiments library, interleaved
# (the switch PF_NO_ROW_GAP lived in the change this script measured; the change was reverted -- profiles/r06_ab.md (e))
out=gpurun_out/r06_row_gap; mkdir -p $out
export PF_LIB=pi-slam-fusion_amd/libpifusion_exp.so
for rep in 1 2 3; do
  python3 bench.py --no-cpu > $out/gap_f32_r$rep.json 2>/dev/null
  PF_NO_ROW_GAP=1 python3 bench.py --no-cpu > $out/nogap_f32_r$rep.json 2>/dev/null
  python3 bench.py --no-cpu --int16 > $out/gap_i16_r$rep.json 2>/dev/null
  PF_NO_ROW_GAP=1 python3 bench.py --no-cpu --int16 > $out/nogap_i16_r$rep.json 2>/dev/null
  python3 bench.py --no-cpu --steps 20 --warmup 5 > $out/gap_drv_r$rep.json 2>/dev/null
  PF_NO_ROW_GAP=1 python3 bench.py --no-cpu --steps 20 --warmup 5 > $out/nogap_drv_r$rep.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for k in ("gap_f32","nogap_f32","gap_i16","nogap_i16","gap_drv","nogap_drv"):
    v=[]
    for f in sorted(glob.glob("gpurun_out/r06_row_gap/%s_r*.json"%k)):
        try:
            j=json.loads(open(f).read().strip().splitlines()[-1]); v.append((j["value"], round(j["ms_per_step"]*1e3,1), j["roofline"]["avg_launch_us"], j["roofline"]["frac"]))
        except Exception as e: v.append(str(e))
    print(k, v)
PY
