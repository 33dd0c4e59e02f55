#!/usr/bin/env python3
"""profiles/r06_lookahead.md from the bench lines tools/ab/r06_lookahead.sh leaves under gpurun_out/r06_lookahead/."""
import glob, json, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_lookahead"
rows = {}
for f in sorted(glob.glob(d + "/*_la*_r*.json")):
    kind, la = f.split("/")[-1].split("_")[0], int(f.split("_la")[1].split("_")[0])
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    r = j["roofline"]
    rows.setdefault((kind, la), []).append((j["value"], j["ms_per_step"] * 1e3, r["avg_launch_us"], r["frac"], j["config"]["rendered_share_rank0"],
                                            j["config"]["level0_run_share_rank0"], r["alg_bytes_run_per_launch"] / 1e6, j["config"]["culled_tiles_rank0"]))
names = {"f32": "fp32 pyramids, 200 keyframes after 20 (`bench.py --no-cpu --steps 200 --warmup 20 --lookahead N`)",
         "i16": "int16 pyramids, 200 keyframes after 20 (`--int16`)", "drv": "fp32, the driver's window (`--steps 20 --warmup 5`)"}
for kind in ("f32", "i16", "drv"):
    print("\n## %s\n" % names[kind])
    print("| lookahead | keyframes/s (each round) | us per step | launch, us (event-bracketed) | roofline.frac | rendered share | level-0 run share | run bytes per launch, MB | tiles culled whole |")
    print("|---|---|---|---|---|---|---|---|---|")
    for (k, la), v in sorted(rows.items()):
        if k != kind:
            continue
        print("| %d | %s | %s | %s | %s | %.3f | %.3f | %.1f | %d |" % (la, " / ".join("%.0f" % x[0] for x in v), " / ".join("%.1f" % x[1] for x in v),
              " / ".join("%.1f" % x[2] for x in v), " / ".join("%.3f" % x[3] for x in v), v[0][4], v[0][5], v[0][6], v[0][7]))
