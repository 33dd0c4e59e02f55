O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_driver.json 2> $O/r05_bench_driver.err; echo "driver rc $?"
python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err; echo "default rc $?"
python bench.py --int16 --no-cpu > $O/r05_bench_int16.json 2>/dev/null
python bench.py --scale 0.5 --no-cpu > $O/r05_bench_scale05.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver_cmd -o s -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $O/prof_driver_cmd.log 2>&1
python3 tools/pmc_summary.py stats $O/prof_driver_cmd > $O/r05_driver_cmd_kernel_stats.md; python3 tools/launch_gaps.py $O/prof_driver_cmd >> $O/r05_driver_cmd_kernel_stats.md
cat $O/r05_driver_cmd_kernel_stats.md
PF_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $O/r05_default_2ranks_gloo_one_gpu.json 2> $O/r05_2ranks.err; echo "2 ranks rc $?"
PF_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu --shard strong > $O/r05_strong_4ranks_gloo_one_gpu.json 2> $O/r05_4ranks.err; echo "4 ranks rc $?"
python - <<'PY'
import json
for f in ("r05_bench_driver","r05_bench_default","r05_bench_int16","r05_bench_scale05","r05_default_2ranks_gloo_one_gpu","r05_strong_4ranks_gloo_one_gpu"):
    try:
        j=json.load(open("gpurun_out/r05/%s.json"%f)); r=j["roofline"]
        print(f, j["value"], j["ms_per_step"], j["scaling"], {k:r.get(k) for k in ("frac","frac_full_canvas","frac_delivered","avg_launch_us","traffic","window")}, "pmc current:", (r.get("pmc_build") or {}).get("current"))
    except Exception as e: print(f, "ERR", e)
PY
