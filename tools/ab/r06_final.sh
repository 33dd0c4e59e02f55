#!/bin/bash
# round 6, final tree: bench lines, PMC windows of the timed launches, kernel stats of the driver's command, host cost.  Run on the GPU box from the repo root.
export TMPDIR=/tmp
out=gpurun_out/r06_final; mkdir -p $out
export PF_GIT_SHA=${PF_GIT_SHA:-unknown}
python3 bench.py --steps 20 --warmup 5 > $out/r06_bench_driver.json 2> $out/bench_driver.err; echo "driver line done"
python3 bench.py > $out/r06_bench_default.json 2> $out/bench_default.err; echo "default line done"
python3 bench.py --int16 --no-cpu > $out/r06_bench_int16.json 2>/dev/null
python3 bench.py --scale 0.5 --no-cpu > $out/r06_bench_scale05.json 2>/dev/null
python3 bench.py --lookahead 0 --no-cpu > $out/r06_bench_lookahead0.json 2>/dev/null; echo "bench lines done"
PF_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $out/r06_default_2ranks_gloo_one_gpu.json 2> $out/bench_2ranks.err; echo "2-rank rehearsal done"
for la in 0 48; do PF_LIB=pi-slam-fusion_amd/libpifusion_exp.so PF_ABLATE=3 LA=$la python3 tools/host_cost.py 2>/dev/null | grep -v "^Map2D" > $out/host_cost_la$la.txt; done; echo "host cost done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/drv_trace -o s -- python3 bench.py --steps 20 --warmup 5 --no-cpu > $out/drv_trace.log 2>&1
python3 tools/pmc_summary.py stats $out/drv_trace > $out/r06_driver_cmd_kernel_stats.md; python3 tools/launch_gaps.py $out/drv_trace >> $out/r06_driver_cmd_kernel_stats.md; echo "driver trace done"
bash tools/profile_windows.sh r06 "20 5 15" "200 20 0" "20 5 15 nocull" > $out/windows.log 2>&1; tail -3 $out/windows.log
