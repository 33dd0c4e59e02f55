O=gpurun_out/r05; mkdir -p $O
(timeout -k 10 600 python -m pytest tests/test_gpu_steady_state.py -x -q -k "run_bytes or cull_leaves") > $O/t16.log 2>&1; tail -5 $O/t16.log
python tools/cull_soak.py 120 100 --steep 20 > $O/cull_soak_final.txt 2>&1; tail -2 $O/cull_soak_final.txt
python tools/fuzz_soak.py 60 > $O/fuzz_soak_final.txt 2>&1; tail -2 $O/fuzz_soak_final.txt
