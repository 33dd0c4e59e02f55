O=gpurun_out/r05; mkdir -p $O
python tools/cull_soak.py 400 1000 --steep 80 > $O/cull_soak_final3.txt 2>&1; tail -3 $O/cull_soak_final3.txt
python tools/fuzz_soak.py 300 > $O/fuzz_soak_final3.txt 2>&1; tail -2 $O/fuzz_soak_final3.txt
