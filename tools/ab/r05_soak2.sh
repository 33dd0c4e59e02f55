O=gpurun_out/r05; mkdir -p $O
python tools/cull_soak.py 260 300 --steep 40 > $O/cull_soak_final2.txt 2>&1; tail -3 $O/cull_soak_final2.txt
python tools/fuzz_soak.py 200 > $O/fuzz_soak_final2.txt 2>&1; tail -2 $O/fuzz_soak_final2.txt
