O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so
(timeout -k 10 300 env PF_LIB=$E PF_ABLATE=8192 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1_plumbing or perspective_and_spread") > $O/t20.log 2>&1; tail -1 $O/t20.log
tools/abn.sh 4 "" "PF_LIB=$E" "PF_LIB=$E PF_ABLATE=8192" > $O/ab19_f32.txt 2>&1
cat $O/ab19_f32.txt
