O=gpurun_out/r05; mkdir -p $O
F=$PWD/tools/ab/lib_rowtab.so
(timeout -k 10 300 env PF_LIB=$F python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1_plumbing or perspective_and_spread or full_size_frame") > $O/t17.log 2>&1; tail -1 $O/t17.log
tools/abn.sh 4 "" "PF_X=1" "PF_LIB=$F" > $O/ab16_f32.txt 2>&1
cat $O/ab16_f32.txt
