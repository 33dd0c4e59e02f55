O=gpurun_out/r05; mkdir -p $O
H=$PWD/tools/ab/lib_hybw.so
(timeout -k 10 300 env PF_LIB=$H python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1_plumbing or perspective_and_spread or full_size_frame") > $O/t12.log 2>&1; tail -1 $O/t12.log
tools/abn.sh 3 "" "PF_X=1" "PF_LIB=$H" > $O/ab10_f32.txt 2>&1
cat $O/ab10_f32.txt
