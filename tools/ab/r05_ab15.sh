O=gpurun_out/r05; mkdir -p $O
F=$PWD/tools/ab/lib_fb3.so
(timeout -k 10 300 env PF_LIB=$F python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1_plumbing or perspective_and_spread or full_size_frame") > $O/t15.log 2>&1; tail -1 $O/t15.log
tools/abn.sh 4 "--steps 20 --warmup 5" "PF_X=1" "PF_LIB=$F" > $O/ab15_driver.txt 2>&1
tools/abn.sh 3 "" "PF_X=1" "PF_LIB=$F" > $O/ab15_f32.txt 2>&1
cat $O/ab15_driver.txt $O/ab15_f32.txt
