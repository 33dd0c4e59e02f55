O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so; R=$PWD/tools/ab/lib_r04.so
(timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_steady_state.py -x -q) > $O/t4.log 2>&1; tail -1 $O/t4.log
tools/abn.sh 3 "" "PF_LIB=$R" "PF_X=1" "PF_LIB=$E PF_A_ILP=0" > $O/ab4_f32.txt 2>&1
tools/abn.sh 2 "--int16" "PF_LIB=$R" "PF_X=1" > $O/ab4_i16.txt 2>&1
cat $O/ab4_f32.txt $O/ab4_i16.txt
