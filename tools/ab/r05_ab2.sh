O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so; R=$PWD/tools/ab/lib_r04.so
(timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cull.py tests/test_gpu_steady_state.py -x -q) > $O/t2.log 2>&1; tail -2 $O/t2.log
tools/abn.sh 3 "" "PF_LIB=$R" "PF_X=1" > $O/ab2_f32.txt 2>&1
tools/abn.sh 3 "--int16" "PF_LIB=$R" "PF_X=1" "PF_LIB=$E PF_A_ILP=2" > $O/ab2_i16.txt 2>&1
cat $O/ab2_f32.txt $O/ab2_i16.txt
PF_LIB=$E PF_STAMP=1 python tools/stamp_phases.py 2>&1 | grep -A7 "job 0: pro" > $O/stamps3_new_f32.txt; cat $O/stamps3_new_f32.txt
