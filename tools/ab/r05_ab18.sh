O=gpurun_out/r05; mkdir -p $O
A=$PWD/tools/ab
(timeout -k 10 300 env PF_LIB=$A/lib_sched_max-ilp.so python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1_plumbing or perspective_and_spread or full_size_frame") > $O/t19.log 2>&1; tail -1 $O/t19.log
tools/abn.sh 3 "" "PF_X=1" "PF_LIB=$A/lib_sched_max-ilp.so" "PF_LIB=$A/lib_sched_iterative-ilp.so" "PF_LIB=$A/lib_sched_iterative-maxocc.so" "PF_LIB=$A/lib_sched_max-memory-clause.so" > $O/ab18_f32.txt 2>&1
tools/abn.sh 3 "--int16" "PF_X=1" "PF_LIB=$A/lib_sched_max-ilp.so" "PF_LIB=$A/lib_sched_iterative-maxocc.so" "PF_LIB=$A/lib_sched_max-memory-clause.so" > $O/ab18_i16.txt 2>&1
cat $O/ab18_f32.txt $O/ab18_i16.txt
