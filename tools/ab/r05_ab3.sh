O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so; R=$PWD/tools/ab/lib_r04.so
(time timeout -k 10 900 python -m pytest tests -m gpu -x -q) > $O/t3.log 2>&1; grep -E "passed|failed" $O/t3.log
tools/abn.sh 3 "" "PF_LIB=$R" "PF_X=1" "PF_LIB=$E PF_A_ILP=0" > $O/ab3_f32.txt 2>&1
tools/abn.sh 3 "--int16" "PF_LIB=$R" "PF_X=1" "PF_LIB=$E PF_A_ILP=0" > $O/ab3_i16.txt 2>&1
cat $O/ab3_f32.txt $O/ab3_i16.txt
PF_LIB=$E PF_STAMP=1 python tools/stamp_phases.py > $O/stamps4_f32.txt 2>&1; grep -A7 "job 0: pro" $O/stamps4_f32.txt
