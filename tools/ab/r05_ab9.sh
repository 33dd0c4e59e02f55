O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so
(timeout -k 10 600 python -m pytest tests/test_gpu_variants.py -x -q -k "A_ILP=0") > $O/t10.log 2>&1; tail -1 $O/t10.log
tools/abn.sh 3 "--int16" "PF_X=1" "PF_LIB=$E PF_A_ILP=0" > $O/ab9_i16.txt 2>&1
tools/abn.sh 2 "" "PF_X=1" "PF_LIB=$E" > $O/ab9_f32.txt 2>&1
cat $O/ab9_i16.txt $O/ab9_f32.txt
