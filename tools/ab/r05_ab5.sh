O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so; R=$PWD/tools/ab/lib_r04.so
tools/abn.sh 3 "" "PF_LIB=$R" "PF_X=1" > $O/ab5_f32.txt 2>&1
tools/abn.sh 3 "--int16" "PF_LIB=$R" "PF_X=1" "PF_LIB=$E PF_A_ILP=0" > $O/ab5_i16.txt 2>&1
tools/abn.sh 2 "--steps 20 --warmup 5" "PF_LIB=$R" "PF_X=1" > $O/ab5_driver.txt 2>&1
cat $O/ab5_f32.txt $O/ab5_i16.txt $O/ab5_driver.txt
python tools/cull_soak.py 36 0 --steep 12 --margins "0,0;0.25,0;0.5,1e-6;1,1e-5;2,1e-5" > $O/cull_margins.txt 2>&1; tail -8 $O/cull_margins.txt
