O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so
PF_LIB=$E PF_STAMP=1 python tools/stamp_phases.py > $O/stamps2_new_f32.txt 2>&1
PF_LIB=$E PF_STAMP=1 python tools/stamp_phases.py --int16 > $O/stamps2_new_i16.txt 2>&1
cat $O/stamps2_new_f32.txt
