O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so
tools/abn.sh 4 "" "PF_LIB=$E" "PF_LIB=$E PF_ABLATE=8192" "PF_LIB=$E PF_ABLATE=16384" > $O/ab17_f32.txt 2>&1
cat $O/ab17_f32.txt
