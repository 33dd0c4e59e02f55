O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so; R=$PWD/tools/ab/lib_r04.so
(time timeout -k 10 900 python -m pytest tests -m gpu -x -q) > $O/t8.log 2>&1; grep -E "passed|failed" $O/t8.log
tools/abn.sh 3 "" "PF_LIB=$R" "PF_X=1" "PF_NO_SEED=1" > $O/ab8_f32.txt 2>&1
tools/abn.sh 2 "--steps 20 --warmup 5" "PF_LIB=$R" "PF_X=1" > $O/ab8_driver.txt 2>&1
cat $O/ab8_f32.txt $O/ab8_driver.txt
