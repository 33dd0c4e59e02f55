O=gpurun_out/r05; mkdir -p $O
R=$PWD/tools/ab/lib_r04.so
tools/abn.sh 5 "--steps 20 --warmup 5" "PF_LIB=$R" "PF_X=1" > $O/ab13_driver.txt 2>&1; cat $O/ab13_driver.txt
