O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so; R=$PWD/tools/ab/lib_r04.so
PF_LIB=$E PF_ABLATE=3 python tools/host_cost.py 2>/dev/null | head -2 > $O/host_cost_new.txt
PF_LIB=$R PF_ABLATE=3 python tools/host_cost.py 2>/dev/null | head -2 > $O/host_cost_r04.txt
cat $O/host_cost_new.txt $O/host_cost_r04.txt | cut -c1-600
tools/abn.sh 4 "--steps 20 --warmup 5" "PF_LIB=$R" "PF_X=1" > $O/ab11_driver.txt 2>&1
tools/abn.sh 4 "--steps 20 --warmup 5 --event-every 0" "PF_LIB=$R" "PF_X=1" > $O/ab11_driver_noev.txt 2>&1
cat $O/ab11_driver.txt; echo; cat $O/ab11_driver_noev.txt
