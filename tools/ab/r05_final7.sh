O=gpurun_out/r05; mkdir -p $O
tools/abn.sh 5 "--steps 20 --warmup 5" "PF_X=1" "PF_SPIN_SYNC=1" > $O/ab14_spin.txt 2>&1; cat $O/ab14_spin.txt
