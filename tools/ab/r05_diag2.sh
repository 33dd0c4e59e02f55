O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so
for nu in 0 1; do for ab in 0 2 1 3; do
  echo "NO_UPPER=$nu ABLATE=$ab ILP3"
  if [ $nu = 1 ]; then export PF_NO_UPPER=1; else unset PF_NO_UPPER; fi
  PF_LIB=$E PF_A_ILP=3 PF_ABLATE=$ab python tools/kprof.py 2>/dev/null | grep -E "level0_fused|wall"
done; done > $O/ablate2_old_f32.txt 2>&1
unset PF_NO_UPPER
for ab in 0 2 1 3; do echo "CULL=0 ABLATE=$ab ILP3"; PF_CULL=0 PF_LIB=$E PF_A_ILP=3 PF_ABLATE=$ab python tools/kprof.py 2>/dev/null | grep -E "level0_fused|wall"; done > $O/ablate2_nocull_f32.txt 2>&1
cat $O/ablate2_old_f32.txt $O/ablate2_nocull_f32.txt
