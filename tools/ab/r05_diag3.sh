O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so
tools/abn.sh 2 "" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "PF_X=1" "PF_TABLE_COPY=1" > $O/ab3_kernarg.txt 2>&1
for v in "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "PF_TABLE_COPY=1"; do
  echo "$v NO_UPPER ABLATE=3"; env $v PF_NO_UPPER=1 PF_LIB=$E PF_A_ILP=3 PF_ABLATE=3 python tools/kprof.py 2>/dev/null | grep -E "level0_fused|wall"
done >> $O/ab3_kernarg.txt 2>&1
cat $O/ab3_kernarg.txt
