O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so
for nu in 1 0; do for ab in 3 2 1 0; do
  if [ $nu = 1 ]; then export PF_NO_UPPER=1; else unset PF_NO_UPPER; fi
  PF_LIB=$E PF_A_ILP=3 PF_ABLATE=$ab rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_nu${nu}_ab${ab} -o s -- python3 tools/kprof.py --no-events > $O/tr_nu${nu}_ab${ab}.log 2>&1
  echo "NO_UPPER=$nu ABLATE=$ab: $(python3 tools/pmc_summary.py stats $O/tr_nu${nu}_ab${ab} | grep -E 'k_levels<true, 32, 512, false, 3, false, true|full-size')"
done; done > $O/ablate4_trace.txt 2>&1
cat $O/ablate4_trace.txt
