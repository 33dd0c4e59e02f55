O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
B="python3 bench.py --no-cpu --steps 200 --warmup 20"
for nu in 0 1; do
  if [ $nu = 1 ]; then export PF_NO_UPPER=1; else unset PF_NO_UPPER; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tail_nu${nu}_fetch -o f -- $B > $O/tail_nu${nu}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/tail_nu${nu}_write -o w -- $B > $O/tail_nu${nu}_write.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tail_nu${nu}_stats -o s -- $B > $O/tail_nu${nu}_stats.log 2>&1
  echo "NO_UPPER=$nu"; python3 tools/pmc_summary.py traffic $O/tail_nu${nu}_fetch $O/tail_nu${nu}_write f32 2>&1 | tail -2
  python3 tools/pmc_summary.py stats $O/tail_nu${nu}_stats | tail -1
done > $O/tail_traffic.txt 2>&1
cat $O/tail_traffic.txt
