O=gpurun_out/r05; mkdir -p $O
PF_GIT_SHA=$1 bash tools/profile_windows.sh r05 "20 5 15" "200 20 0" "20 5 15 nocull" > $O/prof_windows.log 2>&1; tail -8 $O/prof_windows.log
cat gpurun_out/summary_r05/pmc_traffic.json
python tools/cull_soak.py 36 0 --steep 12 --margins "-0.5,0;-1,0;-2,0;-4,0;-8,0;0,-1e-4;0,-1e-3;0,-1e-2" > $O/cull_margins_neg.txt 2>&1; tail -12 $O/cull_margins_neg.txt
