O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
(time timeout -k 10 1000 python -m pytest tests -m gpu -x -q) > $O/t18.log 2>&1; grep -E "passed|failed" $O/t18.log
PF_GIT_SHA=6293fac bash tools/profile_round.sh r05 > $O/profile_round2.log 2>&1; tail -2 $O/profile_round2.log
PF_GIT_SHA=6293fac bash tools/profile_windows.sh r05w '20 5 15' '200 20 0' '20 5 15 nocull' > $O/prof_windows3.log 2>&1; tail -3 $O/prof_windows3.log
