O=gpurun_out/r05; mkdir -p $O
(time timeout -k 10 1100 python -m pytest tests -m gpu -x -q) > $O/t9.log 2>&1; grep -E "passed|failed" $O/t9.log; tail -3 $O/t9.log
