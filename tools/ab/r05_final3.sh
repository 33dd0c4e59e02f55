O=gpurun_out/r05; mkdir -p $O
(time timeout -k 10 1100 python -m pytest tests -m gpu -x -q) > $O/t11.log 2>&1; grep -E "passed|failed" $O/t11.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
