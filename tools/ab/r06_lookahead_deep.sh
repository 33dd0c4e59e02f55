out=gpurun_out/r06_lookahead3; mkdir -p $out
for rep in 1 2; do for la in 4 20 32 48 64; do
  python3 bench.py --no-cpu --lookahead $la --steps 200 --warmup 20 > $out/f32_la${la}_r$rep.json 2>/dev/null
  python3 bench.py --no-cpu --lookahead $la --steps 20 --warmup 5 > $out/drv_la${la}_r$rep.json 2>/dev/null
done; done
python3 tools/ab/r06_lookahead_table.py $out
