O=gpurun_out/r05; mkdir -p $O
python tools/predict_scaling.py --ranks 2,4,8 --cells 8,16 --md $O/pred2_f32_hash.md > $O/pred2_f32_hash.txt 2>&1
python tools/predict_scaling.py --owner cyclic --ranks 2,4,8 --cells 8,16 --md $O/pred2_f32_cyclic.md > $O/pred2_f32_cyclic.txt 2>&1
python tools/predict_scaling.py --int16 --ranks 2,4,8 --cells 8 --md $O/pred2_i16_hash.md > $O/pred2_i16_hash.txt 2>&1
python tools/predict_scaling.py --int16 --owner cyclic --ranks 8 --cells 8,16 --md $O/pred2_i16_cyclic.md > $O/pred2_i16_cyclic.txt 2>&1
cat $O/pred2_f32_hash.md $O/pred2_f32_cyclic.md $O/pred2_i16_hash.md $O/pred2_i16_cyclic.md
python tools/cull_soak.py 36 0 --steep 12 --margins "-16,0;-32,0;-64,0;-128,0;0,-0.03;0,-0.1;0,-0.3" > $O/cull_margins_neg2.txt 2>&1; tail -11 $O/cull_margins_neg2.txt
