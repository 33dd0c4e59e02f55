O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so; R=$PWD/tools/ab/lib_r04.so
(timeout -k 10 900 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_dist.py tests/test_gpu_cull.py tests/test_gpu_at_size.py -x -q) > $O/t6.log 2>&1; tail -1 $O/t6.log
tools/abn.sh 3 "" "PF_LIB=$R" "PF_X=1" "PF_LIB=$E" "PF_LIB=$E PF_NO_COMPACT=1" > $O/ab6_f32.txt 2>&1
cat $O/ab6_f32.txt
python tools/predict_scaling.py --ranks 2,4,8 --cells 8 --md $O/pred_f32_compact.md > $O/pred_f32_compact.txt 2>&1; cat $O/pred_f32_compact.md
PF_LIB=$E PF_NO_COMPACT=1 python tools/predict_scaling.py --ranks 8 --cells 8 --md $O/pred_f32_nocompact.md > $O/pred_f32_nocompact.txt 2>&1; cat $O/pred_f32_nocompact.md
