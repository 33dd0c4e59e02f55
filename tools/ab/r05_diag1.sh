set -x
O=gpurun_out/r05; mkdir -p $O
E=$PWD/pi-slam-fusion_amd/libpifusion_exp.so
(timeout -k 10 600 python -m pytest tests/test_gpu_cull.py -x -q) > $O/t_cull.log 2>&1; tail -3 $O/t_cull.log
./tools/cpp/h2d_probe.bin 40 > $O/h2d_probe.txt 2>&1; cat $O/h2d_probe.txt
PF_LIB=$E PF_STAMP=1 python tools/stamp_phases.py > $O/stamps_new_f32.txt 2>&1
PF_LIB=$E PF_STAMP=1 python tools/stamp_phases.py --int16 > $O/stamps_new_i16.txt 2>&1
for ab in 0 2 1 3; do echo "ABLATE=$ab"; PF_LIB=$E PF_ABLATE=$ab python tools/kprof.py 2>/dev/null | grep -E "level0_fused|wall"; done > $O/ablate_new_f32.txt 2>&1
for ab in 0 2 1 3; do echo "ABLATE=$ab ILP3"; PF_LIB=$E PF_A_ILP=3 PF_ABLATE=$ab python tools/kprof.py 2>/dev/null | grep -E "level0_fused|wall"; done > $O/ablate_old_f32.txt 2>&1
echo NO_UPPER; PF_NO_UPPER=1 python tools/kprof.py 2>/dev/null | grep -E "level0_fused|wall" > $O/noupper_f32.txt
cat $O/stamps_new_f32.txt $O/ablate_new_f32.txt $O/ablate_old_f32.txt $O/noupper_f32.txt
