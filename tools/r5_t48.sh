# round 5: the lean 4 x 8 builds of the stride-1 10x10 blocks (WZ_HP_T48 = 1: eight waves at two per SIMD; 2: four waves at one per SIMD) against the 4 x 4 builds
mkdir -p gpurun_out/r05d
for t in 1 2; do
WZ_HP_T48=$t timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "tensor_by_tensor and robust" > gpurun_out/r05d/pytest_t48_$t.txt 2>&1; tail -3 gpurun_out/r05d/pytest_t48_$t.txt
done
for v in "WZ_HP_T48=0" "WZ_HP_T48=2" "WZ_HP_T48=1" "WZ_HP_T48=0" "WZ_HP_T48=2" "WZ_HP_T48=2 WZ_HP_CG_CAP=64" "WZ_HP_T48=2 WZ_HP_CG_CAP=256"; do
  echo "== $v"
  env $v timeout 150 python tools/stage_table.py --robust --throughput --only expanded_conv_1 2>&1 | grep -E "conv_1[3-6]|throughput"
done > gpurun_out/r05d/variants2.txt 2>&1
cat gpurun_out/r05d/variants2.txt
