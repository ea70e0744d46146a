# round 5: the SSD heads on the three-tile build (WZ_WIDE_NTW=3: 256 registers, two workgroups per CU) against the five-tile one (344 registers, one per CU)
mkdir -p gpurun_out/r05f
WZ_WIDE_NTW=3 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "wide_head or grouped_head or tensor_by_tensor or scores_within" > gpurun_out/r05f/pytest_ntw3.txt 2>&1; tail -3 gpurun_out/r05f/pytest_ntw3.txt
for v in "WZ_WIDE_NTW=5" "WZ_WIDE_NTW=3" "WZ_WIDE_NTW=5" "WZ_WIDE_NTW=3" "WZ_WIDE_NTW=3 WZ_WIDE_CUS=96" "WZ_WIDE_NTW=3 WZ_WIDE_CUS=192" "WZ_WIDE_NTW=3 WZ_WIDE_CUS=256"; do
  echo "== $v"
  env $v timeout 150 python tools/stage_table.py --robust --throughput --only heads 2>&1 | grep -E "heads|throughput"
done > gpurun_out/r05f/variants.txt 2>&1
cat gpurun_out/r05f/variants.txt
WZ_WIDE_NTW=3 timeout 200 python tools/lane_overlap.py --no-product --out gpurun_out/r05f/lane_overlap_ntw3.txt > /dev/null 2>&1; grep -E "^run|KERNELS|CU-SLOT|wide_group|reduce_group|^sum" gpurun_out/r05f/lane_overlap_ntw3.txt
