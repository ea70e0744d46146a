# round 5: channel groups over workgroups for the 19x19 blocks at one / two frames (WZ_HP_CG19, development knob; default on)
mkdir -p gpurun_out/r05i
for b in 1 2 4; do for v in "WZ_HP_CG19=0" "WZ_HP_CG19=1" "WZ_HP_CG19=0" "WZ_HP_CG19=1"; do
  echo "== batch $b $v"
  env $v timeout 150 python tools/stage_table.py --robust --throughput --batch $b --only expanded_conv 2>&1 | grep -E "conv_([7-9]|1[0-2]) |conv_([7-9]|1[0-2])$|throughput|^sum"
done; done > gpurun_out/r05i/cg19.txt 2>&1
cat gpurun_out/r05i/cg19.txt
timeout 700 python -m pytest tests -m gpu -q -rf > gpurun_out/r05i/pytest_gpu.txt 2>&1; tail -6 gpurun_out/r05i/pytest_gpu.txt
