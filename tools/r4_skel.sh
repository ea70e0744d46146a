#!/bin/bash
# the split-block kernel with its arithmetic removed (-DWZ_HP_SKELETON=1) beside the real one: per-block times, single bracket and 8 launches per bracket
OUT=gpurun_out/r4skel; mkdir -p $OUT
for prog in "" "--robust"; do
  for inner in 1 8; do
    echo "== real ${prog:-default} inner=$inner" >> $OUT/skel.txt
    python tools/stage_table.py $prog --inner $inner --only expanded_conv 2>&1 | grep -v amdgpu >> $OUT/skel.txt
    echo "== skeleton ${prog:-default} inner=$inner" >> $OUT/skel.txt
    WATSOR_HIP_DEV_LIBRARY=$PWD/gpurun_tmp_skel.so python tools/stage_table.py $prog --inner $inner --only expanded_conv 2>&1 | grep -v amdgpu >> $OUT/skel.txt
  done
done
echo "== throughput with the skeleton kernels (results meaningless, timing only)" >> $OUT/skel.txt
WATSOR_HIP_DEV_LIBRARY=$PWD/gpurun_tmp_skel.so python tools/stage_table.py --throughput 2>&1 | tail -2 >> $OUT/skel.txt
WATSOR_HIP_DEV_LIBRARY=$PWD/gpurun_tmp_skel.so python tools/stage_table.py --robust --throughput 2>&1 | tail -2 >> $OUT/skel.txt
cat $OUT/skel.txt
