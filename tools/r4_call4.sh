#!/bin/bash
OUT=gpurun_out/r4d; mkdir -p $OUT
for v in e1d1 e0d1 e1d0 e0d0; do
  echo "== $v" >> $OUT/variants.txt
  WATSOR_HIP_DEV_LIBRARY=$PWD/gpurun_tmp_$v.so NFRAMES=1 timeout 200 python tools/robust_check.py 1.5 >> $OUT/variants.txt 2>&1
  WATSOR_HIP_DEV_LIBRARY=$PWD/gpurun_tmp_$v.so WZ_FLOAT_UPTO=16 NFRAMES=1 timeout 200 python tools/robust_check.py 1.5 >> $OUT/variants.txt 2>&1
done
grep -v amdgpu.ids $OUT/variants.txt | tail -30
