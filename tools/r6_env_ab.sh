#!/bin/bash
# round 6: runtime settings that could move the in-stream boundary (kernel-argument placement) -- p50 of a lone batch / frames/s / mean gap
for v in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1 WZ_GRAPH=0" "HIP_FORCE_DEV_KERNARG=0 WZ_GRAPH=0"; do
  echo "== $v"
  env $v python tools/stage_table.py --robust --batch 8 --throughput --only zzz 2>&1 | grep -E "throughput|sum"
  env $v WZ_LANES=1 WZ_GRAPH=${WZ_GRAPH:-1} python tools/boundary_in_engine.py 2>&1 | tail -1
done
