#!/bin/bash
# A/B of the fused extras pairs (WZ_EXTRAS_PAIR, dev knob): per-stage kernel times of the extras chain, headline throughput and the single-frame p50.
#   gpurun -- 'bash tools/r6_pairs.sh'   -> gpurun_out/r6_pair_extras.txt
O=gpurun_out/r6_pair_extras.txt; mkdir -p gpurun_out; : > $O
for rep in 1 2; do for P in 1 0; do
  echo "== WZ_EXTRAS_PAIR=$P" >> $O
  WZ_EXTRAS_PAIR=$P python tools/stage_table.py --robust --throughput --only "layer_19" 2>&1 | grep -E "layer_19|^sum|^throughput" >> $O
done; done
for P in 1 0; do WZ_EXTRAS_PAIR=$P python tools/stage_table.py --robust --batch 1 --throughput --only zzz 2>&1 | tail -1 >> $O; done
