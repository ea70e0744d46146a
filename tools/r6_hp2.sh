#!/bin/bash
# round 6: the two-launch form of blocks 13 .. 16 -- waves per workgroup of launch A (WZ_HP2_NW1: stride-1 blocks, WZ_HP2_NW2: block 13), tiles of launch B
for v in "WZ_HP2_NW1=3 WZ_HP2_NW2=3" "WZ_HP2_NW1=4 WZ_HP2_NW2=4" "WZ_HP2_NW1=4 WZ_HP2_NW2=3" "WZ_HP2_NW1=6 WZ_HP2_NW2=6" "WZ_HP2_NW1=4 WZ_HP2_NW2=3 WZ_HP2_MT=1"; do
  echo "== $v"
  env $v python tools/stage_table.py --robust --batch 8 --throughput --only expanded_conv_1 2>&1 | grep -E "conv_1[3-6]|throughput|sum"
done
