#!/bin/bash
OUT=gpurun_out/r4e; mkdir -p $OUT
for cfg in "9 0" "12 0" "9 1"; do set -- $cfg
  echo "== float form up to block $1, Conv_1 split weights $2" >> $OUT/robust_variants.txt
  WZ_FLOAT_UPTO=$1 WZ_CONV1_SPLIT=$2 NFRAMES=6 timeout 400 python tools/robust_check.py 0 1.0 1.5 2.0 2>&1 | grep -a "^spread" >> $OUT/robust_variants.txt
  WZ_FLOAT_UPTO=$1 WZ_CONV1_SPLIT=$2 timeout 300 python tools/stage_table.py --robust --throughput 2>&1 | tail -1 >> $OUT/robust_variants.txt
done
cat $OUT/robust_variants.txt
