O=gpurun_out/r6_wide_cus.txt; : > $O
for rep in 1 2; do for C in 128 96 160 192 256 64; do
  echo "== WZ_WIDE_CUS=$C" >> $O
  WZ_WIDE_CUS=$C python tools/stage_table.py --robust --throughput --only "heads" 2>&1 | grep -E "heads|^throughput" >> $O
done; done
