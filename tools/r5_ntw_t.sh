mkdir -p gpurun_out/r05f
for v in "WZ_WIDE_NTW=3" "WZ_WIDE_NTW=3 WZ_WIDE_T=20" "WZ_WIDE_NTW=3 WZ_WIDE_T=27" "WZ_WIDE_NTW=3 WZ_WIDE_T=36" "WZ_WIDE_NTW=3 WZ_WIDE_T=45" "WZ_WIDE_NTW=3 WZ_WIDE_T=60" "WZ_WIDE_NTW=5"; do
  echo "== $v"
  env $v timeout 150 python tools/stage_table.py --robust --throughput --only heads 2>&1 | grep -E "heads|throughput"
done > gpurun_out/r05f/t_sweep.txt 2>&1
cat gpurun_out/r05f/t_sweep.txt
