# round 5: the per-CU weight stream (tools/micro/cu_stream.hip), lanes / streams variants of the headline loop, and the GPU suite
mkdir -p gpurun_out/r05e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/micro/cu_stream.hip -o /tmp/cu_stream && timeout 60 /tmp/cu_stream > gpurun_out/r05e/cu_stream.txt 2>&1; cat gpurun_out/r05e/cu_stream.txt
for v in "WZ_LANES=4" "WZ_LANES=8 WZ_STREAMS=4" "WZ_LANES=6 WZ_STREAMS=3" "WZ_LANES=3" "WZ_LANES=5 WZ_STREAMS=5" "WZ_LANES=4"; do
  echo "== $v"
  env $v timeout 150 python tools/stage_table.py --robust --throughput --only NOTHING 2>&1 | grep -E "throughput"
done > gpurun_out/r05e/lanes.txt 2>&1
cat gpurun_out/r05e/lanes.txt
timeout 700 python -m pytest tests -m gpu -q -rf > gpurun_out/r05e/pytest_gpu.txt 2>&1; tail -8 gpurun_out/r05e/pytest_gpu.txt
