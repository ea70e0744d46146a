#!/bin/bash
# The measurement pass of a round on the GPU box: GPU tests, smoke, the bench line, rocprofv3 passes (4 lanes and 1 lane), batch sweep.
#   tools/final_round.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 400 python bench.py --table $OUT/stage_table_b8.json > $OUT/bench_b8.json 2> $OUT/bench.err; tail -4 $OUT/bench.err; cut -c1-300 $OUT/bench_b8.json
timeout 700 tools/profile_bench.sh ${TAG}_lanes4 > $OUT/prof4.log 2>&1
WZ_LANES=1 timeout 700 tools/profile_bench.sh ${TAG}_lanes1 > $OUT/prof1.log 2>&1
timeout 300 python tools/batch_sweep.py --out $OUT/batch_sweep.json > $OUT/batch_sweep.txt 2>&1; grep "^batch" $OUT/batch_sweep.txt
WZ_SWEEP_DEFAULT_PROGRAM=1 timeout 300 python tools/batch_sweep.py --batches 8,16,32 --out $OUT/batch_sweep_default_program.json > $OUT/batch_sweep_default_program.txt 2>&1; grep "^batch" $OUT/batch_sweep_default_program.txt
timeout 200 python tools/stage_table.py --robust --throughput > $OUT/stage_table_robust.txt 2>&1; tail -2 $OUT/stage_table_robust.txt
timeout 200 python tools/stage_table.py --throughput > $OUT/stage_table_default.txt 2>&1; tail -2 $OUT/stage_table_default.txt
timeout 200 python tools/nms_probe.py > $OUT/nms_probe.txt 2>&1; tail -2 $OUT/nms_probe.txt
timeout 300 python tools/parity_report.py --robust --frames 4 > $OUT/parity_report_robust.txt 2>&1; tail -4 $OUT/parity_report_robust.txt
# round 5: kernels in flight / CU-slot-time from in-kernel stamps (no profiler), both programs; the per-CU weight stream; the soak
timeout 200 python tools/lane_overlap.py --out $OUT/lane_overlap_robust.txt --json $OUT/lane_overlap_robust.json > /dev/null 2> $OUT/lane_overlap.err; grep -E "KERNELS|CU-SLOT|^run" $OUT/lane_overlap_robust.txt
timeout 200 python tools/lane_overlap.py --default-program --out $OUT/lane_overlap_default.txt --json $OUT/lane_overlap_default.json > /dev/null 2>> $OUT/lane_overlap.err; grep -E "KERNELS|CU-SLOT|^run" $OUT/lane_overlap_default.txt
WZ_LANES=1 timeout 200 python tools/lane_overlap.py --no-product --out $OUT/lane_overlap_robust_one_lane.txt > /dev/null 2>> $OUT/lane_overlap.err; grep -E "KERNELS|CU-SLOT|^run" $OUT/lane_overlap_robust_one_lane.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/micro/cu_stream.hip -o /tmp/cu_stream && timeout 60 /tmp/cu_stream > $OUT/cu_stream.txt 2>&1
timeout 200 python tools/soak.py 40 --robust > $OUT/soak.txt 2>&1; timeout 100 python tools/soak.py 20 --robust --small >> $OUT/soak.txt 2>&1; tail -4 $OUT/soak.txt
# round 6: what a kernel boundary costs -- micro-benchmark and the same measure inside the engine (stamps build), both launch paths
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -std=c++17 tools/micro/boundary.hip -o /tmp/boundary && timeout 120 /tmp/boundary > $OUT/boundary_microbench.txt 2>&1
WZ_LANES=1 WZ_GRAPH=1 timeout 200 python tools/boundary_in_engine.py > $OUT/boundary_in_engine_graph.txt 2>&1; tail -1 $OUT/boundary_in_engine_graph.txt
WZ_LANES=1 WZ_GRAPH=0 timeout 200 python tools/boundary_in_engine.py > $OUT/boundary_in_engine_eager.txt 2>&1; tail -1 $OUT/boundary_in_engine_eager.txt
timeout 200 python tools/stage_table.py --robust --batch 1 --throughput > $OUT/stage_table_robust_b1.txt 2>&1; tail -2 $OUT/stage_table_robust_b1.txt
