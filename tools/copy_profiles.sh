#!/bin/bash
# Copies what tools/final_round.sh <tag> left under gpurun_out/ (after a gpurun call) over the round's profile set profiles/<set>_*.
#   tools/copy_profiles.sh <tag> <set>      e.g. tools/copy_profiles.sh r04zz3 r04zz
S=gpurun_out/$1; P=profiles; T=$2
cp $S/bench_b8.json $P/${T}_bench_b8.json; cp $S/stage_table_b8.json $P/${T}_stage_table_b8.json
cp $S/stage_table_robust.txt $P/${T}_stage_table_robust.txt; cp $S/stage_table_default.txt $P/${T}_stage_table_default.txt
cp $S/batch_sweep.json $P/${T}_batch_sweep.json; cp $S/batch_sweep.txt $P/${T}_batch_sweep.txt; cp $S/batch_sweep_default_program.txt $P/${T}_batch_sweep_default_program.txt
cp $S/nms_probe.txt $P/${T}_nms_probe.txt; cp $S/parity_report_robust.txt $P/${T}_parity_report_robust.txt; cp $S/pytest_gpu.txt $P/${T}_pytest_gpu.txt; cp $S/smoke.txt $P/${T}_smoke.txt
for L in 1 4; do
  cp gpurun_out/$1_lanes$L/kernel_stats.txt $P/${T}_kernel_stats_bench_b8_lanes$L.txt; cp gpurun_out/$1_lanes$L/pmc_per_kernel.txt $P/${T}_pmc_per_kernel_lanes$L.txt
  cp gpurun_out/$1_lanes$L/rocprof_kernel_avg.json $P/${T}_rocprof_kernel_avg_lanes$L.json
done
cp gpurun_out/$1_lanes1/rocprof_kernel_avg.json $P/rocprof_kernel_avg.json; cp gpurun_out/$1_lanes1/pmc_traffic.json $P/pmc_traffic.json
# round 5
cp $S/lane_overlap_robust.txt $P/${T}_lane_overlap_robust.txt; cp $S/lane_overlap_default.txt $P/${T}_lane_overlap_default.txt; cp $S/lane_overlap_robust_one_lane.txt $P/${T}_lane_overlap_robust_one_lane.txt
cp $S/lane_overlap_robust.json $P/lane_overlap.json; cp $S/cu_stream.txt $P/${T}_cu_stream.txt; cp $S/soak.txt $P/${T}_soak.txt
# round 6
cp $S/boundary_microbench.txt $P/${T}_boundary_microbench.txt; cp $S/boundary_in_engine_graph.txt $P/${T}_boundary_in_engine_graph.txt; cp $S/boundary_in_engine_eager.txt $P/${T}_boundary_in_engine_eager.txt
cp $S/stage_table_robust_b1.txt $P/${T}_stage_table_robust_b1.txt
