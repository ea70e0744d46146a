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
