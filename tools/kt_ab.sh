#!/bin/bash
# A/B of one environment knob under the rocprofv3 kernel trace (single lane): per-kernel average durations side by side.
#   tools/kt_ab.sh <tag> VAR a b      -> gpurun_out/<tag>/kernel_stats_VAR_{a,b}.txt
set -u
TAG=$1; VAR=$2; A=$3; B=$4
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
BENCH="python $ROOT/bench.py --steps 40 --warmup 10 --rounds 3 --no-cpu-baseline --no-legs --no-fp32-leg --no-parity --no-live-pmc"
cd /tmp
for V in $A $B; do
  rm -rf /tmp/wzprof_$V && mkdir -p /tmp/wzprof_$V
  env $VAR=$V WZ_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/wzprof_$V -o kt -- $BENCH > "$OUT/bench_${VAR}_$V.json" 2> "$OUT/kt_${VAR}_$V.err"
  DB=$(find /tmp/wzprof_$V -name '*.db' | head -1)
  [ -n "$DB" ] && python "$ROOT/tools/prof_summary.py" "$DB" "$OUT/kernel_stats_${VAR}_$V.txt" "$OUT/rocprof_kernel_avg_${VAR}_$V.json" > /dev/null
done
