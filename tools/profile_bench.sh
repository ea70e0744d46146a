#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box: kernel trace + stats, then the PMC passes (each on its own, as gpurun requires).
#   tools/profile_bench.sh <tag> [extra bench args]      -> gpurun_out/<tag>/{kernel_stats.txt,rocprof_kernel_avg.json,pmc_per_kernel.txt,pmc_traffic.json}
set -u
TAG=${1:-prof}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 40 --warmup 10 --rounds 3 --no-cpu-baseline --no-legs --no-fp32-leg --no-parity --no-live-pmc $*"
cd /tmp
rm -rf /tmp/wzprof && mkdir -p /tmp/wzprof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/wzprof/kt -o kt -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
DB=$(find /tmp/wzprof/kt -name '*.db' | head -1)
[ -n "$DB" ] && python "$OLDPWD/tools/prof_summary.py" "$DB" "$OUT/kernel_stats.txt" "$OUT/rocprof_kernel_avg.json" > /dev/null
DBS=""
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $C -d /tmp/wzprof/$N -o $N -- $BENCH > /dev/null 2> "$OUT/pmc_$N.err"
  D=$(find /tmp/wzprof/$N -name '*.db' | head -1)
  [ -n "$D" ] && DBS="$DBS $D"
done
[ -n "$DBS" ] && python "$OLDPWD/tools/pmc_summary.py" $DBS --out "$OUT/pmc_per_kernel.txt" --json "$OUT/pmc_traffic.json" > /dev/null
ls -la "$OUT"
