#!/bin/bash
# One rocprofv3 --pmc pass over a short bench run, summarised per kernel:  tools/pmc_pass.sh <tag> "<counters>"
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 20 --warmup 5 --rounds 2 --no-cpu-baseline --no-legs --no-fp32-leg --no-parity --no-live-pmc"
HERE=$PWD; cd /tmp; rm -rf /tmp/wzpmc; mkdir -p /tmp/wzpmc
timeout 300 rocprofv3 --pmc $* -d /tmp/wzpmc/p -o p -- $BENCH > /dev/null 2> $OUT/pmc.err
D=$(find /tmp/wzpmc/p -name '*.db' | head -1)
[ -n "$D" ] && python $HERE/tools/pmc_summary.py $D --out $OUT/pmc.txt > /dev/null
cut -c1-50,63-400 $OUT/pmc.txt | head -14
