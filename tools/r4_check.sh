#!/bin/bash
OUT=gpurun_out/r4chk; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -12 $OUT/bench.err; cut -c1-1500 $OUT/bench.json
