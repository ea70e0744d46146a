#!/bin/bash
# round 4, GPU call 2: robust program v2 (float-form chunk buffer, Conv_1 split weights): accuracy at 0 / 1 / 1.5 / 2 decades, speed
OUT=gpurun_out/r4b; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.txt 2>&1; tail -30 $OUT/pytest.txt
timeout 300 python -m pytest tests/test_gpu_stress.py -m gpu -q -s -k robust > $OUT/stress_robust.txt 2>&1; grep -a "robust program\|passed\|failed" $OUT/stress_robust.txt
timeout 300 python tools/stage_table.py --robust --throughput > $OUT/stage_robust.txt 2>&1; tail -40 $OUT/stage_robust.txt
timeout 200 python tools/stage_table.py --throughput > $OUT/stage_default.txt 2>&1; tail -3 $OUT/stage_default.txt
NFRAMES=3 timeout 400 python tools/robust_check.py 0 1.0 1.5 2.0 > $OUT/robust_check.txt 2>&1; tail -5 $OUT/robust_check.txt
