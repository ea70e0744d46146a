#!/bin/bash
# round 4, GPU call 3: robust v2b (float form up to block 12 / 9, linear behind; cheaper encode / decode), Conv_1 K = 640 kernel choice
OUT=gpurun_out/r4c; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_stress.py -m gpu -q -s -k robust > $OUT/stress_robust.txt 2>&1; grep -a "robust program\|passed\|failed\|Error" $OUT/stress_robust.txt
timeout 300 python tools/stage_table.py --robust --throughput > $OUT/stage_robust_f12.txt 2>&1; tail -38 $OUT/stage_robust_f12.txt
WZ_FLOAT_UPTO=9 timeout 300 python tools/stage_table.py --robust --throughput > $OUT/stage_robust_f9.txt 2>&1; tail -2 $OUT/stage_robust_f9.txt
WZ_CONV_WS=0 timeout 300 python tools/stage_table.py --robust --throughput > $OUT/stage_robust_f12_nows.txt 2>&1; grep "Conv_1\|throughput" $OUT/stage_robust_f12_nows.txt
NFRAMES=4 WZ_FLOAT_UPTO=9 timeout 400 python tools/robust_check.py 1.0 1.5 2.0 > $OUT/robust_check_f9.txt 2>&1; tail -3 $OUT/robust_check_f9.txt
NFRAMES=4 timeout 400 python tools/robust_check.py 1.0 1.5 2.0 > $OUT/robust_check_f12.txt 2>&1; tail -3 $OUT/robust_check_f12.txt
timeout 200 python tools/stage_table.py --throughput > $OUT/stage_default.txt 2>&1; tail -2 $OUT/stage_default.txt
