#!/bin/bash
# NMS kernel iteration: post-processing parity tests, phase stamps, throughput
OUT=gpurun_out/r4nms_$1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_filters.py -m gpu -q -x > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
timeout 200 python tools/nms_probe.py > $OUT/nms_probe.txt 2>&1; tail -6 $OUT/nms_probe.txt
timeout 200 python tools/nms_trained_probe.py > $OUT/nms_trained_probe.txt 2>&1; tail -7 $OUT/nms_trained_probe.txt
timeout 200 python tools/stage_table.py --throughput > $OUT/stage_default.txt 2>&1; tail -4 $OUT/stage_default.txt
