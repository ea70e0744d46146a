#!/bin/bash
# round 4, GPU call 1: the suite with the new tests, parity report (box tolerance, unmatched rows), NMS phases, host-read A/B
OUT=gpurun_out/r4a; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -25 $OUT/pytest.txt
timeout 300 python tools/parity_report.py --frames 4 > $OUT/parity_default.txt 2>&1; tail -12 $OUT/parity_default.txt
timeout 300 python tools/parity_report.py --robust --frames 3 > $OUT/parity_robust.txt 2>&1; tail -8 $OUT/parity_robust.txt
timeout 300 python tools/parity_report.py --robust --spread 1.5 --frames 3 > $OUT/parity_robust_spread15.txt 2>&1; tail -8 $OUT/parity_robust_spread15.txt
timeout 200 python tools/nms_probe.py > $OUT/nms_probe.txt 2>&1; tail -18 $OUT/nms_probe.txt
timeout 200 python tools/nms_trained_probe.py > $OUT/nms_trained_probe.txt 2>&1; tail -8 $OUT/nms_trained_probe.txt
timeout 500 python tools/host_read_ab.py > $OUT/host_read_ab.txt 2>&1; cat $OUT/host_read_ab.txt | tail -40
timeout 200 python tools/worker_bench.py 16 2 --workers 2 --gpus 1 > $OUT/two_workers.txt 2>&1; tail -3 $OUT/two_workers.txt
timeout 200 python tools/stage_table.py --throughput > $OUT/stage_default.txt 2>&1; tail -3 $OUT/stage_default.txt
timeout 200 python tools/stage_table.py --robust --throughput > $OUT/stage_robust.txt 2>&1; tail -3 $OUT/stage_robust.txt
