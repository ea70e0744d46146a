mkdir -p gpurun_out/r2h
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "other_weights" > gpurun_out/r2h/pytest.log 2>&1; tail -5 gpurun_out/r2h/pytest.log
