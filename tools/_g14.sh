mkdir -p gpurun_out/r2m
(timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "offline or tiny_forward or stats_gather" 2>&1 | tail -30) > gpurun_out/r2m/t_off.log
