mkdir -p gpurun_out/r2f
(timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "lds_staged" 2>&1 | tail -40) > gpurun_out/r2f/t_ops.log
