mkdir -p gpurun_out/r2c
(timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "lds_staged" 2>&1 | tail -15) > gpurun_out/r2c/t_lds.log
(timeout 600 python tools/bench_gemm2.py 2>&1 | tail -150) > gpurun_out/r2c/bench196.log
cat gpurun_out/r2c/t_lds.log | tail -4
