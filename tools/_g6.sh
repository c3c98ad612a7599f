mkdir -p gpurun_out/r2f
(timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/r2f/t_ops.log
(timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r2f/t_model.log
(timeout 300 python tools/trace_gemm.py 13 2>&1 | grep "wg\|==" | cut -c1-300) > gpurun_out/r2f/trace13.log
(timeout 600 python tools/bench_gemm2.py 2>&1 | tail -150) > gpurun_out/r2f/bench196.log
(timeout 600 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -3) > gpurun_out/r2f/bench.log
