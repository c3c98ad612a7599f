mkdir -p gpurun_out/r2e
(timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -8) > gpurun_out/r2e/t_ops.log
(timeout 900 python -m pytest tests/test_model_gpu.py tests/test_rope_cref.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r2e/t_model.log
(timeout 600 python tools/bench_gemm2.py 2>&1 | tail -150) > gpurun_out/r2e/bench196.log
(timeout 600 python tools/bench_gemm2.py --big 2>&1 | tail -80) > gpurun_out/r2e/benchbig.log
(timeout 300 python tools/trace_gemm.py 13 2>&1 | grep "wg\|==" | cut -c1-300) > gpurun_out/r2e/trace13.log
(timeout 600 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -3) > gpurun_out/r2e/bench.log
tail -3 gpurun_out/r2e/t_ops.log gpurun_out/r2e/t_model.log
