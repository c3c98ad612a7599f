mkdir -p gpurun_out/r2a
(timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "lds_staged" 2>&1 | tail -15) > gpurun_out/r2a/t_lds.log
(timeout 900 python -m pytest tests/test_model_gpu.py -q -k "true_shape or general_path or cfg2 or cfg3" -s 2>&1 | tail -40) > gpurun_out/r2a/t_model.log
(timeout 600 python tools/bench_gemm2.py 2>&1 | tail -150) > gpurun_out/r2a/bench196.log
(timeout 600 python tools/bench_gemm2.py --big 2>&1 | tail -80) > gpurun_out/r2a/benchbig.log
tail -5 gpurun_out/r2a/t_lds.log gpurun_out/r2a/t_model.log
