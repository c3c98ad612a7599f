mkdir -p gpurun_out/r2z
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "f32x3 or gemm" > gpurun_out/r2z/t.log 2>&1; tail -3 gpurun_out/r2z/t.log
