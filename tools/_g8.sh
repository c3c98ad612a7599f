mkdir -p gpurun_out/r2g
(echo "== no prefetch"; timeout 600 python tools/bench_gemm2.py --tiles 0,13 2>&1 | grep -v amdgpu | grep -E " 0     0|13    0"; echo "== prefetch next"; timeout 600 python tools/bench_gemm2.py --tiles 0,13 --prefetch 1 2>&1 | grep -v amdgpu | grep -E " 0     0|13    0") > gpurun_out/r2g/prefetch.log
(timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm" 2>&1 | tail -3) >> gpurun_out/r2g/prefetch.log
