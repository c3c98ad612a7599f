mkdir -p gpurun_out/r2g
(timeout 600 python tools/bench_gemm2.py --tiles 0,1,2,3,13 2>&1 | grep -v amdgpu | tail -70) > gpurun_out/r2g/tiles0123.log
