mkdir -p gpurun_out/r2g
(timeout 600 python tools/bench_gemm2.py --tiles 0,18,13 2>&1 | grep -v amdgpu | tail -70) > gpurun_out/r2g/tiles18.log
(timeout 300 python tools/trace_gemm.py 18 2>&1 | grep "wg\|==" | cut -c1-200) > gpurun_out/r2g/trace18.log
