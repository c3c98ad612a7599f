mkdir -p gpurun_out/r2d
(timeout 300 python tools/trace_gemm.py 0 2>&1 | grep "wg\|==" | cut -c1-300) > gpurun_out/r2d/trace0.log
