mkdir -p gpurun_out/r2d
for t in 16 17; do (timeout 300 python tools/trace_gemm.py $t 2>&1 | grep "wg\|==" | cut -c1-300) > gpurun_out/r2d/trace$t.log; done
(timeout 600 python tools/bench_gemm2.py 2>&1 | tail -150) > gpurun_out/r2d/bench196.log
