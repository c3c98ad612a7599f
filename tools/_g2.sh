mkdir -p gpurun_out/r2b
(timeout 300 tools/ubench/stream.bin 2>&1) > gpurun_out/r2b/stream.log
(timeout 300 python tools/bench_gemm2.py --mb 16 2>&1 | grep -E "tile|val fc1|val proj|dec qkv|dec fc2" | grep -E "tile| 0  |11  " ) > gpurun_out/r2b/bench196_warm.log
tail -3 gpurun_out/r2b/stream.log
