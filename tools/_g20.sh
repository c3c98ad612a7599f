mkdir -p gpurun_out/r2r
python tools/bench_gemm2.py --tiles 0,4,18 > gpurun_out/r2r/tiles.log 2>&1
cat gpurun_out/r2r/tiles.log
