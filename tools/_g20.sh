mkdir -p gpurun_out/r2r
python tools/bench_gemm2.py --big --tiles 1,6,20,21 --only "c3 dec fc2,c3 dec proj,c3 val proj,c3 val fc2,c3 key 2,enc proj,enc fc2" > gpurun_out/r2r/tiles_c3.log 2>&1
cat gpurun_out/r2r/tiles_c3.log
