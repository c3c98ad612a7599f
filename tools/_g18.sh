mkdir -p gpurun_out/r2q
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "two_launch" > gpurun_out/r2q/ops.log 2>&1
tail -5 gpurun_out/r2q/ops.log
python tools/bench_memread.py > gpurun_out/r2q/memread.log 2>&1
cat gpurun_out/r2q/memread.log
