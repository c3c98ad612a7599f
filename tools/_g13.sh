mkdir -p gpurun_out/r2l
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r2l/t_all.log
(timeout 900 python bench.py 2>&1 | tail -2) > gpurun_out/r2l/bench.log
(timeout 900 python bench.py --gpus 2 2>&1 | tail -3; echo "rc=$?") > gpurun_out/r2l/bench_gpus2.log
