mkdir -p gpurun_out/r2o
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3) > gpurun_out/r2o/t_all.log
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/r2o/bench.log
