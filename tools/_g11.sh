mkdir -p gpurun_out/r2j
(timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "softmax or bank_write or cos_sim" 2>&1 | tail -30) > gpurun_out/r2j/t_mem.log
(timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -30) > gpurun_out/r2j/t_model.log
(timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/r2j/bench.log
