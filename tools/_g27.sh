mkdir -p gpurun_out/r2z
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > gpurun_out/r2z/t.log 2>&1; tail -3 gpurun_out/r2z/t.log
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r2z/m.log 2>&1; grep -n "passed\|failed" gpurun_out/r2z/m.log
timeout 600 python bench.py --precision fp32 --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r2z/fp32.log 2>&1
tail -1 gpurun_out/r2z/fp32.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('fp32', round(d['value'],1))"
