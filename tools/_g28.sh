mkdir -p gpurun_out/r2z
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > gpurun_out/r2z/t.log 2>&1; tail -3 gpurun_out/r2z/t.log
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -s -k "tiny_forward or cfg2" > gpurun_out/r2z/m.log 2>&1; grep -n "passed\|failed\|f32x3" gpurun_out/r2z/m.log | head
timeout 600 python bench.py --precision f32x3 --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r2z/x3.log 2>&1
tail -1 gpurun_out/r2z/x3.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('f32x3', round(d['value'],1))"
