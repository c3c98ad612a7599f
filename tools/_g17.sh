mkdir -p gpurun_out/r2q
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r2q/model.log 2>&1
grep -n "passed\|failed" gpurun_out/r2q/model.log; grep -n "Error\|error" gpurun_out/r2q/model.log | head -5
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2q/bench.log 2>&1
tail -1 gpurun_out/r2q/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(d['value']); 
"
