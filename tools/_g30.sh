mkdir -p gpurun_out/r2z
for v in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r2z/base_$v.log 2>&1
tail -1 gpurun_out/r2z/base_$v.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(round(d['value'],1))
"
done
rocm-smi --showclocks 2>/dev/null | head -12
