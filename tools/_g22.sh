mkdir -p gpurun_out/r2t
timeout 600 python bench.py --precision f32x3 --steps 4 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2t/x3_prof.log 2>&1
tail -1 gpurun_out/r2t/x3_prof.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(d['value'], d.get('profiled_sequence_kernel_ms'))
for k in d['kernel_breakdown']: print(k)
"
