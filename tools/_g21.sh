mkdir -p gpurun_out/r2t
for v in 1 0 1 0; do
SP3_DEC_STREAMS=$v timeout 600 python bench.py --precision fp32 --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r2t/fp32_$v.log 2>&1
echo "dec_streams=$v" $(tail -1 gpurun_out/r2t/fp32_$v.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(round(d['value'],1))
")
done
