mkdir -p gpurun_out/r2x
for v in 0 256 800 0 256; do
SP3_CONV_WK16=$v timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r2x/b_$v.log 2>&1
echo "wk16_max_rows=$v" $(tail -1 gpurun_out/r2x/b_$v.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(round(d['value'],1))
")
done
SP3_CONV_WK16=256 timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "tiny or cfg2" > gpurun_out/r2x/t.log 2>&1; grep -n "passed\|failed" gpurun_out/r2x/t.log
