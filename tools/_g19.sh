mkdir -p gpurun_out/r2q
for v in "1 1" "0 1" "0 0" "1 1" "0 0"; do set -- $v
SP3_READ_SIDE=$1 SP3_READ_FUSED=$2 timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2q/bench_$1_$2.log 2>&1
echo "side=$1 fused=$2" $(tail -1 gpurun_out/r2q/bench_$1_$2.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(round(d['value'],1), round(d['memread']['us'],2))
")
done
