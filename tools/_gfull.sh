mkdir -p gpurun_out/r2w
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r2w/gpu_tests.log 2>&1
tail -3 gpurun_out/r2w/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2w/smoke.log 2>&1
tail -2 gpurun_out/r2w/smoke.log
timeout 1200 python bench.py > gpurun_out/r2w/bench.log 2>&1
tail -1 gpurun_out/r2w/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(d['value'], d['roofline']['avg_us'], d['memread']['us'], d['memread']['critical_path']['us'], d['fp32']['value'], d['f32x3']['value'], d['batch4']['value'], d['config3']['value'], d['cpu_baseline']['value'])
"
