mkdir -p gpurun_out/r2w
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r2w/gpu_tests.log 2>&1
tail -4 gpurun_out/r2w/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2w/smoke.log 2>&1
tail -2 gpurun_out/r2w/smoke.log
timeout 600 python demo_loop.py --tiny --frames 5 --save_ply gpurun_out/r2w/demo.ply > gpurun_out/r2w/demo.log 2>&1
tail -6 gpurun_out/r2w/demo.log
