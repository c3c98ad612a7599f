mkdir -p gpurun_out/r2h
(timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r2h/t_all.log
(timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/r2h/bench.log
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2h/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-extras 2>&1 | tail -2) > $GRAFT_REPO_ROOT/gpurun_out/r2h/prof.log
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/r2h/prof | head -20
