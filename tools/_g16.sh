mkdir -p gpurun_out/r2v
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2v/kt -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-extras > $R/gpurun_out/r2v/kt.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/r2v/kt/p_results.db > gpurun_out/r2v/stats.md 2>&1
python tools/rocpd_gaps.py gpurun_out/r2v/kt/p_results.db 0.3 > gpurun_out/r2v/gaps.txt 2>&1
python tools/rocpd_timeline.py gpurun_out/r2v/kt/p_results.db 0.9 3.0 > gpurun_out/r2v/timeline.txt 2>&1
rm -rf gpurun_out/r2v/kt
tail -1 gpurun_out/r2v/kt.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/r2v/bench.log 2>&1
tail -1 gpurun_out/r2v/bench.log | cut -c1-300
