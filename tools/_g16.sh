mkdir -p gpurun_out/r2p
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2p/kt -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-extras > $R/gpurun_out/r2p/kt.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/r2p/kt/p_results.db > gpurun_out/r2p/stats.md 2>&1
python tools/rocpd_gaps.py gpurun_out/r2p/kt/p_results.db 0.3 > gpurun_out/r2p/gaps.txt 2>&1
python tools/rocpd_timeline.py gpurun_out/r2p/kt/p_results.db 0.9 6.0 > gpurun_out/r2p/timeline.txt 2>&1
rm -rf gpurun_out/r2p/kt
tail -2 gpurun_out/r2p/kt.log
head -12 gpurun_out/r2p/gaps.txt
