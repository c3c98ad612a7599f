mkdir -p gpurun_out/r2y
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r2y/kt -o p -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-profile --no-extras > $R/gpurun_out/r2y/kt.log 2>&1
cd $R
python tools/rocpd_lastseq.py gpurun_out/r2y/kt/p_results.db 24.0 > gpurun_out/r2y/lastseq.txt 2>&1
rm -rf gpurun_out/r2y/kt
head -45 gpurun_out/r2y/lastseq.txt
