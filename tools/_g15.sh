mkdir -p gpurun_out/r2n
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/r2n/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-extras --no-graphs > $GRAFT_REPO_ROOT/gpurun_out/r2n/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do python tools/rocpd_pmc.py gpurun_out/r2n/pmc_$c/p_results.db --json gpurun_out/r2n/pmc_$c.json > gpurun_out/r2n/pmc_$c.md; rm -rf gpurun_out/r2n/pmc_$c; done
head -8 gpurun_out/r2n/pmc_FETCH_SIZE.md
