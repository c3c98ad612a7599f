mkdir -p gpurun_out/r2u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/r2u/pmc_$c -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-extras --no-graphs > $R/gpurun_out/r2u/pmc_$c.log 2>&1
done
cd $R
for c in FETCH_SIZE WRITE_SIZE; do python tools/rocpd_pmc_grid.py gpurun_out/r2u/pmc_$c/p_results.db --json gpurun_out/r2u/pmcg_$c.json > gpurun_out/r2u/pmcg_$c.md 2>&1; done
head -30 gpurun_out/r2u/pmcg_FETCH_SIZE.md | cut -c1-220
rm -rf gpurun_out/r2u/pmc_FETCH_SIZE gpurun_out/r2u/pmc_WRITE_SIZE
