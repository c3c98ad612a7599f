mkdir -p gpurun_out/r2i
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $GRAFT_REPO_ROOT/gpurun_out/r2i/pmc -o p -- python $GRAFT_REPO_ROOT/tools/bench_gemm2.py --tiles 0,13 --only "val fc1,val proj,dec fc1" > $GRAFT_REPO_ROOT/gpurun_out/r2i/pmc.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/r2i/pmc
