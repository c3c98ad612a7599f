#!/bin/bash
# One entry point for the GPU-box sessions of a round (gpurun ships the repository and runs ONE command):
#   gpurun --timeout 900 -- 'bash tools/gpu_run.sh <tag> <stage> [<stage> ...]'
# Every stage writes under gpurun_out/<tag>/ (merged back by gpurun).  Stages:
#   optests   kernel-level parity tests (tests/test_ops_gpu.py)          gemmsweep  tile sweep of the many-row GEMMs
#   modeltests tests/test_model_gpu.py                                    alltests   the whole -m gpu suite
#   bench     python bench.py (default line)                              benchfast  bench.py without extras / CPU baseline
#   cold      tools/cold_start.py (first call / later calls / eager)      prof       rocprofv3 --kernel-trace --stats of benchfast
#   train     tools/train_step_time.py (bf16, fp32; trainbf: bf16 only)   memread    tools/bench_memread.py
#   trainprof rocprofv3 --kernel-trace --stats of the training step       traintests tests/test_train.py + test_loss.py
#   trainsweep tile sweep of the training step's ~800-row GEMM shapes     f16x3      the fp16-split mode: parity tests + bench
#   pmcbench  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the bench (then tools/pmc_traffic_json.py)
#   prof3     rocprofv3 kernel stats of config 3 (512 x 512, 50 frames)  pmcmemlong FETCH / WRITE of the long-bank memory read
#   pmcattn   SQ / LDS / TA counters of tools/ubench/attn_qb.bin (long-sequence attention and its candidate variants)
#   pmcmanyrow SQ / MFMA-busy / FETCH / WRITE counters of the many-row lean GEMMs at 8192 / 16384 rows (tools/bench_manyrow.py)
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for stage in "$@"; do
  echo "=== $stage $(date +%T)" | tee -a "$OUT/log.txt"
  case $stage in
    optests)    timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > "$OUT/optests.txt" 2>&1; tail -5 "$OUT/optests.txt" ;;
    pipetests)  timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "lds_staged or pipelined" > "$OUT/pipetests.txt" 2>&1; tail -15 "$OUT/pipetests.txt" ;;
    modeltests) timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu > "$OUT/modeltests.txt" 2>&1; tail -5 "$OUT/modeltests.txt" ;;
    alltests)   timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/alltests.txt" 2>&1; tail -8 "$OUT/alltests.txt" ;;
    gemmsweep)  timeout 600 python tools/bench_gemm2.py --big > "$OUT/gemmsweep.txt" 2>&1; cat "$OUT/gemmsweep.txt" ;;
    trainattn)  for a in "" "--cross" "--precision fp32"; do timeout 300 python tools/bench_train_attention.py $a >> "$OUT/trainattn.txt" 2>&1; done; grep "us per" "$OUT/trainattn.txt" ;;
    trainsweep) timeout 600 python tools/bench_gemm2.py --train --mb 1 > "$OUT/trainsweep.txt" 2>&1; cat "$OUT/trainsweep.txt" ;;
    ksweep)     timeout 600 python tools/bench_gemm2.py --big --ksweep > "$OUT/ksweep_cold.txt" 2>&1; cat "$OUT/ksweep_cold.txt"; timeout 600 python tools/bench_gemm2.py --big --ksweep --mb 1 > "$OUT/ksweep_warm.txt" 2>&1; cat "$OUT/ksweep_warm.txt" ;;
    pmcgemm)    for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" ; do
                  n=$(echo $C | cut -d" " -f1); (cd /tmp && timeout 300 rocprofv3 --pmc $C -d "$OLDPWD/$OUT/pmc_$n" -o run --output-format csv -- python "$OLDPWD/tools/bench_gemm2.py" --big --only "enc16 fc2,enc fc1" --tiles 20,5 > "$OLDPWD/$OUT/pmc_$n.log" 2>&1); F=$(find "$OUT/pmc_$n" -name "*counter_collection.csv" | head -1); python tools/pmc_csv.py "$F" > "$OUT/pmc_$n.txt" 2>&1; rm -rf "$OUT/pmc_$n"; cat "$OUT/pmc_$n.txt"; done ;;
    pmcattn)    # SQ / LDS / TA counters of the packed attention at long sequences (product kernel, QB variants, v2): what the launch waits for
                for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" ; do
                  n=$(echo $C | cut -d" " -f1); (cd /tmp && timeout 200 rocprofv3 --pmc $C -d "$OLDPWD/$OUT/pmca_$n" -o run --output-format csv -- "$OLDPWD/tools/ubench/attn_qb.bin" > "$OLDPWD/$OUT/pmca_$n.log" 2>&1); F=$(find "$OUT/pmca_$n" -name "*counter_collection.csv" | head -1); python tools/pmc_csv.py "$F" attention > "$OUT/pmca_$n.txt" 2>&1; cp "$F" "$OUT/pmca_$n.csv"; rm -rf "$OUT/pmca_$n"; cat "$OUT/pmca_$n.txt" | cut -c1-220; done ;;
    pmcmanyrow) # SQ counters of the many-row lean GEMMs at 8192 / 16384 rows (tools/bench_manyrow.py): MFMA pipe busy next to wave / issue-stall cycles
                for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
                  n=$(echo $C | cut -d" " -f1); (cd /tmp && timeout 300 rocprofv3 --pmc $C -d "$OLDPWD/$OUT/pmcr_$n" -o run --output-format csv -- python "$OLDPWD/tools/bench_manyrow.py" --rows 16384,8192 > "$OLDPWD/$OUT/pmcr_$n.log" 2>&1); F=$(find "$OUT/pmcr_$n" -name "*counter_collection.csv" | head -1); python tools/pmc_csv.py "$F" bm_kernel > "$OUT/pmcmanyrow_$n.txt" 2>&1; rm -rf "$OUT/pmcr_$n"; cat "$OUT/pmcmanyrow_$n.txt" | cut -c1-160 | head -60; done ;;
    tracegemm)  for t in 20 22; do for a in gelu noact; do timeout 120 python tools/trace_gemm.py $t $a >> "$OUT/tracegemm.txt" 2>&1; done; done; cat "$OUT/tracegemm.txt" ;;
    bench)      timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; head -c 1500 "$OUT/bench.json"; tail -3 "$OUT/bench.err" ;;
    benchfast)  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/benchfast.json" 2> "$OUT/benchfast.err"; head -c 1200 "$OUT/benchfast.json"; tail -3 "$OUT/benchfast.err" ;;
    cold)       timeout 400 python tools/cold_start.py > "$OUT/cold.txt" 2>&1; tail -12 "$OUT/cold.txt"; timeout 400 python tools/cold_start.py --size 512 --frames 50 --train-policy --calls 4 > "$OUT/cold512.txt" 2>&1; tail -9 "$OUT/cold512.txt" ;;
    bench3)     timeout 600 python bench.py --size 512 --frames 50 --train-policy --steps 3 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/bench3.json" 2> "$OUT/bench3.err"; head -c 1500 "$OUT/bench3.json"; tail -3 "$OUT/bench3.err" ;;
    prof)       (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o run -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-extras > "$OLDPWD/$OUT/prof.log" 2>&1); DB=$(find "$OUT/prof" -name "*results.db" | head -1); python tools/rocpd_stats.py "$DB" > "$OUT/prof_stats.md" 2>&1; python tools/rocpd_lastseq.py "$DB" > "$OUT/prof_lastseq.txt" 2>&1; rm -rf "$OUT/prof"; head -40 "$OUT/prof_stats.md" ;;
    pmcbench)   for C in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 600 rocprofv3 --pmc $C -d "$OLDPWD/$OUT/pmcb_$C" -o run -- python "$OLDPWD/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-extras --no-graphs > "$OLDPWD/$OUT/pmcb_$C.log" 2>&1); DB=$(find "$OUT/pmcb_$C" -name "*results.db" | head -1); python tools/rocpd_pmc_grid.py "$DB" --json "$OUT/pmc_$C.json" > "$OUT/pmc_${C}_by_grid.md" 2>&1; rm -rf "$OUT/pmcb_$C"; head -12 "$OUT/pmc_${C}_by_grid.md" | cut -c1-200; done ;;
    proff16)    # rocprofv3 kernel stats of the f16x3 parity mode on the headline workload
                (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/proff16" -o run -- python "$OLDPWD/bench.py" --precision f16x3 --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-extras > "$OLDPWD/$OUT/proff16.log" 2>&1); DB=$(find "$OUT/proff16" -name "*results.db" | head -1); python tools/rocpd_stats.py "$DB" > "$OUT/proff16_stats.md" 2>&1; python tools/rocpd_lastseq.py "$DB" > "$OUT/proff16_lastseq.txt" 2>&1; rm -rf "$OUT/proff16"; head -40 "$OUT/proff16_stats.md" | cut -c1-200 ;;
    prof3)      # rocprofv3 kernel stats of BASELINE config 3 (512 x 512, 50 frames, growing bank)
                (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof3" -o run -- python "$OLDPWD/bench.py" --size 512 --frames 50 --train-policy --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras > "$OLDPWD/$OUT/prof3.log" 2>&1); DB=$(find "$OUT/prof3" -name "*results.db" | head -1); python tools/rocpd_stats.py "$DB" > "$OUT/prof3_stats.md" 2>&1; rm -rf "$OUT/prof3"; head -30 "$OUT/prof3_stats.md" | cut -c1-180 ;;
    pmcmemlong) # HBM traffic of the long-bank memory read (50 176 tokens x 1024 queries), launch by launch: FETCH_SIZE / WRITE_SIZE passes
                for C in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 300 rocprofv3 --pmc $C -d "$OLDPWD/$OUT/pmcm_$C" -o run --output-format csv -- python "$OLDPWD/tools/bench_memread.py" --tokens 50176 --rows 1024 --copies 3 > "$OLDPWD/$OUT/pmcm_$C.log" 2>&1); F=$(find "$OUT/pmcm_$C" -name "*counter_collection.csv" | head -1); python tools/pmc_csv.py "$F" gemm bm_kernel pvs_kernel prob_merge colsum_prob softmax_ reduce_ln colsum_packed > "$OUT/pmcmemlong_$C.txt" 2>&1; rm -rf "$OUT/pmcm_$C"; cat "$OUT/pmcmemlong_$C.txt" | cut -c1-200; done ;;
    trainbf)    timeout 900 python tools/train_step_time.py --precision bf16 >> "$OUT/train.txt" 2>&1; tail -12 "$OUT/train.txt" ;;
    train)      for pr in bf16 fp32; do timeout 900 python tools/train_step_time.py --precision $pr >> "$OUT/train.txt" 2>&1; done; tail -20 "$OUT/train.txt" ;;
    trainprof)  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/tprof" -o run -- python "$OLDPWD/tools/train_step_time.py" --precision bf16 --steps 2 > "$OLDPWD/$OUT/trainprof.log" 2>&1); DB=$(find "$OUT/tprof" -name "*results.db" | head -1); python tools/rocpd_stats.py "$DB" > "$OUT/trainprof_stats.md" 2>&1; rm -rf "$OUT/tprof"; head -45 "$OUT/trainprof_stats.md" | cut -c1-200 ;;
    traintests) timeout 1500 python -m pytest tests/test_train.py tests/test_loss.py -q -m gpu -s > "$OUT/traintests.txt" 2>&1; grep -v "^$" "$OUT/traintests.txt" | tail -25 ;;
    memread)    timeout 600 python tools/bench_memread.py > "$OUT/memread.txt" 2>&1; tail -30 "$OUT/memread.txt" ;;
    memreadlong) timeout 600 python tools/bench_memread.py --tokens 50176 --rows 1024 --copies 3 > "$OUT/memread_long.txt" 2>&1; tail -30 "$OUT/memread_long.txt" ;;
    f16x3)      timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -s -k "f16x3 or stress or precision or f32x3" > "$OUT/f16x3_tests.txt" 2>&1; grep -v "^$" "$OUT/f16x3_tests.txt" | tail -25
                timeout 600 python bench.py --precision f16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_f16x3.json" 2> "$OUT/bench_f16x3.err"; head -c 1200 "$OUT/bench_f16x3.json"; tail -3 "$OUT/bench_f16x3.err" ;;
    *)          echo "unknown stage $stage" ;;
  esac
done
echo "=== done $(date +%T)" | tee -a "$OUT/log.txt"
