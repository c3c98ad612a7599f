#!/usr/bin/env python
"""bench.py -- Spann3R per-frame forward hot path on MI355X.

A "step" is one pass of the hot path over one synthetic sequence: Spann3R.forward on a 10-frame 224x224 sequence,
batch 1, bf16 MFMA (BASELINE.json configs[1]).  value = whole-job frames/s with inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; independent sequences are sharded across ranks (no data-path collective); the only collective is
one RCCL all_gather of {frames, seconds} per rank.  Rank 0 prints ONE JSON line with the extra objects
  roofline     : dominant kernel (by GPU time) -- algorithmic FLOPs / HIP-event time vs the dense MFMA peak
  memread      : spatial-memory read (S = q.K^T, softmax/threshold, P.V) algorithmic bytes / HIP-event time vs HBM peak
  cpu_baseline : the CPU oracle (a port of the reference algorithm, oracle/) timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "frames/sec/GPU (224px, 10-frame seq) + mem-bank cross-attn HBM GB/s"
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}       # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def flops_per_sequence(n_frames, size):
    """Algorithmic FLOPs of one sequence (FlopCounterMode on the reference)."""
    if size == 224:
        enc, step, read = 122.46e9, 225.59e9, 0.1573e9
    elif size == 512:
        enc, step, read = 723.2e9, 1324.4e9, 4.295e9
    else:
        s = (size / 224.0) ** 2
        enc, step, read = 122.46e9 * s, 225.59e9 * s, 0.1573e9 * s * s
    return n_frames * enc + (n_frames - 1) * step + read * sum(range(1, n_frames - 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--train-policy", action="store_true", help="growing bank (train-mode memory policy, dropout off)")
    ap.add_argument("--schedule", default="", help="comma list of schedule switches to turn OFF: batch_encode, defer_head2, "
                                                   "grouped_decoder (debugging / A-B runs; default = the shipped schedule)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", init_method="env://")   # 'nccl' IS RCCL on ROCm
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from spann3r_amd import Spann3R, FULL
    from spann3r_amd import ops
    from spann3r_amd.runner import make_sequence, run_sequences, gather_stats, aggregate, shard
    from spann3r_amd.weights import synth_state_dict

    sd = synth_state_dict(0, FULL)                      # seeded synthetic weights of the named architecture
    model = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval().set_precision(args.precision)
    if args.train_policy:
        model.train()
        model.mem_dropout.eval()
    if hasattr(model, "use_graphs"):
        model.use_graphs = not args.no_graphs
    for off in filter(None, args.schedule.split(",")):
        assert hasattr(model, off), off
        setattr(model, off, False)

    # rank r owns sequences {s : s mod world == r}; a handful of distinct sequences, cycled
    n_distinct = 4
    my_ids = shard(n_distinct * world, rank, world)
    seqs = [make_sequence(s, args.frames, args.size, args.size, device=dev) for s in my_ids]

    def fwd(seq):
        return model(seq)

    sync = torch.cuda.synchronize
    for i in range(args.warmup):
        fwd(seqs[i % len(seqs)])
    sync()
    if world > 1:
        dist.barrier()
    sync()
    frames, seconds, _ = run_sequences(fwd, [seqs[i % len(seqs)] for i in range(args.steps)], sync=sync)
    if world > 1:
        dist.barrier()
    stats = gather_stats(frames, seconds, device=dev)
    fps, tot_frames, max_seconds = aggregate(stats)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out = {
        "metric": METRIC, "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * max_seconds / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic (seeded frames, seeded random-init weights of the DUSt3R/Spann3R geometry)",
        "config": {"workload": "Spann3R.forward, %d-frame %dx%d sequence, batch 1, ViT-L enc / ViT-B dec / DPT heads, %s memory policy"
                               % (args.frames, args.size, args.size, "train (growing bank)" if args.train_policy else "eval"),
                   "frames_per_step": args.frames, "parallelism": "sequences sharded 1/GPU x%d" % world,
                   "hip_graphs": bool(getattr(model, "use_graphs", False))},
        "per_gpu_frames_per_s": fps / world,
    }
    fl = flops_per_sequence(args.frames, args.size)
    out["end_to_end"] = {"algorithmic_gflop_per_step": fl / 1e9,
                         "achieved_tflops_per_gpu": fl * args.steps / max_seconds / 1e12,
                         "frac_of_mfma_peak": fl * args.steps / max_seconds / 1e12 / PEAK_TFLOPS[args.precision]}

    # ---- per-kernel HIP-event timing over one more sequence (eager launches, same kernels, same stream)
    if not args.no_profile:
        graphs = getattr(model, "use_graphs", False)
        if graphs:
            model.use_graphs = False
        fwd(seqs[0])
        prof = ops.Profiler()
        out["event_bracket_overhead_us"] = 1e3 * prof.calibrate()
        ops.set_profiler(prof)
        fwd(seqs[0])
        ops.set_profiler(None)
        if graphs:
            model.use_graphs = True
        agg = prof.summary()
        total_ms = sum(a["ms"] for a in agg.values())
        top = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
        key, a = top[0]
        # the roof that binds the dominant kernel: the larger of (algorithmic FLOPs / dense MFMA peak) and
        # (algorithmic bytes / HBM peak); frac = that lower bound on the launch time / the measured launch time
        t_ms = a["ms"] * 1e-3
        ach_fl, ach_by = a["flops"] / t_ms / 1e12, a["bytes"] / t_ms / 1e9
        fr_fl, fr_by = ach_fl / PEAK_TFLOPS[args.precision], ach_by / PEAK_HBM_GBS
        common = {"kernel": key, "traffic": None, "launches": a["launches"], "avg_us": 1e3 * a["ms"] / a["launches"],
                  "avg_gflop_per_launch": a["flops"] / a["launches"] / 1e9, "avg_mbyte_per_launch": a["bytes"] / a["launches"] / 1e6,
                  "share_of_gpu_time": a["ms"] / total_ms, "mfma_frac": fr_fl, "hbm_frac": fr_by}
        # HBM traffic of that kernel from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this
        # process): bytes per launch, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_f_pmc_hbm_traffic.json")))["kernels"].get(key)
            if pm and args.precision == "bf16" and args.size == 224:
                common["traffic"] = pm["traffic_bytes"]
                common["traffic_note"] = ("HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                                          "profiles/r01_f_pmc_hbm_traffic.json; algorithmic bytes per launch = %d" % round(a["bytes"] / a["launches"]))
        except (OSError, ValueError, KeyError):
            pass
        if fr_fl >= fr_by:
            out["roofline"] = dict(bound="mfma", achieved=ach_fl, peak=PEAK_TFLOPS[args.precision], unit="TFLOP/s", frac=fr_fl, **common)
        else:
            out["roofline"] = dict(bound="hbm", achieved=ach_by, peak=PEAK_HBM_GBS, unit="GB/s", frac=fr_by, **common)
        out["kernel_breakdown"] = [{"kernel": k, "launches": v["launches"], "ms": round(v["ms"], 4),
                                    "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None}
                                   for k, v in top[:8]]
        out["profiled_sequence_kernel_ms"] = total_ms
        reads = [r for r in prof.region_summary() if r["key"] == "memread"]
        if reads:
            big = max(reads, key=lambda r: r["info"]["M"])
            gbs = big["bytes"] / (big["ms"] * 1e-3) / 1e9
            out["memread"] = {"what": "spatial-memory read (LN_q, S = q.K_hat^T/32, softmax+threshold, P.V_hat + q, colsum)",
                              "bank_tokens": big["info"]["M"], "algorithmic_bytes": big["bytes"], "us": 1e3 * big["ms"],
                              "launches": big["launches"], "us_uncorrected": 1e3 * big["raw_ms"],
                              "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                              "reads_per_sequence": len(reads), "all_reads_us": [round(1e3 * r["ms"], 2) for r in reads]}

    # ---- CPU baseline: the oracle (port of the reference algorithm) on this box's host cores, bounded sample
    if not args.no_cpu_baseline:
        from oracle import spann3r_oracle as O
        nfr = 3 if args.size > 224 else 5
        cpu_frames = make_sequence(0, nfr, args.size, args.size)
        t0 = time.perf_counter()
        O.forward(cpu_frames, sd, FULL, training_policy=args.train_policy)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nfr / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "one %d-frame %dx%d sequence, fp32, torch-CPU oracle (oracle/spann3r_oracle.py), %.1f s"
                                         % (nfr, args.size, args.size, dt)}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
