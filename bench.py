#!/usr/bin/env python
"""bench.py -- Spann3R per-frame forward hot path on MI355X.

A "step" is one pass of the hot path over one synthetic sequence: Spann3R.forward on a 10-frame 224x224 sequence,
batch 1, bf16 MFMA (BASELINE.json configs[1]).  value = whole-job frames/s with inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a launcher: this process spawns N ranks itself (one per GPU, pinned to a slice of the host cores,
      RCCL rendezvous on 127.0.0.1) and prints rank 0's line; it exits non-zero instead of silently running fewer ranks.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
      the launcher's RANK / LOCAL_RANK / WORLD_SIZE are used as they are.

One process per GPU; independent sequences are sharded across ranks (no data-path collective); the only collective is
one RCCL all_gather of {frames, seconds} per rank, on the device.  Rank 0 prints ONE JSON line with the extra objects
  roofline     : dominant kernel (by GPU time): algorithmic bytes / FLOPs per launch over its average launch duration,
                 measured live with HIP events on the launch stream (event cost calibrated against a spin kernel)
  memread      : the spatial-memory read region (all its launches): algorithmic bytes / HIP-event time vs the HBM peak
  fp32, config3: (N = 1 only) the parity mode on the same workload, and BASELINE config 3 (512x512, 50 frames, growing bank)
  cpu_baseline : (N = 1 only) the CPU oracle (a port of the reference algorithm, oracle/) on this box's host cores
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "frames/sec/GPU (224px, 10-frame seq) + mem-bank cross-attn HBM GB/s"
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "f32x3": 2500.0 / 3, "f32x6": 2500.0 / 6, "f16x3": 2500.0 / 3}       # dense MFMA peaks, MI355X_MICROARCH.md (f32x3 / f32x6: 3 / 6 bf16 MFMAs per product)
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "f32x3", "f32x6", "f16x3"])
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 and config-3 secondary measurements")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--train-policy", action="store_true", help="growing bank (train-mode memory policy, dropout off)")
    ap.add_argument("--train", action="store_true", help="time the training step (BASELINE config 5 per rank: batch 4, 5 frames of 224x224, "
                                                         "ConfLoss backward through the memory fusion, AdamW) instead of the forward")
    ap.add_argument("--train-precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--train-batch", type=int, default=4)
    ap.add_argument("--schedule", default="", help="comma list of schedule switches to turn OFF: batch_encode, defer_head2, "
                                                   "grouped_decoder (debugging / A-B runs; default = the shipped schedule)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- self-spawn (no launcher)
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Start n ranks of this script (one per GPU), CPU cores split evenly, and relay rank 0's JSON line.
    Returns the exit code: non-zero if any rank fails or the box has fewer than n GPUs."""
    import torch
    have = torch.cuda.device_count()
    if have < n:
        print("bench.py: --gpus %d requested but only %d GPU(s) visible; refusing to run a smaller job" % (n, have), file=sys.stderr)
        return 2
    port = _free_port()
    cores = sorted(os.sched_getaffinity(0))
    per = max(1, len(cores) // n)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   SP3_BENCH_CORES=",".join(map(str, cores[r * per:(r + 1) * per])), OMP_NUM_THREADS=str(per))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(rcs):
        print("bench.py: rank exit codes %s" % rcs, file=sys.stderr)
        return 1
    sys.stdout.write(out0)
    return 0


# ----------------------------------------------------------------------------- measurement pieces
def build_model(precision, dev, train_policy=False, schedule_off=(), graphs=True):
    from spann3r_amd import Spann3R, FULL
    from spann3r_amd.weights import synth_state_dict
    sd = synth_state_dict(0, FULL)                      # seeded synthetic weights of the named architecture
    model = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval().set_precision(precision)
    if train_policy:
        model.train()
        model.mem_dropout.eval()
    model.use_graphs = graphs
    for off in schedule_off:
        assert hasattr(model, off), off
        setattr(model, off, False)
    return model, sd


def time_sequences(model, seqs, steps, warmup, world=1):
    import torch
    import torch.distributed as dist
    from spann3r_amd.runner import run_sequences
    sync = torch.cuda.synchronize
    for i in range(warmup):
        model(seqs[i % len(seqs)])
    sync()
    if world > 1:
        dist.barrier()
    sync()
    frames, seconds, _ = run_sequences(model, [seqs[i % len(seqs)] for i in range(steps)], sync=sync)
    if world > 1:
        dist.barrier()
    return frames, seconds


def instance_key(key):
    """profiler key -> kernel instance: the paired form of a GEMM launch (gemm2<..>) is the same instantiation as the single one"""
    return "gemm<" + key[len("gemm2<"):] if key.startswith("gemm2<") else key


def kernel_symbol(key):
    """profiler key -> the __global__ template the launch instantiates (csrc/): what rocprofv3 --kernel-trace groups by"""
    if key.startswith("gemm<") or key.startswith("gemm2<"):
        tile = key.rstrip(">").rsplit(",", 1)[-1]
        if tile.startswith("lean-conv"):
            return "conv_sm_kernel"
        if tile.startswith("lean-softmax-pv"):
            return "pv_kernel"
        if tile.startswith("lean-prob-pv"):
            return "pvs_kernel"
        if tile.startswith("lean-"):
            return "bm_kernel" if tile.rsplit("-", 1)[-1] in ("256x128", "128x128", "128x64") else "sm_kernel"
        return "gemm_kernel"
    if key.startswith("attention_packed_qproj"):
        return "attention_packed_kernel"                    # (the same template with the q projection switched on)
    return key.split("<", 1)[0] + "_kernel"


def kernel_profile(model, seq, precision):
    """One more sequence with eager launches (same kernels, same stream, same order) and every launch timed with HIP
    events on the launch stream.  Returns (roofline dict, breakdown list, total kernel ms, memread dict or None)."""
    from spann3r_amd import ops
    graphs = model.use_graphs
    model.use_graphs = False
    try:
        model(seq)
        prof = ops.Profiler()
        cal = prof.calibrate()
        ops.set_profiler(prof)
        model(seq)
    finally:
        ops.set_profiler(None)
        model.use_graphs = graphs
    agg = prof.summary()
    total_ms = sum(a["ms"] for a in agg.values())
    # The dominant kernel is chosen by kernel SYMBOL (the template a launch instantiates), then by instance inside that symbol:
    # a paired launch (gemm2<..>) and a single launch (gemm<..>) of ONE lean instance are the same code and are booked together,
    # and two different symbols that used to share a key (conv3x3_wide_kernel / conv3x3_tile_kernel) are not.
    inst = {}
    for k, v in agg.items():
        a = inst.setdefault(instance_key(k), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, keys=[]))
        for f in ("launches", "ms", "flops", "bytes"):
            a[f] += v[f]
        a["keys"].append(k)
    fam = {}
    for k, v in inst.items():
        a = fam.setdefault(kernel_symbol(k), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, instances=[]))
        for f in ("launches", "ms", "flops", "bytes"):
            a[f] += v[f]
        a["instances"].append(k)
    sym = max(fam, key=lambda k: fam[k]["ms"])
    top = sorted(inst.items(), key=lambda kv: -kv[1]["ms"])
    key, a = max(((k, inst[k]) for k in fam[sym]["instances"]), key=lambda kv: kv[1]["ms"])
    fa = fam[sym]
    family = {"symbol": sym, "instances": len(fa["instances"]), "launches": fa["launches"], "share_of_gpu_time": fa["ms"] / total_ms,
              # time-weighted over every launch of the symbol: total algorithmic work / total time
              "mfma_frac": fa["flops"] / (fa["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS[precision],
              "hbm_frac": fa["bytes"] / (fa["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
              "by_symbol": {k: round(v["ms"] / total_ms, 4) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:8]}}
    # the roof that binds the dominant kernel: the larger of (algorithmic FLOPs / dense MFMA peak) and
    # (algorithmic bytes / HBM peak); frac = that lower bound on the launch time / the measured launch time
    t_s = a["ms"] * 1e-3
    ach_fl, ach_by = a["flops"] / t_s / 1e12, a["bytes"] / t_s / 1e9
    fr_fl, fr_by = ach_fl / PEAK_TFLOPS[precision], ach_by / PEAK_HBM_GBS
    common = {"kernel": key, "family": family, "traffic": None, "launches": a["launches"], "avg_us": 1e3 * a["ms"] / a["launches"],
              "avg_gflop_per_launch": a["flops"] / a["launches"] / 1e9, "avg_mbyte_per_launch": a["bytes"] / a["launches"] / 1e6,
              "share_of_gpu_time": a["ms"] / total_ms, "mfma_frac": fr_fl, "hbm_frac": fr_by,
              "timing": "HIP events on the launch stream, one bracket per launch, minus the event cost measured around a spin "
                        "kernel of known length (%.2f us); compare profiles/ rocprofv3 --kernel-trace of the same build" % (1e3 * cal)}
    # HBM traffic of that kernel from PMC passes (rocprofv3 cannot run inside this process): bytes per launch, FETCH_SIZE
    # doubled as MI355X_MICROARCH.md prescribes for gfx950.  Only reported when the committed file names THIS kernel.
    try:
        pm_all = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")))
        # (an instance booked from several profiler keys -- paired and single launches -- gets the launch-weighted mean of their entries)
        pms = [pm_all["kernels"][k] for k in a["keys"] if k in pm_all["kernels"] and pm_all["kernels"][k].get("traffic_bytes") is not None]
        pm = None
        if pms:
            n = sum(max(q.get("launches", 1), 1) for q in pms)
            pm = {"traffic_bytes": int(round(sum(q["traffic_bytes"] * max(q.get("launches", 1), 1) for q in pms) / n))}
        if pm and precision == "bf16":
            common["traffic"] = pm["traffic_bytes"]
            common["traffic_build"] = pm_all.get("build", "unknown")
            common["traffic_note"] = ("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of build '%s' "
                                      "(profiles/pmc_hbm_traffic.json), not of this process" % pm_all.get("build", "unknown"))
    except (OSError, ValueError, KeyError):
        pass
    # the launches closest to the MFMA roof (instances with >= 1 % of the GPU time): where the many-row kernels stand when a launch holds
    # several tiles per CU (the 16-image encoder chunks of a 512 x 512 sequence), next to the dominant kernel above
    best = sorted(((k, v) for k, v in inst.items() if v["ms"] >= 0.01 * total_ms and v["flops"] > 0), key=lambda kv: -kv[1]["flops"] / kv[1]["ms"])[:3]
    common["highest_mfma"] = [{"kernel": k, "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                               "mfma_frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS[precision], 4),
                               "avg_us": round(1e3 * v["ms"] / v["launches"], 2), "launches": v["launches"],
                               "share_of_gpu_time": round(v["ms"] / total_ms, 4)} for k, v in best]
    if fr_fl >= fr_by:
        roof = dict(bound="mfma", achieved=ach_fl, peak=PEAK_TFLOPS[precision], unit="TFLOP/s", frac=fr_fl, **common)
    else:
        roof = dict(bound="hbm", achieved=ach_by, peak=PEAK_HBM_GBS, unit="GB/s", frac=fr_by, **common)
    breakdown = [{"kernel": k, "launches": v["launches"], "ms": round(v["ms"], 4), "avg_us": round(1e3 * v["ms"] / v["launches"], 2),
                  "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None,
                  "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None}
                 for k, v in top[:16]]
    memread = None
    reads = [r for r in prof.region_summary() if r["key"] == "memread"]
    if reads:
        big = max(reads, key=lambda r: r["info"]["M"])
        gbs = big["bytes"] / (big["ms"] * 1e-3) / 1e9
        memread = {"what": "spatial-memory read: score GEMM (LN_q folded, softmax statistics in its epilogue) -> P.V_hat + q with the "
                           "thresholded probabilities built on load (short banks); long banks of a 512x512 frame: score stage that writes bf16 "
                           "exp(s - group max) + group statistics, merge, P.V stage with the groups' rescale in its loop, split-K reduce + q "
                           "(no score matrix; round 5: fp32 scores + softmax launch + split-K P.V); column sums",
                   "bank_tokens": big["info"]["M"], "queries": big["info"]["tokens_per_frame"], "algorithmic_bytes": big["bytes"],
                   "us": 1e3 * big["ms"], "launches": big["launches"], "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                   "frac": gbs / PEAK_HBM_GBS, "gflop": big["info"].get("flops", 0.0) / 1e9,
                   "tflops": big["info"].get("flops", 0.0) / (big["ms"] * 1e-3) / 1e12,
                   "reads_per_sequence": len(reads), "all_reads_us": [round(1e3 * r["ms"], 2) for r in reads]}
    return roof, breakdown, total_ms, memread


def _promote_replay(memread, rep):
    """The model runs under hipGraph replay, so the replayed read is the headline number of the object; the per-launch
    event brackets of the eager profiling pass (event cost inside) stay as `eager_event_brackets`."""
    if not rep:
        return
    memread["eager_event_brackets"] = {k: memread[k] for k in ("bank_tokens", "us", "achieved", "frac", "tflops", "all_reads_us") if k in memread}
    for k in ("bank_tokens", "queries", "us", "achieved", "frac", "tflops", "algorithmic_bytes"):
        memread[k] = rep[k]
    if "critical_path" in rep:
        memread["critical_path"] = rep["critical_path"]
        memread["launches"] = 3
    memread["timing"] = rep["timing"]
    memread.pop("all_reads_us", None)
    _memread_roofline(memread)


def _memread_roofline(memread, precision="bf16"):
    """which roof binds THIS read (few queries, short bank: HBM bytes of K_hat + V_hat; 1024 queries over a long bank: 963 FLOP/B,
    the matrix pipe) and how far the replayed read is from it"""
    us = memread.get("us")
    if not us:
        return
    flops = 4.0 * memread["queries"] * memread["bank_tokens"] * 1024
    t_mfma, t_hbm = flops / (PEAK_TFLOPS[precision] * 1e12), memread["algorithmic_bytes"] / (PEAK_HBM_GBS * 1e9)
    bound = "mfma" if t_mfma >= t_hbm else "hbm"
    if bound == "mfma":
        memread["roofline"] = {"bound": "mfma", "achieved": flops / (us * 1e-6) / 1e12, "peak": PEAK_TFLOPS[precision], "unit": "TFLOP/s",
                               "frac": t_mfma / (us * 1e-6)}
    else:
        memread["roofline"] = {"bound": "hbm", "achieved": memread["algorithmic_bytes"] / (us * 1e-6) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": t_hbm / (us * 1e-6)}


def memread_replay(model, reps=20):
    """The spatial-memory read at the bank size the last sequence ended with, as the model issues it (S GEMM with LN_q
    folded, softmax / threshold, P.V GEMM + q, column sums), captured in one hipGraph of `reps` back-to-back reads and
    replayed: microseconds per read including the real launch boundaries, without event brackets between the kernels."""
    import torch
    run = next(reversed(model._runners.values()), None) if model._runners else None
    mem = None if run is None else run.mem
    if mem is None or mem.M == 0:
        return None
    aux = run.k2_aux if run.B == 1 else (None, None)
    keep = mem.bank["attn"].clone()

    def timed(defer):
        def read():
            # defer=False: incl. the column-sum launch (the per-frame step folds it into the launch that commits the frame)
            mem.memory_read(run.k2, run.fuse, *aux, defer_attn=defer)
            mem._pending_attn = None
        read()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                read()
        g.replay()
        torch.cuda.synchronize()
        t = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            t = min(t, e0.elapsed_time(e1) * 1e3 / reps)
        return t
    best = timed(False)
    crit = timed(True) if mem._read_plan()[1] else None
    mem.bank["attn"].copy_(keep)
    es = mem.bank["k_hat"].element_size()
    nbytes = mem.B * (2.0 * mem.M * mem.C * es + 2.0 * mem.P * mem.C * 4)
    flops = 4.0 * mem.B * mem.P * mem.M * mem.C
    extra = {} if crit is None else {"critical_path": {
        "us": crit, "achieved": nbytes / crit / 1e3, "launches": 2,
        "what": "score GEMM + P.V GEMM only: what the next frame's decoder waits for; the column sums (mem_attn) ride in the launch "
                "that commits the frame (the former mem_append launch)"}}
    return {**extra, "bank_tokens": mem.M, "queries": mem.P, "us": best, "achieved": nbytes / best / 1e3, "unit": "GB/s", "peak": PEAK_HBM_GBS,
            "frac": nbytes / best / 1e3 / PEAK_HBM_GBS, "tflops": flops / best / 1e6, "algorithmic_bytes": nbytes,
            "timing": "hipGraph of %d back-to-back reads (every launch of the read, column sums included), HIP events around the "
                      "replay" % reps}


def synth_training_batch(seq_id, n_frames, size, batch, dev):
    """frames + synthetic ground truth (pts3d ~ N((0,0,3), 1), valid_mask ~ Bernoulli(0.9), camera_pose = I)"""
    import torch
    from spann3r_amd.runner import make_sequence
    frames = make_sequence(seq_id, n_frames, size, size, batch=batch, device=dev)
    g = torch.Generator().manual_seed(7000 + seq_id)
    gts = []
    for _ in range(n_frames):
        gts.append(dict(pts3d=(torch.randn(batch, size, size, 3, generator=g) + torch.tensor([0., 0., 3.])).to(dev),
                        valid_mask=(torch.rand(batch, size, size, generator=g) < 0.9).to(dev),
                        camera_pose=torch.eye(4).repeat(batch, 1, 1).to(dev)))
    return frames, gts


def train_measure(dev, steps, warmup, precision, batch, n_frames=5, size=224, world=1, force_collectives=False):
    """seconds per training step of the full model on this rank (forward + ConfLoss + backward incl. bucket all-reduces + clip + AdamW)"""
    import torch
    import torch.distributed as dist
    from spann3r_amd import Spann3R, FULL
    from spann3r_amd.weights import synth_state_dict
    from spann3r_amd import train as T
    model = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False)
    model.load_state_dict(synth_state_dict(0, FULL), strict=True)
    model = model.to(dev)
    # one execution mode at every world size: the step (bucket all-reduces included) is captured into a hipGraph and replayed
    # force_collectives (world 1 only): the bucket all-reduces are issued on a one-rank RCCL group, i.e. the launches, their order and
    # their place inside backward are those of an N-rank job; the data each moves is its own bucket
    ts = T.TrainStep(model, precision=precision, graph=True, force_collectives=force_collectives)
    rank = int(os.environ.get("RANK", "0"))
    frames, gts = synth_training_batch(rank, n_frames, size, batch, dev)
    torch.cuda.reset_peak_memory_stats()
    for _ in range(warmup):
        loss, norm = ts.run(frames, gts)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, norm = ts.run(frames, gts)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    sec = (time.perf_counter() - t0) / steps
    # algorithmic FLOPs: forward of (n frames encoded, n-1 steps) per sample, backward = 2x forward
    fl = 3.0 * batch * flops_per_sequence(n_frames, size)
    out = {"seconds_per_step": sec, "frames_per_s": batch * n_frames / sec, "loss": float(loss), "grad_norm": float(norm) if norm is not None else None,
           "achieved_tflops": fl / sec / 1e12, "frac_of_mfma_peak": fl / sec / 1e12 / PEAK_TFLOPS["bf16" if precision == "bf16" else "fp32"],
           "peak_memory_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "precision": precision, "batch": batch, "frames": n_frames, "size": size,
           "rccl_ranks": world, "buckets": len(ts.reducer.buckets), "buckets_reduced_inside_backward": ts.reducer.launched_in_backward, "hip_graph": bool(ts.graph),
           "collectives_in_graph": bool(ts.graph and ts.reducer.active()),
           "what": "one optimisation step: train-mode Spann3R.forward (HIP autograd ops) + ConfLoss_t(Regr3D_t(L21, avg_dis), 0.4) + backward through the "
                   "memory fusion + bucketed gradient all-reduce (RCCL, world %d) + global-norm clip 1.0 + AdamW on flat buckets" % world}
    T.set_precision("fp32")
    del ts, model
    torch.cuda.empty_cache()
    return out


def parity_errors(model, dev, precision="f16x3"):
    """Errors of `precision` against the committed dumps of the unmodified reference (tests/golden/*.npz: data, not code): the config-2
    fixture on `model`'s weights and the stress fixture on trained-like weights.  Returns {} when the fixtures are absent."""
    import numpy as np
    import torch
    from spann3r_amd import Spann3R, FULL
    from spann3r_amd.weights import synth_frames, stress_state_dict
    out = {}
    keep = model.precision
    for tag, name, stress in (("config2", "spann3r_cfg2_224x10.npz", False), ("stress", "spann3r_stress_224x6.npz", True)):
        path = os.path.join(ROOT, "tests", "golden", name)
        if not os.path.exists(path):
            continue
        g = np.load(path)
        H, W = map(int, g["meta_hw"])
        S, n = int(g["meta_sub"]), int(g["meta_frames"])
        m = model
        if stress:
            m = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False)
            m.load_state_dict(stress_state_dict(7, FULL), strict=True)
            m = m.to(dev).eval()
        m.set_precision(precision)
        frames = [{"img": f["img"].to(dev)} for f in synth_frames(n, H, W)]
        with torch.no_grad():
            preds, _ = m(frames)
        e_max = e_conf = 0.0
        pp = []
        for j, p in enumerate(preds):
            pts = p["pts3d" if j == 0 else "pts3d_in_other_view"][:, ::S, ::S].double().cpu()
            ref = torch.as_tensor(g["pred%d_pts_sub" % j]).double()
            e_max = max(e_max, float((pts - ref).abs().max() / ref.abs().max()))
            pp.append(((pts - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-30)).reshape(-1))
            cr = torch.as_tensor(g["pred%d_conf_sub" % j]).double()
            e_conf = max(e_conf, float((p["conf"][:, ::S, ::S].double().cpu() - cr).abs().max() / cr.abs().max()))
        pp = torch.cat(pp)
        out[tag] = {"pts3d_max_norm": e_max, "conf_max_norm": e_conf, "pts3d_per_point_p999": float(torch.quantile(pp, 0.999)),
                    "pts3d_per_point_max": float(pp.max()), "frames": n}
        if stress:
            del m
            torch.cuda.empty_cache()
    model.set_precision(keep)
    return out


def cold_path(dev, precision, frames, size, steady_fps, sd):
    """demo.py:123-127 calls forward ONCE per scene: the first call of a fresh model (weights packed -- a per-checkpoint cost -- but no
    workspace, no memory arena, no hipGraph yet), the second call, and the eager steady state (use_graphs = False: the same kernels
    launched one by one).  Frames resident in HBM, wall clock incl. the final synchronise.  A first call happens once, so it is
    measured on TWO fresh models (the cyclic collector off while the clock runs: a generation-2 pass over the garbage of a model
    build is 40-70 ms) and both values are reported; `value` is the better one."""
    import gc
    import torch
    from spann3r_amd import Spann3R, FULL
    from spann3r_amd.runner import make_sequence
    seqs = [make_sequence(300 + i, frames, size, size, device=dev) for i in range(4)]

    def timed(m, seq):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m(seq)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    firsts, seconds, thirds, n_graphs, te = [], [], [], 0, None
    for trial in range(2):
        m = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).eval().set_precision(precision)
        m.engine                                              # weight packing
        gc.collect()
        torch.cuda.synchronize()
        gc.disable()
        try:
            firsts.append(timed(m, seqs[0]))
            seconds.append(timed(m, seqs[1]))
            thirds.append(timed(m, seqs[2]))
            n_graphs = len(next(iter(m._runners.values())).graphs)
            if trial == 1:
                m.use_graphs = False
                timed(m, seqs[3])
                te = min(timed(m, seqs[0]), timed(m, seqs[1]))
        finally:
            gc.enable()
        del m
        gc.collect()
        torch.cuda.empty_cache()
    t1, t2, t3 = min(firsts), min(seconds), min(thirds)
    return ({"value": frames / t1, "unit": "frames/s", "ms": 1e3 * t1, "frac_of_steady": frames / t1 / steady_fps,
             "first_call_ms_of_each_fresh_model": [round(1e3 * t, 2) for t in firsts],
             "second_call_frames_per_s": frames / t2, "third_call_frames_per_s": frames / t3, "hip_graphs_after_three_calls": n_graphs,
             "what": "first forward() of a fresh model on one %d-frame %dx%d sequence (engine built, everything else cold: workspaces, memory "
                     "arena, hipGraph capture inside the call), then the second and third call; tools/cold_start.py is the stand-alone form" % (frames, size, size)},
            {"value": frames / te, "unit": "frames/s", "frac_of_steady": frames / te / steady_fps,
             "what": "steady state with use_graphs = False: every kernel of the sequence launched eagerly through the C-ABI"})


def cpu_baseline(sd, size, train_policy):
    """The CPU oracle on this box's host cores: warm-up, a thread-count sweep on a short sequence, then the median of 3
    runs of the bounded sample at the best count."""
    import torch
    from oracle import spann3r_oracle as O
    from spann3r_amd import FULL
    from spann3r_amd.runner import make_sequence
    cores = len(os.sched_getaffinity(0))
    nfr = 3 if size > 224 else 10            # the benched 10-frame sequence itself at 224 x 224
    short = make_sequence(0, 3, size, size)
    sample = make_sequence(0, nfr, size, size)

    def run(frames):
        t0 = time.perf_counter()
        O.forward(frames, sd, FULL, training_policy=train_policy)
        return time.perf_counter() - t0
    run(short)                                                   # warm-up (allocator, oneDNN primitive caches)
    sweep = {}
    for nt in sorted({min(cores, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nt)
        sweep[nt] = len(short) / run(short)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    ts = sorted(run(sample) for _ in range(3))
    ref_note = ("the unmodified reference on the survey host (8 vCPU Xeon @ 2.1 GHz, 8 threads, 10 frames 224x224, warm): 1.53 frames/s "
                "(BASELINE.md section 2); /root/reference does not exist on the GPU box, so the number timed here is the oracle port")
    model_name = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": nfr / ts[1], "unit": "frames/s", "cores": best, "kind": "port", "frames": nfr, "cpu": model_name, "host_cores": cores,
            "reference_measured": ref_note,
            "thread_sweep_frames_per_s": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": "median of 3 runs of one %d-frame %dx%d sequence, fp32, torch-CPU oracle (oracle/spann3r_oracle.py) after a "
                      "warm-up, at the best thread count of the sweep; %.1f s per run" % (nfr, size, size, ts[1])}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    # stdout carries exactly ONE line, the JSON record: anything the model prints on the way (the reference-style
    # "Similarity detected" / "Memory pruned" notes of the memory policy) goes to stderr -- and so does everything native code writes to
    # file descriptor 1 (RCCL prints its version banner through C stdio when its first communicator comes up; buffered, it would land
    # BEHIND the JSON line at exit): descriptor 1 points at stderr for the rest of the process, the record goes to a duplicate of the
    # original descriptor
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    py_stdout, sys.stdout = sys.stdout, sys.stderr
    try:
        _main(args, real_stdout)
    finally:
        real_stdout.flush()
        sys.stdout = py_stdout


def _main(args, real_stdout):

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d; refusing to report a job of a different size" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if os.environ.get("SP3_BENCH_CORES"):
        os.sched_setaffinity(0, {int(c) for c in os.environ["SP3_BENCH_CORES"].split(",")})
    # SP3_FORCE_DIST=1: bring RCCL up also for a one-rank job (tests: the stats all_gather and the rendezvous of the self-spawned
    # ranks run through the same code as an 8-GPU job)
    use_dist = world > 1 or os.environ.get("SP3_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", init_method="env://")   # 'nccl' IS RCCL on ROCm
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from spann3r_amd.runner import make_sequence, gather_stats, aggregate, shard

    if args.train:
        # BASELINE config 5: data-parallel training, batch 4 per rank ("scaling": weak); value = frames/s over all ranks
        tr = train_measure(dev, args.steps, args.warmup, args.train_precision, args.train_batch, world=world)
        stats = gather_stats(args.train_batch * 5 * args.steps, tr["seconds_per_step"] * args.steps, extra=[float(tr["hip_graph"])], device=dev)
        fps, _, max_seconds = aggregate(stats)
        tr["hip_graph_per_rank"] = [bool(x) for x in stats[:, 2].tolist()]       # every rank reports its own execution mode
        if rank == 0:
            print(file=real_stdout, flush=True, *[json.dumps({"metric": "training frames/sec (batch %d x %d ranks, 5-frame 224px sequences, ConfLoss backward, AdamW)" % (args.train_batch, world),
                              "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": 1e3 * max_seconds / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": args.train_precision, "data": "synthetic (seeded frames / ground truth, seeded random-init weights)",
                              "config": {"workload": "train step, BASELINE config 5 per rank: batch %d, 5 frames of 224x224" % args.train_batch,
                                         "parallelism": "dp%d (RCCL bucket all-reduce inside backward)" % world}, "train": tr})])
        if use_dist:
            dist.destroy_process_group()
        return

    model, sd = build_model(args.precision, dev, args.train_policy, tuple(filter(None, args.schedule.split(","))), not args.no_graphs)
    # rank r owns sequences {s : s mod world == r}; a handful of distinct sequences, cycled
    n_distinct = 4
    my_ids = shard(n_distinct * world, rank, world)
    seqs = [make_sequence(s, args.frames, args.size, args.size, device=dev) for s in my_ids]
    frames, seconds = time_sequences(model, seqs, args.steps, args.warmup, world)
    stats = gather_stats(frames, seconds, device=dev)           # RCCL all_gather on the device when world > 1
    fps, tot_frames, max_seconds = aggregate(stats)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    out = {
        "metric": METRIC, "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * max_seconds / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic (seeded frames, seeded random-init weights of the DUSt3R/Spann3R geometry)",
        "config": {"workload": "Spann3R.forward, %d-frame %dx%d sequence, batch 1, ViT-L enc / ViT-B dec / DPT heads, %s memory policy"
                               % (args.frames, args.size, args.size, "train (growing bank)" if args.train_policy else "eval"),
                   "frames_per_step": args.frames, "parallelism": "sequences sharded 1/GPU x%d" % world,
                   "hip_graphs": bool(model.use_graphs)},
        "per_gpu_frames_per_s": fps / world,
        "per_rank_seconds": [round(float(s), 6) for s in stats[:, 1]],
        "rccl_ranks": dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 0,
    }
    fl = flops_per_sequence(args.frames, args.size)
    out["end_to_end"] = {"algorithmic_gflop_per_step": fl / 1e9,
                         "achieved_tflops_per_gpu": fl * args.steps / max_seconds / 1e12,
                         "frac_of_mfma_peak": fl * args.steps / max_seconds / 1e12 / PEAK_TFLOPS[args.precision]}

    # ---- per-kernel HIP-event timing over one more sequence (eager launches, same kernels, same stream)
    if not args.no_profile:
        roof, breakdown, total_ms, memread = kernel_profile(model, seqs[0], args.precision)
        out["roofline"], out["kernel_breakdown"], out["profiled_sequence_kernel_ms"] = roof, breakdown, total_ms
        if memread:
            out["memread"] = memread
            rep = memread_replay(model)
            _promote_replay(out["memread"], rep)
            if rep:
                out["memread"]["graph_replay"] = rep

    # ---- secondary measurements on this GPU (N = 1 only): the fp32 parity mode, and BASELINE config 3
    if world == 1 and not args.no_extras and (args.size, args.frames, args.precision, args.train_policy) == (224, 10, "bf16", False):
        print("bench.py: extras: fp32 mode", file=sys.stderr, flush=True)
        model.set_precision("fp32")
        f32_frames, f32_s = time_sequences(model, seqs, 6, 3)
        out["fp32"] = {"value": f32_frames / f32_s, "unit": "frames/s", "what": "same workload in the fp32 MFMA parity mode "
                       "(<=1e-3 vs the reference, tests/test_model_gpu.py)", "steps": 6,
                       "frac_of_mfma_peak": fl * 6 / f32_s / 1e12 / PEAK_TFLOPS["fp32"]}
        model.set_precision("f32x3")
        x3_frames, x3_s = time_sequences(model, seqs, 6, 3)
        out["f32x3"] = {"value": x3_frames / x3_s, "unit": "frames/s", "steps": 6, "parity_mode": False,
                        "what": "same workload, fp32 operands with every GEMM product through three bf16 MFMAs of a (hi, lo) split (16 "
                                "mantissa bits per product, fp32 accumulate).  NOT a parity mode: it FAILS the 1e-3 bar on the trained-like "
                                "stress fixture (1.4e-3 pointmaps / 2.0e-3 mem_attn; <= 4e-4 on the reference-initialised fixtures); "
                                "kept as a throughput data point only -- the parity-carrying modes are fp32, f32x6 and f16x3"}
        model.set_precision("f32x6")
        x6_frames, x6_s = time_sequences(model, seqs, 6, 3)
        out["f32x6"] = {"value": x6_frames / x6_s, "unit": "frames/s", "steps": 6,
                        "what": "same workload, fp32 operands, every GEMM product through SIX bf16 MFMAs of a three-way (h, m, l) split: 24 operand "
                                "bits, fp32-grade products, fp32 accumulate; exact-fp32 attention.  The fast parity mode that also holds 1e-3 on "
                                "trained-like weight statistics (stress fixture), where f32x3 measures 1.4e-3"}
        print("bench.py: extras: f16x3 mode", file=sys.stderr, flush=True)
        model.set_precision("f16x3")
        h3_frames, h3_s = time_sequences(model, seqs, 6, 3)
        out["f16x3"] = {"value": h3_frames / h3_s, "unit": "frames/s", "steps": 6,
                        "what": "same workload, fp32 operands, every GEMM and attention product through THREE fp16 MFMAs of a two-way split "
                                "x = h + l * 2^-11 (22 operand bits, fp32 accumulate): the fast fp32-grade mode -- pointmaps within 5e-6 of the reference on "
                                "the config-2 fixture and 1.6e-4 on the trained-like stress fixture, where f32x3 measures 1.4e-3 and exact fp32 1.4e-4 "
                                "(tests/test_model_gpu.py)"}
        print("bench.py: extras: parity errors of f16x3 against the reference dumps", file=sys.stderr, flush=True)
        # the >= 200 frames/s-at-1e-3 claim as ONE record: the fp32-grade mode's rate next to its errors against the reference
        # dumps (tests/golden: outputs of the unmodified reference), measured here, now, on this build
        out["parity_mode"] = {"mode": "f16x3", "value": h3_frames / h3_s, "unit": "frames/s", "tolerance": 1e-3,
                              "errors_vs_reference": parity_errors(model, dev),
                              "what": "pointmaps / confidences of the f16x3 mode against dumps of the unmodified reference (torch CPU fp32) on the "
                                      "config-2 fixture (10 x 224x224, synthetic reference-initialised weights) and the stress fixture (6 frames, "
                                      "trained-like weight statistics): max-norm relative error max|d| / max|ref| and the 99.9th percentile of the "
                                      "per-point error |d| / |p|; asserted in tests/test_model_gpu.py"}
        model.set_precision("bf16")
        print("bench.py: extras: cold path (first call of a fresh model, eager steady state)", file=sys.stderr, flush=True)
        out["cold_first_call"], out["eager"] = cold_path(dev, "bf16", args.frames, args.size, fps, sd)
        # how the same GPU fills with more independent work per launch: 4 sequences batched into one forward (B = 4).
        # NOT the headline configuration (BASELINE config 2 is batch 1): reported next to it.
        print("bench.py: extras: batch 4", file=sys.stderr, flush=True)
        seqs_b4 = [make_sequence(200, args.frames, args.size, args.size, batch=4, device=dev)]
        b4_frames, b4_s = time_sequences(model, seqs_b4, 6, 3)
        out["batch4"] = {"value": 4 * b4_frames / b4_s, "unit": "frames/s", "steps": 6,
                         "what": "same model and sequence length, 4 independent sequences per forward call (batch 4): "
                                 "every launch carries 4x the rows; not the BASELINE configuration"}
        del model
        torch.cuda.empty_cache()
        print("bench.py: extras: config 3 (512x512, 50 frames)", file=sys.stderr, flush=True)
        m3, _ = build_model("bf16", dev, train_policy=True)
        seq3 = [make_sequence(100, 50, 512, 512, device=dev)]
        c3_frames, c3_s = time_sequences(m3, seq3, 3, 3)
        fl3 = flops_per_sequence(50, 512)
        c3 = {"workload": "Spann3R.forward, 50-frame 512x512 sequence, batch 1, growing bank (train memory policy, dropout off), bf16",
              "value": c3_frames / c3_s, "unit": "frames/s", "steps": 3, "ms_per_sequence": 1e3 * c3_s / 3,
              "frac_of_mfma_peak": fl3 * 3 / c3_s / 1e12 / PEAK_TFLOPS["bf16"]}
        if not args.no_profile:
            roof3, breakdown3, _, memread3 = kernel_profile(m3, seq3[0], "bf16")
            c3["dominant_kernel"] = {k: roof3[k] for k in ("kernel", "avg_us", "launches", "share_of_gpu_time", "mfma_frac", "hbm_frac")}
            c3["kernel_breakdown"] = breakdown3[:5]
            c3["highest_mfma_kernels"] = roof3["highest_mfma"]
            if memread3:
                c3["memread"] = memread3
                rep3 = memread_replay(m3, reps=5)
                _promote_replay(c3["memread"], rep3)
                if rep3:
                    c3["memread"]["graph_replay"] = rep3
        out["config3"] = c3
        del m3
        import gc
        gc.collect()                     # runner <-> model cycles hold hipGraphs: finalise them now, not inside a later capture
        torch.cuda.empty_cache()

    # ---- the training step (BASELINE config 5, one rank's share): bf16 products, flat buckets, device-side clip
    if world == 1 and not args.no_extras and args.precision == "bf16":
        try:
            del model
        except NameError:
            pass
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        print("bench.py: extras: training step", file=sys.stderr, flush=True)
        out["train"] = train_measure(dev, 3, 2, "bf16", 4)
        # the same step with the data-parallel machinery live on ONE rank: RCCL group of size 1, every gradient bucket all-reduced from
        # inside backward and captured into the step's hipGraph (what each of the 8 ranks of BASELINE config 5 executes)
        try:
            own_group = not dist.is_initialized()
            if own_group:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29543")
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                dist.init_process_group(backend="nccl", rank=0, world_size=1)
            trf = train_measure(dev, 3, 2, "bf16", 4, force_collectives=True)
            out["train"]["rccl_world1"] = {k: trf[k] for k in ("seconds_per_step", "frames_per_s", "buckets", "buckets_reduced_inside_backward",
                                                                "hip_graph", "collectives_in_graph", "loss", "grad_norm")}
            out["train"]["rccl_world1"]["what"] = ("the same step with its gradient buckets all-reduced over a ONE-rank RCCL group from inside backward "
                                                   "(TrainStep(force_collectives=True)): the launch sequence of a rank of the 8-GPU job")
            if own_group:
                dist.destroy_process_group()
        except Exception as e:       # the extra must never cost the line
            out["train"]["rccl_world1"] = {"error": repr(e)[:300]}

    # ---- CPU baseline: the oracle (port of the reference algorithm) on this box's host cores, bounded sample
    if world == 1 and not args.no_cpu_baseline:
        print("bench.py: cpu baseline", file=sys.stderr, flush=True)
        out["cpu_baseline"] = cpu_baseline(sd, args.size, args.train_policy)
    print(json.dumps(out), file=real_stdout, flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
