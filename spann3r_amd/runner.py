"""Sequence-level driver shared by bench.py, demo_loop.py and the multi-process tests.

Independent video sequences share nothing, so the multi-GPU path is: rank r takes
sequences {s : s mod world == r}, no data-path collective, and ONE all_gather of a small stats
record per rank at the end (RCCL over xGMI on GPUs; gloo in the CPU tests).
"""
import time

import torch
import torch.distributed as dist

from .weights import synth_frames


def shard(n_items, rank, world):
    """Indices of the sequences rank `rank` owns."""
    return list(range(rank, n_items, world))


def make_sequence(seq_id, n_frames, h, w, batch=1, device=None):
    frames = synth_frames(n_frames, h, w, batch=batch, seed=1000 + seq_id)
    if device is not None:
        frames = [{"img": f["img"].to(device)} for f in frames]     # demo.py:94-95 moves only 'img'
    return frames


def run_sequences(forward_fn, sequences, sync=None):
    """Times forward_fn over the given sequences exactly like demo.py:123-129 (plus the device sync the
    reference forgets).  Returns (frames, seconds, last_outputs)."""
    if sync:
        sync()
    t0 = time.perf_counter()
    frames = 0
    out = None
    for seq in sequences:
        out = forward_fn(seq)
        frames += len(seq)
    if sync:
        sync()
    return frames, time.perf_counter() - t0, out


def gather_stats(frames, seconds, extra=(), device="cpu"):
    """all_gather of [frames, seconds, *extra] (float64) -> tensor [world, 2+len(extra)].
    Works without an initialised process group (world 1)."""
    rec = torch.tensor([float(frames), float(seconds)] + [float(x) for x in extra], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return rec[None].cpu()
    outs = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, rec)
    return torch.stack(outs).cpu()


def aggregate(stats):
    """Whole-job throughput from the gathered records: total frames / max over ranks of seconds."""
    frames = float(stats[:, 0].sum())
    seconds = float(stats[:, 1].max())
    return frames / seconds, frames, seconds
