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


def pair_graph(dust3r, frames, batch_size=2, scene_graph="complete"):
    """The DUSt3R pair graph offline_reconstruction starts from, built the way demo.py:100-117 does it with
    dust3r.image_pairs.make_pairs(symmetrize=True) + dust3r.inference.inference(): every frame becomes a view dict
    (img, true_shape, idx, instance), the pair list is symmetrised, and each batch of pairs is run in both orders
    (inference.py:27-37 make_batch_symmetric), so the result holds 4 entries per unordered pair.  Returns
    dict(view1, view2, pred1, pred2) with tensors on the CPU, lists chained."""
    views = [dict(img=f["img"], true_shape=torch.tensor(f["img"].shape[-2:])[None], idx=j, instance=str(j)) for j, f in enumerate(frames)]
    n = len(views)
    if scene_graph != "complete":
        raise NotImplementedError("pair_graph builds the complete graph (demo.py default)")
    pairs = [(i, j) for i in range(n) for j in range(i)]
    pairs += [(j, i) for i, j in pairs]

    def batch_of(ids):                                      # collate + interleave the two orders of every pair
        v1, v2 = [], []
        for a, b in ids:
            v1 += [views[a], views[b]]
            v2 += [views[b], views[a]]
        pack = lambda vs: dict(img=torch.cat([v["img"] for v in vs]), true_shape=torch.cat([v["true_shape"] for v in vs]),
                               idx=[v["idx"] for v in vs], instance=[v["instance"] for v in vs])
        return pack(v1), pack(v2)
    out = dict(view1=dict(idx=[], instance=[], true_shape=[]), view2=dict(idx=[], instance=[], true_shape=[]), pred1={}, pred2={})
    for c0 in range(0, len(pairs), batch_size):
        v1, v2 = batch_of(pairs[c0:c0 + batch_size])
        p1, p2 = dust3r(v1, v2)
        for name, v in (("view1", v1), ("view2", v2)):
            out[name]["idx"] += v["idx"]
            out[name]["instance"] += v["instance"]
            out[name]["true_shape"].append(v["true_shape"])
        for name, p in (("pred1", p1), ("pred2", p2)):
            for k, t in p.items():
                out[name].setdefault(k, []).append(t.detach().cpu())
    for name in ("view1", "view2"):
        out[name]["true_shape"] = torch.cat(out[name]["true_shape"])
    for name in ("pred1", "pred2"):
        out[name] = {k: torch.cat(v) for k, v in out[name].items()}
    return out
