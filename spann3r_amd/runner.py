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


class GradReducer:
    """Gradient averaging across the data-parallel ranks, the part of torch DistributedDataParallel that training.py:322-325
    relies on (one process per GPU; backend "nccl" is RCCL over xGMI, "gloo" in the CPU tests).

    Gradients are packed into flat fp32 buckets in REVERSE parameter order (the order a backward pass finishes them) and
    each bucket is all-reduced with async_op=True as soon as it is packed, so the collectives of the early buckets overlap
    the packing of the later ones (and, on the GPU, run on RCCL's own stream next to the backward kernels); `finish()`
    waits, scales by 1/world and scatters the averages back into `.grad`.  Bucket size: xGMI is point-to-point (7 links x
    ~153 GB/s per GPU) and a ring all-reduce of S bytes over N ranks moves 2 S (N-1)/N per link at ~2 (N-1) latency hops, so
    buckets are large (64 MB default: > 95 % of the bandwidth term at 8 ranks) -- the NVSwitch-era 25 MB default of DDP
    is latency-dominated here.  Parameters without a gradient contribute zeros (find_unused_parameters=True semantics).
    With no process group initialised every call is a no-op (single-GPU runs).

    overlap=True registers a post-accumulate-grad hook on every parameter: `prepare()` before `backward()` arms the buckets, a
    bucket's all-reduce is launched from inside the backward pass the moment its last gradient has been accumulated (the
    collective of the last layers runs while the first layers are still being differentiated), and `finish()` after
    `backward()` launches whatever is left (buckets holding unused parameters), waits and writes the means back."""

    def __init__(self, params, bucket_mb=64.0, group=None, overlap=False):
        import torch
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.buckets, cur, size = [], [], 0
        cap = int(bucket_mb * (1 << 20)) // 4
        for p in reversed(self.params):
            if cur and size + p.numel() > cap:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)
        self._work = []
        self._torch = torch
        self.launched_in_backward = 0          # (statistics of the last step: buckets whose collective started from a hook)
        self._armed = False
        self._pending = None
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        if overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def prepare(self):
        """arm the hooks for the backward pass that follows (overlap=True)"""
        self._work = [None] * len(self.buckets)
        self._pending = [len(b) for b in self.buckets]
        self.launched_in_backward = 0
        self._armed = self.active()

    def _on_grad(self, p):
        if not self._armed:
            return
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._work[i] = self._launch(i)
            self.launched_in_backward += 1

    def _launch(self, i):
        import torch.distributed as dist
        torch = self._torch
        bucket = self.buckets[i]
        n = sum(p.numel() for p in bucket)
        dev = bucket[0].device
        if self._flat[i] is None or self._flat[i].device != dev:
            self._flat[i] = torch.empty(n, dtype=torch.float32, device=dev)
        flat, o = self._flat[i], 0
        for p in bucket:
            v = flat[o:o + p.numel()]
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad.reshape(-1))
            o += p.numel()
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def active(self):
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def start(self):
        """pack + launch the all-reduces of every bucket that has not started yet (call right after backward)"""
        if not self.active():
            return
        if not self._armed:
            self._work = [None] * len(self.buckets)
        for i in range(len(self.buckets)):
            if self._work[i] is None:
                self._work[i] = self._launch(i)
        self._armed = False

    def finish(self):
        """launch what the hooks have not (unused parameters), wait for the collectives, write the averaged gradients back"""
        if not self.active():
            self._work, self._armed = [], False
            return
        if self._armed:
            self.start()
        if not self._work or any(w is None for w in self._work):
            return
        import torch.distributed as dist
        torch = self._torch
        inv = 1.0 / dist.get_world_size(self.group)
        for w, flat, bucket in zip(self._work, self._flat, self.buckets):
            w.wait()
            o = 0
            for p in bucket:
                g = flat[o:o + p.numel()].view_as(p) * inv
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                o += p.numel()
        self._work = []

    def reduce(self):
        self.start()
        self.finish()
