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


def pair_indices(n, scene_graph="complete", prefilter=None, symmetrize=True):
    """dust3r/image_pairs.py:11-46 make_pairs on indices: the (i, j) list for n images in the reference's order.
    scene_graph: 'complete' | 'swin[-W]' (window W = 3 with loop closure) | 'oneref[-R]' (every image against image R) | 'prev'
    (every image against all earlier ones, earlier first); prefilter 'seqN' / 'cycN' keeps pairs at most N frames apart
    (cyclically for 'cyc'), applied after the symmetrisation as there."""
    pairs = []
    if scene_graph == "complete":
        pairs = [(i, j) for i in range(n) for j in range(i)]
    elif scene_graph.startswith("swin"):
        win = int(scene_graph.split("-")[1]) if "-" in scene_graph else 3
        ids = set()                                          # (the reference iterates this set: same construction, same order)
        for i in range(n):
            for j in range(1, win + 1):
                k = (i + j) % n
                ids.add((i, k) if i < k else (k, i))
        pairs = list(ids)
    elif scene_graph.startswith("oneref"):
        ref = int(scene_graph.split("-")[1]) if "-" in scene_graph else 0
        pairs = [(ref, j) for j in range(n) if j != ref]
    elif scene_graph.startswith("prev"):
        pairs = [(j, i) for i in range(1, n) for j in range(i)]
    else:
        raise ValueError("pair_indices: unknown scene_graph %r" % (scene_graph,))
    if symmetrize:
        pairs = pairs + [(b, a) for a, b in pairs]
    if isinstance(prefilter, str) and (prefilter.startswith("seq") or prefilter.startswith("cyc")):
        thr, cyc = int(prefilter[3:]), prefilter.startswith("cyc")
        m = max(max(e) for e in pairs) + 1 if pairs else 0   # (the reference counts images from the surviving edges)
        keep = []
        for a, b in pairs:
            d = abs(a - b)
            if cyc:
                d = min(d, abs(a + m - b), abs(a - m - b))
            if d <= thr:
                keep.append((a, b))
        pairs = keep
    return pairs


def pair_graph(dust3r, frames, batch_size=2, scene_graph="complete", prefilter=None):
    """The DUSt3R pair graph offline_reconstruction starts from, built the way demo.py:100-117 does it with
    dust3r.image_pairs.make_pairs(symmetrize=True) + dust3r.inference.inference(): every frame becomes a view dict
    (img, true_shape, idx, instance), the pair list (`pair_indices`: every scene_graph / prefilter of make_pairs) is symmetrised, and each batch of pairs is run in both orders
    (inference.py:27-37 make_batch_symmetric), so the result holds 4 entries per unordered pair.  Returns
    dict(view1, view2, pred1, pred2) with tensors on the CPU, lists chained."""
    views = [dict(img=f["img"], true_shape=torch.tensor(f["img"].shape[-2:])[None], idx=j, instance=str(j)) for j, f in enumerate(frames)]
    n = len(views)
    pairs = pair_indices(n, scene_graph, prefilter, symmetrize=True)

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

    Layout.  Parameters are grouped into buckets in REVERSE parameter order (the order a backward pass finishes them) and
    every bucket owns ONE flat fp32 buffer; each parameter's `.grad` IS a view of its bucket (`flat=True`, the default), so
    autograd accumulates straight into the buffer the collective runs on: no pack / unpack copies, one all-reduce and one
    scale per bucket.  A parameter starts at a multiple of `align` elements (default 1024) so that bucket-wide kernels (the
    multi-tensor AdamW, the gradient-norm reduction) can look up per-parameter constants per 1024-element chunk
    (`chunk_table`).  `zero_grad()` zeroes the buckets (never `set_to_none`: that would detach the views; a detached or
    replaced `.grad` is noticed at launch time and copied in, so foreign optimizers still work, just slower).
    Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU) and a ring all-reduce of S bytes over N ranks moves
    2 S (N-1)/N per link at ~2 (N-1) latency hops, so buckets are large (64 MB default: > 95 % of the bandwidth term at 8
    ranks) -- the NVSwitch-era 25 MB default of DDP is latency-dominated here.

    Ordering.  Collectives are matched across ranks by ISSUE ORDER, so buckets are launched strictly in bucket-index order
    on every rank: with `overlap=True` a post-accumulate-grad hook marks a bucket ready when its last gradient of this
    backward pass has arrived and launches every consecutive ready bucket from the first unlaunched one (a ready bucket
    behind an unready one waits, as in DDP); `finish()` launches the rest in order -- buckets holding parameters that got
    no gradient on THIS rank (find_unused_parameters=True semantics: they contribute zeros) -- waits, and divides by the
    world size.  Which parameters were used on ANY rank is exchanged as a bitmap (one small MAX all-reduce, as DDP does):
    `unused_everywhere()` lists the parameters an optimizer should skip (torch leaves their .grad None and skips them).
    With no process group (or world 1, unless `force=True`: single-GPU tests of the RCCL path) every call is a no-op."""

    def __init__(self, params, bucket_mb=64.0, group=None, overlap=False, flat=True, align=1024, force=False):
        self.params = [p for p in params if p.requires_grad]
        self.group, self.flat, self.align, self.force = group, flat, (align if flat else 1), force
        self.buckets, cur, size = [], [], 0
        cap = int(bucket_mb * (1 << 20)) // 4
        pad = lambda n: (n + self.align - 1) // self.align * self.align
        for p in reversed(self.params):
            if cur and size + pad(p.numel()) > cap:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += pad(p.numel())
        if cur:
            self.buckets.append(cur)
        self.offsets = []                               # per bucket: element offset of each parameter
        self.sizes = []
        for b in self.buckets:
            o, offs = 0, []
            for p in b:
                offs.append(o)
                o += pad(p.numel())
            self.offsets.append(offs)
            self.sizes.append(o)
        self._flat = [None] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._used = None
        self.capture = False         # True while train.TrainStep captures / replays the step as a hipGraph (no host transfers in start())
        self._used_work = None
        self.launched_in_backward = 0          # (statistics of the last step: buckets whose collective started from a hook)
        self._armed = False
        self._started = False
        self._next = 0
        self._pending = None
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._index_of = {id(p): j for j, p in enumerate(self.params)}
        self._touched = [False] * len(self.params)
        if flat:
            self._attach()
        if overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    # ------------------------------------------------------------------ flat buckets
    def _buffer(self, i):
        dev = self.buckets[i][0].device
        if self._flat[i] is None or self._flat[i].device != dev:
            self._flat[i] = torch.zeros(self.sizes[i], dtype=torch.float32, device=dev)
        return self._flat[i]

    def _view(self, i, j):
        p = self.buckets[i][j]
        o = self.offsets[i][j]
        return self._buffer(i)[o:o + p.numel()].view_as(p)

    def _attach(self):
        """(re)point every .grad at its bucket slice, keeping the values of gradients that already exist"""
        for i, b in enumerate(self.buckets):
            for j, p in enumerate(b):
                v = self._view(i, j)
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                    self._touched[self._index_of[id(p)]] = True
                p.grad = v

    def flat_buffers(self):
        """[(flat gradient buffer, [(parameter, element offset)])] per bucket, for bucket-wide kernels"""
        return [(self._buffer(i), list(zip(b, self.offsets[i]))) for i, b in enumerate(self.buckets)]

    def zero_grad(self):
        for i in range(len(self.buckets)):
            if self.flat:
                self._buffer(i).zero_()
            else:
                for p in self.buckets[i]:
                    p.grad = None
        self._touched = [False] * len(self.params)
        self._reset_pending()
        if self.flat:
            self._attach()

    # ------------------------------------------------------------------ collectives
    def active(self):
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return self.force or dist.get_world_size(self.group) > 1

    def _reset_pending(self):
        # direct-to-bucket bookkeeping of spann3r_amd.train (_expect / _contributed): contributions owed by a tape whose backward
        # never ran (skipped step, a forward under grad that missed the loss) must not silence the next step's hooks
        for p in self.params:
            p._sp3_pending = 0

    def prepare(self, arm=True):
        """arm the hooks for the backward pass that follows (overlap=True).  arm=False: a backward that only accumulates (gradient
        accumulation, spann3r/training.py:228-233) -- no collective starts from it, the window's last backward reduces the sums"""
        self._reset_pending()
        self._work = [None] * len(self.buckets)
        self._pending = [len(b) for b in self.buckets]
        self._next = 0
        self._started = False
        self.launched_in_backward = 0
        self._armed = bool(arm) and self.active()

    def untouched(self):
        """parameters that got no gradient on THIS rank since zero_grad() (one-rank counterpart of unused_everywhere())"""
        if not any(self._touched):
            return []
        return [p for p, t in zip(self.params, self._touched) if not t]

    def _on_grad(self, p):
        self._touched[self._index_of[id(p)]] = True
        if not self._armed:
            return
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        # strictly in bucket order: a ready bucket behind an unready one waits for it (or for finish())
        while self._next < len(self.buckets) and self._pending[self._next] <= 0:
            self._work[self._next] = self._launch(self._next)
            self._next += 1
            self.launched_in_backward += 1

    def _launch(self, i):
        flat = self._buffer(i)
        for j, p in enumerate(self.buckets[i]):
            v = self._view(i, j)
            if p.grad is None:
                if not self.flat:
                    v.zero_()
                else:
                    v.zero_()
                    p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():       # somebody replaced .grad (foreign zero_grad / optimizer): copy it in
                v.copy_(p.grad.reshape(p.shape))
                if self.flat:
                    p.grad = v
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def start(self):
        """launch, in bucket order, the all-reduce of every bucket that has not started yet (call right after backward)"""
        if not self.active() or self._started:
            return
        if not self._armed:
            self._work = [None] * len(self.buckets)
            self._next = 0
        self._started = True
        while self._next < len(self.buckets):
            self._work[self._next] = self._launch(self._next)
            self._next += 1
        self._armed = False
        # which parameters got a gradient on ANY rank (without hooks: whatever holds a non-None gradient counts as used)
        touched = self._touched
        if not any(touched):
            touched = [p.grad is not None for p in self.params]
        if self.capture:
            # inside a hipGraph capture (train.TrainStep(graph=True)): no host -> device upload, no read-back.  Every rank replays
            # the SAME captured kernel sequence, so the parameters a step touches are the same on every rank: the used-set is the
            # local one and needs no collective.
            self._touched_host = list(touched)
            self._used = self._used_work = None
            return
        used = torch.tensor([1 if t else 0 for t in touched], dtype=torch.int32, device=self.params[0].device)
        self._used = used
        self._used_work = dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.group, async_op=True)

    def finish(self):
        """launch what the hooks have not (buckets with parameters unused on this rank), wait for the collectives, average"""
        if not self.active():
            self._armed = False
            return
        self.start()                                      # (also the path of finish() without prepare() / start())
        inv = 1.0 / dist.get_world_size(self.group)
        for i, (w, bucket) in enumerate(zip(self._work, self.buckets)):
            w.wait()
            flat = self._flat[i]
            flat.mul_(inv)
            if not self.flat:
                for j, p in enumerate(bucket):
                    g = self._view(i, j)
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.copy_(g)
        if self._used_work is not None:
            self._used_work.wait()
        self._work = [None] * len(self.buckets)
        self._next = 0
        self._started = False

    def unused_everywhere(self):
        """parameters that received no gradient on any rank in the step just reduced (DDP leaves their .grad None)"""
        if self.capture and getattr(self, "_touched_host", None) is not None:
            return [p for p, t in zip(self.params, self._touched_host) if not t]
        if self._used is None:
            return []
        u = self._used.cpu().tolist()
        return [p for p, f in zip(self.params, u) if not f]

    def reduce(self):
        self.start()
        self.finish()
