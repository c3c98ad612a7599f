"""Drop-in for `spann3r.model` (reference: /root/reference/spann3r/model.py).

`Spann3R` keeps the reference's constructor, state-dict keys, `.dust3r`
attribute and `forward(frames, return_memory=False) -> (preds, preds_all[, sp_mem])` contract;
`SpatialMemory` keeps the reference's policy (working/long-term memory, similarity gate, prune)
but is a static-capacity device arena with LayerNorm applied once at write time.
All arithmetic runs in the HIP kernels of libspann3r_hip.so; there is no CPU / eager-torch path.
"""
import argparse
import collections
import os
import weakref

import torch
import torch.nn as nn

from . import ops
from .config import Spann3RConfig, FULL
from .engine import Engine
from .weights import param_spec, alias_of, synth_state_dict


# ----------------------------------------------------------------------------- parameter tree
def _build_param_tree(root: nn.Module, spec):
    """Registers one nn.Parameter per state-dict key under the same dotted path the reference's
    module tree produces (so load_state_dict / state_dict / parameters() behave identically)."""
    made = {}
    for key, shape in spec.items():
        parts = key.split(".")
        mod = root
        for name in parts[:-1]:
            if not hasattr(mod, name) or not isinstance(getattr(mod, name), nn.Module):
                mod.add_module(name, nn.Module())
            mod = getattr(mod, name)
        src = alias_of(key)
        if src is not None:                      # same Parameter object under a second name
            mod.register_parameter(parts[-1], made[src])
            continue
        prm = nn.Parameter(torch.empty(shape))
        mod.register_parameter(parts[-1], prm)
        made[key] = prm
    return made


class _Dust3RFacade(nn.Module):
    """`model.dust3r` (spann3r/model.py:222): holds the DUSt3R parameters under the reference's names and is callable
    like AsymmetricCroCo3DStereo.forward(view1, view2) -> (res1, res2) (dust3r/model.py:213-227), which is what
    dust3r.inference.inference() drives to build the pair graph for offline_reconstruction (demo.py:116-118)."""

    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner_ref", weakref.ref(owner))      # not a submodule: no registration cycle

    @staticmethod
    def _is_symmetrized(v1, v2):
        """dust3r/utils/misc.py:29-37: batch entries (2i, 2i+1) are the two orderings of one pair"""
        x, y = v1.get("instance"), v2.get("instance")
        if x is None or y is None or len(x) != len(y) or len(x) == 1 or len(x) % 2:
            return False
        return all(x[i] == y[i + 1] and x[i + 1] == y[i] for i in range(0, len(x), 2))

    @torch.no_grad()
    def forward(self, view1, view2):
        own = self._owner_ref()
        eng = own.engine
        img1, img2 = view1["img"], view2["img"]
        B = img1.shape[0]
        shape1, shape2 = own._true_shape(view1), own._true_shape(view2)
        p = own.cfg.patch

        def enc_pair(a, b):                                  # dust3r/model.py:156-165
            if a.shape[-2:] == b.shape[-2:]:
                f, _ = eng.encode_image(torch.cat((a, b), 0).float())
                fa, fb = f.chunk(2, 0)
                return fa.contiguous(), fb.contiguous()
            return eng.encode_image(a.float())[0], eng.encode_image(b.float())[0]
        if self._is_symmetrized(view1, view2):               # half of the batch is encoded (:176-180), then interleaved
            fa, fb = enc_pair(img1[::2], img2[::2])
            feat1 = torch.stack((fa, fb), 1).flatten(0, 1).contiguous()
            feat2 = torch.stack((fb, fa), 1).flatten(0, 1).contiguous()
        else:
            feat1, feat2 = enc_pair(img1, img2)
        g1 = (img1.shape[-2] // p, img1.shape[-1] // p)
        g2 = (img2.shape[-2] // p, img2.shape[-1] // p)
        dec1, dec2 = eng.decoder(feat1, feat2, B, g1[0], g1[1], g2[0], g2[1])
        res1 = own.downstream_head(dec1, shape1, 1)
        res2 = own.downstream_head(dec2, shape2, 2)
        res2["pts3d_in_other_view"] = res2.pop("pts3d")     # :226
        return res1, res2


class SpatialMemory:
    """Device-resident spatial memory (reference: spann3r/model.py:11-210).

    Bank layout per batch element (capacity `cap` rows, a multiple of 64, fixed for the sequence):
      k_raw [cap,1024] fp32 = reference mem_k            v_raw [cap,1024] fp32 = reference mem_v
      k_hat  fragment-order [cap,1024] wdt = gamma_q (.) LN_k(mem_k)   (W operand of the S GEMM; LN_q is folded into it)
      v_hat_t fragment-order [1024,cap] wdt = LN_v(mem_v)^T             (W operand of the P.V GEMM)
      s_bank, b_bank [cap] fp32 (fold constants of LN_q), mem_attn [cap] fp32, mem_count [cap] fp32
    One launch writes a frame (sp3_bank_write).  A read is two launches while the score matrix stays cache-sized (score
    GEMM with the softmax statistics in its epilogue; P.V GEMM that builds the thresholded probabilities on load and
    renormalises in its epilogue) plus the column sums for mem_attn; long banks materialise P (softmax launch, split-K P.V).
    """
    FUSED_READ_MAX = 2_000_000        # queries x bank tokens: fp32 scores <= 8 MB (L2 / MALL resident for the 32 column tiles)

    STATE_BUCKET = 2048               # device-state mode: a step's hipGraph serves every bank length up to the next multiple of this
                                      # (the score launch carries <= 2048 / 32 x 7 idle workgroups for it: ~1 us at 196 tokens)
    STATE_READ_MAX = 8192             # ... on the two-launch read, whose statistics merge holds <= 256 groups of 32 keys per row

    def __init__(self, engine: Engine, batch, num_patches, capacity, attn_thresh=5e-4, long_mem_size=4000,
                 work_mem_size=5, sim_thresh=0.95, device_state=False):
        """device_state (round 6; the static hipGraph runner in bf16 mode): the fill level (M, wm) also lives in a device int32 block
        that the step's kernels read -- score GEMM N, P.V GEMM K, the row a frame is written to, the working-memory window of the
        similarity gate -- so the captured step does not depend on it: one graph per step KIND (and grid bucket)
        instead of one per bank length.  The host still owns the policy and pushes the values after every change (_push_state)."""
        self.eng = engine
        self.B, self.P, self.C = batch, num_patches, engine.cfg.enc_dim
        self.attn_thresh = attn_thresh
        self.long_mem_size = long_mem_size
        self.work_mem_size = work_mem_size
        self.top_k = long_mem_size
        self.sim_thresh = sim_thresh
        self.num_patches = num_patches
        # (> 256 query rows: the long-bank read's score stage works in 128-key tiles -- capacity in whole tiles)
        self.cap = (capacity + 127) // 128 * 128 if num_patches > 256 else (capacity + 63) // 64 * 64
        dev, wdt = engine.device, engine.wdt
        self.kb = 64 if wdt == torch.bfloat16 else 32           # k-block of the MFMA dtype
        self._banks = [self._alloc(dev, wdt), None]   # second bank allocated on first prune
        self._cur = 0
        self.M = 0
        self.wm = 0
        self.lm = 0
        self.events = []
        self._score = torch.zeros(batch, max(work_mem_size, 1), device=dev)
        self._score_host, self._score_event, self._score_pending = None, None, False
        self._cos_scratch = torch.empty(max(work_mem_size, 1) * num_patches, device=dev)
        self._sel = torch.zeros(batch, max(long_mem_size, 1), dtype=torch.int32, device=dev)
        w = engine.w
        self._norms = (w["norm_k.w"], w["norm_k.b"], w["norm_v.w"], w["norm_v.b"], w["norm_q.w"], w["norm_q.b"])
        self._pending_attn = None
        self.state = torch.zeros(4, dtype=torch.int32, device=dev) if (device_state and wdt == torch.bfloat16 and ops.LEAN) else None
        self._state_pushed = (0, 0)
        # the read takes its extent from the device where its lean instances serve it (gemm_sm.hip sm_find / pv_ok: <= 256 query rows,
        # 1024-wide keys); elsewhere (512 x 512 frames) the read's launches keep the token count as an argument and stay keyed by it
        self.read_dyn = self.state is not None and num_patches <= 256 and self.C == 1024
        self._window_sel = {}            # (P, M) -> the selection [P, M) of the sliding-window policy (_drop_oldest)

    def _alloc(self, dev, wdt):
        B, cap, C = self.B, self.cap, self.C
        return dict(k_raw=torch.zeros(B, cap, C, device=dev), v_raw=torch.zeros(B, cap, C, device=dev),
                    k_hat=torch.zeros((B,) + ops.packed_shape(cap, C, wdt), dtype=wdt, device=dev),
                    v_hat_t=torch.zeros((B,) + ops.packed_shape(C, cap, wdt), dtype=wdt, device=dev),
                    s_bank=torch.zeros(B, cap, device=dev), b_bank=torch.zeros(B, cap, device=dev),
                    attn=torch.zeros(B, cap, device=dev), count=torch.zeros(B, cap, device=dev))

    def bank_bytes(self):
        return sum(sum(t.numel() * t.element_size() for t in bk.values()) for bk in self._banks if bk is not None)

    @property
    def bank(self):
        return self._banks[self._cur]

    def _bank_of(self, bk, b):
        return {k: v[b] for k, v in bk.items()}

    def reset(self):
        """Start a new sequence on the same arena (stale rows are never read: every read is bounded by M)."""
        self._cur, self.M, self.wm, self.lm, self.events = 0, 0, 0, 0, []
        self._pending_attn = None
        self._push_state()

    def _push_state(self):
        """device copy of (M, wm) <- the host's values, if they changed (one tiny launch; nothing waits for it but the next step)"""
        if self.state is not None and self._state_pushed != (self.M, self.wm):
            ops.bank_state_set(self.state, self.M, self.wm)
            self._state_pushed = (self.M, self.wm)

    def _bucket(self, step=None):
        """device-state mode: the token count the read's launches are SIZED for (grid, argument checks); the kernels take the real
        count from the device"""
        step = step or self.STATE_BUCKET
        return min(self.cap, (self.M + step - 1) // step * step)

    def graph_key(self):
        """what a captured step depends on: (M, wm) when the launches carry them as arguments; the grid bucket of the two-launch read
        in device-state mode (long banks: their launch list still depends on M)"""
        if self.state is None:
            return (self.M, self.wm)
        if self.M == 0:
            return ("s", 0)
        if self.read_dyn and self._read_plan()[1]:
            return ("s", self._bucket())
        if self._prob_read():
            return ("p", self._bucket(self.PROB_BUCKET))
        return ("l", self.M)

    PROB_BUCKET = 8192                # device-state mode, long banks: grid bucket of the score-matrix-free read (<= 64 idle N-tiles)

    def _prob_read(self):
        """Long bank, > 256 query rows (a 512 x 512 frame), bf16, no attention threshold (the growing-bank policy of BASELINE config 3):
        the read runs without a score matrix -- score stage writes bf16 p~ = exp(s - group max) + group statistics (lean tile 45), a
        small merge launch, P.V stage with the groups' rescale in its loop (tile 46), split-K reduce, column sums."""
        return (self.kb == 64 and ops.LEAN and self.attn_thresh == 0.0 and self.P > 256 and self.C == 1024 and not self._read_plan()[1])

    def snapshot(self):
        """Detached copy of the reference-visible state (what `return_memory=True` hands out)."""
        self._flush_attn()
        snap = MemorySnapshot()
        for name in ("mem_k", "mem_v", "mem_attn", "mem_count"):
            t = getattr(self, name)
            setattr(snap, name, None if t is None else t.clone())
        snap.wm, snap.lm, snap.M, snap.num_patches, snap.events = self.wm, self.lm, self.M, self.P, list(self.events)
        return snap

    # reference-compatible views (what return_memory=True exposes)
    @property
    def mem_k(self):
        return None if self.M == 0 else self.bank["k_raw"][:, :self.M]

    @property
    def mem_v(self):
        return None if self.M == 0 else self.bank["v_raw"][:, :self.M]

    @property
    def mem_attn(self):
        return None if self.M == 0 else self.bank["attn"][:, :self.M, None]

    @property
    def mem_count(self):
        return None if self.M == 0 else self.bank["count"][:, :self.M, None]

    # ------------------------------------------------------------------ read (:145-183)
    def memory_read(self, feat, out, feat_packed=None, feat_stats=None, defer_attn=False, out_packed=None):
        """feat fp32 [B,P,1024] (the query, feat_k2) -> out = attn . LN_v(mem_v) + feat ; mem_attn += colsum(attn).
        feat_packed / feat_stats: fragment-order copy and row-statistics partials of `feat` when its producer already
        wrote them (B == 1: the key-MLP GEMM's c2 / stats_out); otherwise one sp3_pack_stats launch makes them.
        defer_attn: the column sums of the two-launch read (mem_attn += ..., consumed by nothing before the next prune) are
        not launched here but folded into the launch that commits / drops the staged frame (`commit`, `finish_staged`) --
        a side stream for them costs more in graph fork/join edges than the launch itself (measured: -8 % frames/s).
        out_packed (B == 1, two-launch read): receives a fragment-order bf16 copy of `out` from the same epilogue (decoder_embed's A
        operand); returns whether it was written."""
        self._flush_attn()
        self.wrote_packed = False
        eng, bk = self.eng.activate(), self.bank
        B, P, C, M, kb = self.B, self.P, self.C, self.M, self.kb
        assert M > 0
        prof = ops._prof
        e0 = prof.region_begin() if prof is not None else None
        Kp = (M + kb - 1) // kb * kb
        ld = self.cap
        Pp = (P + 15) // 16 * 16
        if feat_packed is None or B > 1:
            qp = [eng.wsp("mem_q_packed%d" % b, P, C) for b in range(B)]
            qs = eng.ws("mem_q_stats", (B, P, C // 32, 2))
            for b in range(B):
                ops.pack_stats(feat[b], qp[b], qs[b], rows=P, C_=C)
        else:
            qp, qs = [feat_packed], feat_stats.view(1, P, C // 32, 2)
        alpha = 1.0 / (C ** 0.5)
        S_k, fused = self._read_plan()
        if not fused and self._prob_read():
            self._memory_read_prob(feat, out, qp, qs, alpha)
            return self._read_tail(feat, out, out_packed, prof, e0)
        S = eng.ws("mem_S", (B, P, ld))
        # Two launches while the bank is short (every per-frame read of the 224x224 demo): the score GEMM leaves the softmax
        # statistics of its 32-key groups behind, the P.V GEMM turns scores into thresholded probabilities as it loads
        # them and renormalises in its epilogue; the column sums (mem_attn, only read by the next prune) follow.
        # Long banks (split K; the scores no longer sit in L2 for 32 column tiles to re-read) keep a materialised P.
        st = eng.ws("mem_sm_stats", (B, P, (self.cap + 31) // 32, 2)) if fused else None
        zk = eng.ws("mem_sm_rowz", (B, P, 4)) if fused else None
        # device-state mode: the two launches are sized for the bucket and read the token count from the device (dyn_n)
        dyn = self.state if (fused and self.read_dyn) else None
        Mg = self._bucket() if dyn is not None else M
        for b in range(B):
            # S = LN_q(q) . K_hat^T / 32: raw q (fragment order) x (gamma_q (.) K_hat), LN_q folded through s_bank / b_bank
            ops.gemm(qp[b], ops.PackedWeight.wrap(bk["k_hat"][b], Mg, C), S[b], M=P, N=Mg, K=C, lda=C, ldc=ld, alpha=alpha,
                     bias=bk["b_bank"][b], ln=ops.LnFold(qs[b], C, bk["s_bank"][b], 1e-5), sm_stats_out=st[b] if fused else None, dyn_n=dyn)
        if fused:
            for b in range(B):
                c2 = out_packed if (out_packed is not None and B == 1) else None
                ops.gemm(S[b], ops.PackedWeight.wrap(bk["v_hat_t"][b], C, self.cap), out[b], M=P, N=C, K=Mg, lda=ld, ldc=C,
                         ldw=self.cap, res1=feat[b], ldr1=C, softmax=(st[b], self.attn_thresh, zk[b]), c2=c2, dyn_n=dyn)
                self.wrote_packed = c2 is not None
            self.note_deferred_read()
            if not defer_attn:
                self._flush_attn()
        else:
            pk = eng.ws("mem_P_packed", (B, Pp * self.cap), eng.adt, zero=True)
            ops.softmax_pack(S, pk, eng.ws("mem_rowstat", (B * P * 4,)), ld=ld, rows=P, M=M, thresh=self.attn_thresh, batch=B, strideS=P * ld,
                             stride_packed=pk.shape[1])
            for b in range(B):
                A = ops.PackedAct(P, Kp, eng.adt, eng.device, data=pk[b])
                Wv = ops.PackedWeight.wrap(bk["v_hat_t"][b], C, self.cap)
                if S_k == 1:
                    ops.gemm(A, Wv, out[b], M=P, N=C, K=Kp, lda=Kp, ldc=C, ldw=self.cap, res1=feat[b], ldr1=C)
                else:
                    part = eng.ws("mem_pv_partial", (S_k * P * C,))
                    ops.gemm(A, Wv, part, M=P, N=C, K=Kp, lda=Kp, ldc=C, ldw=self.cap, splitk=S_k)
                    ops.reduce_ln(part, S_k, P, C, res=feat[b], ldres=C, x_out=out[b], ldx=C)
                ops.colsum_packed(pk[b], P, M, bk["attn"][b])
            return self._read_tail(feat, out, out_packed, prof, e0)
        return self._read_tail(feat, out, None, prof, e0)

    def _read_tail(self, feat, out, out_packed, prof, e0):
        eng, bk = self.eng, self.bank
        B, P, C, M = self.B, self.P, self.C, self.M
        if out_packed is not None and B == 1:
            # the long-bank compositions end in fp32 rows: one small launch makes the fragment-order copy decoder_embed reads
            # (otherwise that GEMM and the key MLP's first layer fall back to fp32 rows on the general kernel)
            ops.pack_stats(out[0], out_packed, eng.ws("mem_out_stats", (P, C // 32, 2)), rows=P, C_=C)
            self.wrote_packed = True
        if prof is not None:
            es = bk["k_hat"].element_size()
            # algorithmic bytes of one read: K_hat + V_hat once, plus the query in and the fused features out
            prof.region_end("memread", e0, B * (2.0 * M * C * es + 2.0 * P * C * 4),
                            info={"M": M, "tokens_per_frame": P, "flops": 4.0 * B * P * M * C})
        return out

    def _memory_read_prob(self, feat, out, qp, qs, alpha):
        """the long-bank read without a score matrix (see _prob_read; include/spann3r_hip.h sp3_prob_merge)"""
        eng, bk = self.eng, self.bank
        B, P, C, M, cap = self.B, self.P, self.C, self.M, self.cap
        dyn = self.state
        Mg = self._bucket(self.PROB_BUCKET) if dyn is not None else M
        S_k = 8        # one slice per XCD (256 workgroups at 1024 rows: one round); 16 slices measured 4-20 % slower at 8 k .. 50 k tokens and double the reduce
        Pp = (P + 15) // 16 * 16
        rows_pad = (P + 255) // 256 * 256
        ngc = cap // 64
        pk = eng.ws("mem_P_packed", (B, Pp * cap), eng.adt, zero=True)
        stats = eng.ws("mem_prob_stats", (B, ngc * rows_pad * 2))
        scale = eng.ws("mem_prob_scale", (B, ngc * rows_pad))
        part = eng.ws("mem_pv_partial", (16 * P * C,))
        for b in range(B):
            pt = ops.PackedAct(P, cap, eng.adt, eng.device, data=pk[b])
            # stage 1: p~ = exp(s - group max), s = LN_q(q) . K_hat^T / 32 (LN_q folded through s_bank / b_bank), + (max, sum) per 64-key group
            ops.gemm(qp[b], ops.PackedWeight.wrap(bk["k_hat"][b], Mg, C), pt, M=P, N=Mg, K=C, lda=C, ldc=cap, alpha=alpha,
                     bias=bk["b_bank"][b], ln=ops.LnFold(qs[b], C, bk["s_bank"][b], 1e-5), sm_stats_out=stats[b], dyn_n=dyn)
            ops.prob_merge(stats[b], scale[b], P, Mg, cap, dyn_n=dyn)
            # stage 2: split-K partials of sum_g scale_g (p~_g . V_hat_g), then slices + q
            ops.gemm(pt, ops.PackedWeight.wrap(bk["v_hat_t"][b], C, cap), part, M=P, N=C, K=Mg, lda=cap, ldc=C, ldw=cap, splitk=S_k,
                     softmax=(scale[b], 0.0, None), dyn_n=dyn)
            ops.reduce_ln(part, S_k, P, C, res=feat[b], ldres=C, x_out=out[b], ldx=C)
            ops.colsum_prob(pk[b], scale[b], P, Mg, cap, bk["attn"][b], dyn_n=dyn)

    # ------------------------------------------------------------------ write (:80-95)
    def stage_write(self, feat_k, feat_v):
        """Speculatively writes frame (k, v) into rows [M, M+P) of the bank; `commit()` makes it visible.
        (A skipped frame simply never advances M, so the slot is overwritten by the next write.)"""
        bk = self.bank
        B, P, C, M = self.B, self.P, self.C, self.M
        assert M + P <= self.cap, "spatial memory capacity exceeded"
        for b in range(B):
            ops.bank_write(feat_k[b], feat_v[b], self._bank_of(bk, b), M, P, C, self.cap, self._norms, 1.0 / (C ** 0.5), state=self.state)

    def _read_plan(self):
        """(split-K factor of the P.V GEMM, two-launch read?) at the current bank length"""
        M, kb = self.M, self.kb
        Kp = (M + kb - 1) // kb * kb
        # long banks: K split over several workgroups per output tile (64 tiles otherwise), one reduce launch adds q
        S_k = 1
        while Kp // (2 * S_k) >= 2048 and S_k < 16:
            S_k *= 2
        fused = S_k == 1 and self.P * M <= self.FUSED_READ_MAX and M % 4 == 0
        if self.read_dyn:
            fused = fused and self._bucket() <= self.STATE_READ_MAX
        return S_k, fused

    def note_deferred_read(self):
        """A two-launch read of the current bank was issued (directly or by replaying the step's hipGraph, which runs no
        Python): its column sums are owed to mem_attn -- from the static score / row-statistic workspaces."""
        if self.M > 0 and self._read_plan()[1]:
            eng = self.eng
            self._pending_attn = (eng.ws("mem_S", (self.B, self.P, self.cap)), self.cap, self.M, eng.ws("mem_sm_rowz", (self.B, self.P, 4)))

    def _flush_attn(self, append=False):
        """launch the pending column sums of the last two-launch read (optionally with the append bookkeeping)"""
        pend, self._pending_attn = self._pending_attn, None
        if pend is None:
            return False
        S, ld, M, zk = pend
        bk = self.bank
        fuse = append and M == self.M
        for b in range(self.B):
            ops.colsum_softmax(S[b], ld, self.P, M, zk[b], self.attn_thresh, bk["attn"][b],
                               bk["count"][b] if fuse else None, self.P if fuse else 0)
        return fuse

    def commit(self):
        bk = self.bank
        if not self._flush_attn(append=True):
            for b in range(self.B):
                ops.mem_append(bk["count"][b], bk["attn"][b], self.M, self.P)
        self.M += self.P

    def add_mem(self, feat_k, feat_v):
        self.stage_write(feat_k, feat_v)
        self.commit()
        self._push_state()

    # ------------------------------------------------------------------ similarity gate (:97-118)
    def sim_scores(self, feat_k, to_host=False):
        """[B, wm] mean cosine similarity of feat_k against each of the last `wm` stored frames.
        to_host (single-graph step, device-resident bank state): the reduction kernel stores the scores STRAIGHT into the pinned host
        buffer the host polls (pinned memory is mapped into the device's address space) -- no copy node between the step's halves."""
        B, P, C = self.B, self.P, self.C
        n = self.wm * P
        for b in range(B):
            if self.state is not None:       # the window [M - wm P, M) comes from the device: the launch is sized for work_mem_size frames
                dst = self._host_scores()[b] if to_host else self._score[b]
                ops.cos_sim_state(feat_k[b], self.bank["k_raw"][b], max(self.work_mem_size, 1), P, C, self.state, dst, self._cos_scratch)
            else:
                ops.cos_sim(feat_k[b], self.bank["k_raw"][b, self.M - n:self.M], self.wm, P, C, self._score[b], self._cos_scratch)
        return self._score[:, :self.wm]

    def sim_needed(self):
        if self.M == 0 or self.sim_thresh == 1.0 or self.wm == 0:
            return False
        if self.wm * self.P > self.M:
            # the reference reshapes mem_k[:, -wm*P:] and raises here (SURVEY.md §7 quirk ii, 512x512 after a prune)
            raise RuntimeError("working memory (%d tokens) larger than the bank (%d): reference check_sim is undefined"
                               % (self.wm * self.P, self.M))
        return True

    def _host_scores(self):
        if self._score_host is None:
            self._score_host = torch.empty(self._score.shape, dtype=torch.float32).pin_memory()
            self._score_event = torch.cuda.Event()
        return self._score_host

    def fetch_scores_async(self):
        """queue the device->host copy of the scores behind the kernels launched so far; sim_verdict() waits for it only"""
        self._host_scores().copy_(self._score, non_blocking=True)
        self._score_event.record()
        self._score_pending = True

    def copy_scores_in_graph(self):
        """The same copy as a node INSIDE the step's hipGraph (single-graph step): no event can be recorded in the middle of a
        replay, so the host learns that the copy has landed from the data itself -- arm_score_poll() fills the pinned buffer with
        a sentinel no cosine can take (2.0; a NaN sentinel could not tell "not landed" from a genuinely NaN score of degenerate
        features) before the launch, sim_verdict() spins until the wm scores have replaced it."""
        self._host_scores().copy_(self._score, non_blocking=True)

    _SENTINEL = 2.0                       # outside [-1, 1]: never a mean cosine (NaN scores pass through: a NaN maximum compares False below)

    def arm_score_poll(self):
        self._host_scores().fill_(self._SENTINEL)
        self._score_pending = "poll"

    def sim_verdict(self):
        """Host side of check_sim: reads the scores the cos_sim kernels left in self._score (host sync, as :114)."""
        if self._score_pending == "poll":
            import time
            self._score_pending = False
            view = self._score_host[:, :self.wm]
            t0 = time.perf_counter()
            while bool((view == self._SENTINEL).any()):
                if time.perf_counter() - t0 > 0.25:          # (never observed: fall back to a stream sync)
                    torch.cuda.current_stream().synchronize()
                    if bool((view == self._SENTINEL).any()):  # (the copy-node form: the scores are in the device buffer)
                        view = self._score[:, :self.wm].cpu()
                    break
            mx = float(view.max())                 # (torch's max propagates NaN, as the reference's mean_corr.max() does: never 'similar')
        elif self._score_pending:
            self._score_event.synchronize()
            self._score_pending = False
            mx = float(self._score_host[:, :self.wm].max())
        else:
            mx = float(self._score[:, :self.wm].max())
        if mx > self.sim_thresh:
            print("Similarity detected:", mx)
            return True
        return False

    def check_sim(self, feat_k):
        if not self.sim_needed():
            return False
        self.sim_scores(feat_k)
        return self.sim_verdict()

    def add_mem_check(self, feat_k, feat_v):
        """Eval-mode write policy (:120-143)."""
        if self.check_sim(feat_k):
            self.events.append("skip")
            return
        self.add_mem(feat_k, feat_v)
        self._after_write()
        self._push_state()

    def finish_staged(self, similar):
        """Second half of add_mem_check when the similarity kernel and the speculative write were already launched
        (hipGraph path): commit or drop the staged frame, then the working/long-term bookkeeping and prune."""
        if similar:
            self.events.append("skip")
            self._flush_attn()
            return
        self.commit()
        self._after_write()
        self._push_state()

    def _after_write(self):
        self.wm += 1
        if self.wm > self.work_mem_size:
            self.wm -= 1
            if self.long_mem_size == 0:
                self._drop_oldest()                              # sliding window (spann3r/model.py:132-137)
            else:
                self.lm += self.P
        if self.lm > self.long_mem_size:
            self.memory_prune()
            self.lm = self.top_k - self.wm * self.P

    def _drop_oldest(self):
        """long_mem_size == 0 (spann3r/model.py:132-137): the bank is a sliding window of work_mem_size frames -- the oldest frame's
        P tokens leave.  P is not a multiple of the 16-row fragment blocks, so the fragment-order copies are re-gathered like a
        prune with the selection [P, M) (same kernels, other arena half)."""
        self._flush_attn()
        B, C, M, P = self.B, self.C, self.M, self.P
        k = M - P
        src = self.bank
        if self._banks[1 - self._cur] is None:
            self._banks[1 - self._cur] = self._alloc(self.eng.device, self.eng.wdt)
        dst = self._banks[1 - self._cur]
        sel = self._window_sel.get((P, M))
        if sel is None:
            sel = self._window_sel[(P, M)] = torch.arange(P, M, dtype=torch.int32, device=self.eng.device)
        for b in range(B):
            ops.gather_rows(src["k_raw"][b], dst["k_raw"][b], sel, k, C)
            ops.gather_rows(src["v_raw"][b], dst["v_raw"][b], sel, k, C)
            ops.gather_packed_rows(src["k_hat"][b], dst["k_hat"][b], sel, k, C)
            ops.gather_packed_cols(src["v_hat_t"][b], dst["v_hat_t"][b], sel, k, self.cap, C, self.cap)
            for name in ("s_bank", "b_bank", "attn", "count"):
                ops.gather_1d(src[name][b], dst[name][b], sel, k)
        print("Memory pruned:", torch.Size((B, k, C)))              # (spann3r/model.py:137 prints mem_k.shape)
        self.events.append("window %d->%d" % (M, k))
        self._cur = 1 - self._cur
        self.M = k

    # ------------------------------------------------------------------ prune (:185-210)
    def memory_prune(self):
        """Keep the top_k tokens by mem_attn/mem_count (tokens younger than work_mem_size+5 steps protected).
        The kept tokens are stored sorted by weight descending, ties by index ascending (the reference's torch.topk
        leaves the tie order implementation-defined, SURVEY.md §7 quirk i)."""
        self._flush_attn()
        B, C, M, k = self.B, self.C, self.M, self.top_k
        src = self.bank
        if self._banks[1 - self._cur] is None:
            self._banks[1 - self._cur] = self._alloc(self.eng.device, self.eng.wdt)
        dst = self._banks[1 - self._cur]
        for b in range(B):
            ops.prune_select(src["attn"][b], src["count"][b], M, self.work_mem_size + 5, k, self._sel[b])
            sel = self._sel[b]
            ops.gather_rows(src["k_raw"][b], dst["k_raw"][b], sel, k, C)
            ops.gather_rows(src["v_raw"][b], dst["v_raw"][b], sel, k, C)
            ops.gather_packed_rows(src["k_hat"][b], dst["k_hat"][b], sel, k, C)
            ops.gather_packed_cols(src["v_hat_t"][b], dst["v_hat_t"][b], sel, k, self.cap, C, self.cap)
            for name in ("s_bank", "b_bank", "attn", "count"):
                ops.gather_1d(src[name][b], dst[name][b], sel, k)
        print("Memory pruned:", M, "->", k)
        self.events.append("prune %d->%d" % (M, k))
        self._cur = 1 - self._cur
        self.M = k


class MemorySnapshot:
    """What forward(..., return_memory=True) returns: the reference SpatialMemory's visible fields."""
    mem_k = mem_v = mem_attn = mem_count = None
    wm = lm = M = 0


class _SequenceRunner:
    """Static-buffer, hipGraph-captured execution of Spann3R.forward for one (batch, H, W, policy) geometry.

    Every per-frame step is a fixed kernel sequence over persistent buffers; the only things that vary between steps
    of a sequence are the memory fill M, the working-memory count wm and the active bank, so graphs are cached by
    (M, wm, bank).  A key runs eagerly the first time it is seen (that run also creates every workspace buffer), is
    captured + replayed the second time, and replayed afterwards.  The host only touches the device between steps to
    read the similarity score (the reference's own host sync, spann3r/model.py:114) and to clone the outputs.
    """

    def __init__(self, model, eng, B, H, W, training, true_hw=None):
        cfg = model.cfg
        # (a weak reference: model -> runner -> model would be a cycle, and the hipGraphs of a dropped model would then be destroyed
        # whenever the cyclic collector runs -- inside some later model's capture or replay; round 6 saw that as a segfault in
        # hipGraphLaunch.  Without the cycle they go when the model goes.)
        self._model_ref = weakref.ref(model)
        self.eng, self.B, self.H, self.W, self.training = eng, B, H, W, training
        p = cfg.patch
        self.nh, self.nw = H // p, W // p                 # token raster of the image (PatchEmbedDust3R ignores true_shape)
        # The heads see the tokens as a (true_h/p, true_w/p) grid and portrait results come back axis-swapped
        # (transpose_to_landscape wrapper, dust3r/utils/misc.py:54-96 with landscape_only=True, spann3r/model.py:222).
        th, tw = true_hw or (H, W)
        self.hh, self.hw = th // p, tw // p
        self.swap = th > tw
        self.P, self.E = self.nh * self.nw, cfg.enc_dim
        dev = eng.device
        self.img_pair = torch.empty(2 * B, 3, H, W, device=dev)
        self.img_next = torch.empty(B, 3, H, W, device=dev)          # frame i+2, encoded one step ahead
        self.feat_pre = torch.empty(B, self.P, self.E, device=dev)
        self.featpair = torch.empty(2 * B, self.P, self.E, device=dev)
        self.feat1, self.feat2 = self.featpair[:B], self.featpair[B:]
        self.fuse = torch.empty(B, self.P, self.E, device=dev)
        # bf16 mode, whole-sequence encoder: fragment-order copies of the features (enc_norm writes them next to the fp32 rows) so
        # that decoder_embed and the key MLPs run on the lean small-M instances
        self.packed_feats = eng.packed_attn
        self.feats_p = None
        self.featpair_p = ops.PackedAct.group(2, B * self.P, self.E, torch.bfloat16, dev) if self.packed_feats else None
        self.fuse_p = ops.PackedAct(B * self.P, self.E, torch.bfloat16, dev) if self.packed_feats else None
        self.k1 = torch.empty(B, self.P, self.E, device=dev)
        self.k2 = torch.empty(B, self.P, self.E, device=dev)
        self.v = torch.empty(B, self.P, self.E, device=dev)
        self.mem = None
        self.k2_aux = (None, None)
        self.graphs = collections.OrderedDict()        # key -> (hipGraph, device bytes its capture allocated), least recently used first
        self.graph_bytes = 0
        self.seen = set()
        self.length_keys, self.length_keys_eager = set(), False    # keys that carry the bank length; True: they run eagerly (see _graphed)
        self.out = None
        self._scores_to_host = False  # set around _part1 by the single-graph step (SpatialMemory.sim_scores)
        self.batched = False          # True: the frames of the sequence were encoded up front (encode_sequence)
        self.img_all = self.feats = None
        self.defer2 = False           # True: the view-2 DPT head runs once for all steps after the loop (finish_head2)
        self.seq_dec2 = None          # per hook: [steps*B*P, D] copies of the side-2 decoder tokens of every step
        self.dec2_hooks = None
        self.head2_out = {}

    @property
    def model(self):
        m = self._model_ref()
        if m is None:
            raise RuntimeError("the Spann3R model this runner belongs to is gone")
        return m

    def ensure_memory(self, n_frames):
        need = (n_frames - 1) * self.P if self.training else 4000 + 8 * self.P
        if self.mem is None or self.mem.cap < need:
            self.mem = SpatialMemory(self.eng, self.B, self.P, capacity=need, attn_thresh=0.0 if self.training else 5e-4, device_state=True)
            self.graphs.clear()
            self.graph_bytes = 0
            self.seen.clear()
            self.length_keys, self.length_keys_eager = set(), False
        self.mem.reset()
        return self.mem

    # ---- whole-sequence encoder -------------------------------------------------------------------------------
    ENC_CHUNK_ROWS = 16          # images per encoder launch group (bounds the workspace for long sequences)

    def encode_sequence(self, frames, use_graphs):
        """The ViT-L encoder sees each frame exactly once and does not depend on the memory (spann3r/model.py:293-295),
        so for a sequence that is handed over as a whole all frames go through it together: M = n*B*P rows per GEMM
        instead of B*P, i.e. 64-row tiles and 5-10x fewer weight passes.  feats[i] is what the reference calls feat of
        frame i."""
        B, n = self.B, len(frames)
        if self.feats is None or self.feats.shape[0] < n * B:
            self.img_all = torch.empty(n * B, 3, self.H, self.W, device=self.eng.device)
            self.feats = torch.empty(n * B, self.P, self.E, device=self.eng.device)
            self.feats_p = ops.PackedAct.group(n, B * self.P, self.E, torch.bfloat16, self.eng.device) if self.packed_feats else None
            # every graph that baked in the old buffers dies with them: the encoder's AND the deferred head's (reads feats)
            self._drop_graphs(lambda k: k[0] in ("enc", "head2"))
            self.seen = {k for k in self.seen if k[0] not in ("enc", "head2")}
        slots = [(f["img"], self.img_all[i * B:(i + 1) * B]) for i, f in enumerate(frames)]
        fast = [p for p in slots if p[0].is_cuda and p[0].dtype == torch.float32 and p[0].is_contiguous() and p[0].shape == p[1].shape]
        for c0 in range(0, len(fast), 8):                      # 8 frames per launch instead of one copy each
            ops.copy_multi(fast[c0:c0 + 8])
        for src, dst in slots:
            if not any(src is q[0] for q in fast):
                dst.copy_(src)
        per = max(1, self.ENC_CHUNK_ROWS // B)
        for c0 in range(0, n, per):
            c1 = min(n, c0 + per)
            img, out = self.img_all[c0 * B:c1 * B], self.feats[c0 * B:c1 * B]
            outp = self.feats_p.at(c0) if self.feats_p is not None else None          # frame f -> group f of the packed copy
            self._graphed(("enc", c0, c1), lambda: self.eng.encode_image(img, out=out, tag="_seq", out_packed=outp, group_rows=B * self.P),
                          use_graphs)
        self.batched = True

    # ---- deferred view-2 head --------------------------------------------------------------------------------
    HEAD_CHUNK = 16              # images per deferred head launch group

    def enable_deferred_head2(self, n_frames):
        """The view-2 pointmap/confidence of a step is an output only (nothing of the recurrence reads it: the memory is
        written from view 1, spann3r/model.py:505-521), so the per-step DPT head 2 is replaced by ONE pass over all
        steps after the loop: B*(n-1) images per convolution instead of B."""
        rows = (n_frames - 1) * self.B * self.P
        D = self.model.cfg.dec_dim
        if self.seq_dec2 is None or self.seq_dec2[0].shape[0] < rows:
            self.seq_dec2 = [torch.empty(rows, D, device=self.eng.device) for _ in range(3)]
            self._drop_graphs(lambda k: k[0] == "head2")
            self.seen = {k for k in self.seen if k[0] != "head2"}
        self.defer2 = True

    def dec2_copies(self, i):
        """after step i: the three decoder hook outputs of side 2 go to their sequence slots (hook 0 is the encoder feature,
        already in feats) -> (src, dst) pairs for the step's bookkeeping launch"""
        BP = self.B * self.P
        return [(src, dst[i * BP:(i + 1) * BP]) for src, dst in zip(self.dec2_hooks, self.seq_dec2)]

    def finish_head2(self, n_frames, use_graphs):
        """-> list of (pts2, conf2) per step, fresh tensors"""
        B, P, hk = self.B, self.P, self.model.cfg.hooks
        steps = n_frames - 1
        per = max(1, self.HEAD_CHUNK // B)
        outs = []
        for c0 in range(0, steps, per):
            c1 = min(steps, c0 + per)
            dec = [None] * (hk[-1] + 1)
            dec[hk[0]] = self.feats[(c0 + 1) * B:(c1 + 1) * B]                   # frames c0+1 .. c1 are the view-2 frames
            for j, t in enumerate(self.seq_dec2):
                dec[hk[j + 1]] = t[c0 * B * P:c1 * B * P]
            key = ("head2", c0, c1)

            def fn(dec=dec, key=key, nb=(c1 - c0) * B):
                pts, conf, _ = self.eng.dpt_head(dec, nb, self.hh, self.hw, 2)
                self.head2_out[key] = (pts, conf)
            self._graphed(key, fn, use_graphs)
            pts, conf = self.head2_out[key]
            pts, conf = pts.clone(), conf.clone()
            for j in range(c1 - c0):
                outs.append((pts[j * B:(j + 1) * B], conf[j * B:(j + 1) * B]))
        return outs

    def begin_outputs(self, n_steps):
        """The results of a sequence are handed out as views into ONE fresh allocation per tensor kind (made here, once per forward
        call), not four torch.empty per step: the caller still owns fresh tensors, the caching allocator is out
        of the step loop."""
        self._out_slabs, self._out_steps, self._out_i = None, n_steps, 0
        # step graphs that carry the bank length (fp32-operand modes): decided for the WHOLE sequence before its first step -- if one
        # graph (or two, two-graph steps) per step does not fit next to the encoder / head graphs, every step launches eagerly
        if self.mem is not None and self.mem.state is None and not self.length_keys_eager:
            per_step = 1 if self.model.single_graph_step else 2
            if n_steps * per_step > self.MAX_GRAPHS - 4:
                self.length_keys_eager = True
                self._drop_graphs(lambda k: k in self.length_keys)

    def _step_outputs(self, srcs):
        if getattr(self, "_out_steps", 0) <= 0 or self._out_i >= self._out_steps:
            return [torch.empty_like(t) for t in srcs]               # (a step outside a begin_outputs() window)
        if self._out_slabs is None or len(self._out_slabs) != len(srcs):
            self._out_slabs = [t.new_empty((self._out_steps,) + tuple(t.shape)) for t in srcs]
        i, self._out_i = self._out_i, self._out_i + 1
        return [slab[i] for slab in self._out_slabs]

    def pair_copies(self, i):
        """featpair <- (feat of frame i, feat of frame i+1): two adjacent slabs of the sequence buffer, one copy (plus the same for
        the fragment-order bf16 copies)"""
        B = self.B
        pairs = [(self.feats[i * B:(i + 2) * B], self.featpair)]
        if self.feats_p is not None:
            st = self.feats_p.stride
            pairs.append((self.feats_p.data.view(-1)[i * st:(i + 2) * st], self.featpair_p.data.view(-1)[:2 * st]))
        return pairs

    def load_pair(self, i):
        ops.copy_multi(self.pair_copies(i))

    def _drop_graphs(self, pred):
        for k in [k for k in self.graphs if pred(k)]:
            self.graph_bytes -= self.graphs.pop(k)[1]

    MAX_GRAPHS = 16                      # per geometry.  bf16: the bank's fill level is device state, a 50-frame growing-bank sequence holds
                                         # ~15 (encoder / head chunks, one step graph per grid bucket); round 5 held ~100 (one per bank length)
    MAX_GRAPH_BYTES = 1 << 30            # device memory the captures of one geometry may pin in their pool

    def _graphed(self, key, fn, use_graphs, per_length=False):
        """eager the first time a key is seen (creates the workspaces), captured the second time, replayed afterwards.
        The captured graphs are an LRU bounded in count and in the device bytes their captures allocated (all captures of a
        runner share one memory pool); an evicted key is simply captured again when it comes back.
        per_length: the key carries the bank length (the fp32-operand modes, whose launches take it as an argument).  Such steps are
        captured only if ALL of a sequence's fit next to the runner's other graphs: a long sequence would otherwise evict and
        re-capture one graph per step, which costs more than launching eagerly (these modes are GPU-bound: eager launches keep up).
        The decision is per runner, never per step (`length_keys_eager`): one sequence is either replayed or launched eagerly
        throughout."""
        if per_length and use_graphs and not self.length_keys_eager and key not in self.graphs and key in self.seen \
                and len(self.graphs) >= self.MAX_GRAPHS - 2:
            # no room for this sequence's per-length steps (two slots stay free for the encoder / deferred-head graphs): from
            # here on they all run eagerly, and the ones captured so far go
            self.length_keys_eager = True
            self._drop_graphs(lambda k: k in self.length_keys)
        if per_length:
            self.length_keys.add(key)
        if not use_graphs or (per_length and self.length_keys_eager):
            fn()
        elif key in self.graphs:
            self.graphs.move_to_end(key)
            self.graphs[key][0].replay()
        elif key in self.seen:
            if getattr(self, "_pool", None) is None or not self.graphs:
                # a pool whose last graph is gone is retired by the allocator (its blocks are freed at the next empty_cache / OOM
                # retry) and must not be captured into again: a runner that dropped all of its graphs starts a new pool
                self._pool = torch.cuda.graph_pool_handle()
            if getattr(self, "_capture_stream", None) is None:
                self._capture_stream = torch.cuda.Stream()
            before = torch.cuda.memory_allocated()
            g = torch.cuda.CUDAGraph()
            # capture_begin / capture_end directly instead of the torch.cuda.graph context manager, which opens with a device
            # synchronise and torch.cuda.empty_cache(): a capture records launches, it does not need an idle GPU -- the eager step
            # queued before it keeps running while the host records this one (a first forward call captures its step graph between
            # two steps), and the allocator keeps the blocks the next call's outputs will want.
            # (no cyclic garbage collection during the capture: finalising another runner's hipGraph is not permitted while a
            # stream captures)
            import gc
            gc_was = gc.isenabled()
            gc.disable()
            main = torch.cuda.current_stream()
            cap = self._capture_stream
            cap.wait_stream(main)
            try:
                with torch.cuda.stream(cap):
                    g.capture_begin(self._pool, capture_error_mode="thread_local")
                    try:
                        fn()
                    finally:
                        g.capture_end()
            finally:
                if gc_was:
                    gc.enable()
            main.wait_stream(cap)
            nbytes = max(0, torch.cuda.memory_allocated() - before)
            self.graphs[key] = (g, nbytes)
            self.graph_bytes += nbytes
            while len(self.graphs) > self.MAX_GRAPHS or (self.graph_bytes > self.MAX_GRAPH_BYTES and len(self.graphs) > 1):
                _, (old, ob) = self.graphs.popitem(last=False)
                self.graph_bytes -= ob
                del old
            g.replay()
        else:
            self.seen.add(key)
            fn()

    # ---- the kernel sequences (spann3r/model.py:485-531) ----------------------------------------------------
    def _part1(self, first, has_next):
        """Everything up to the similarity scores: (memory read) + decoder + key MLPs + check_sim's cosine scores.
        The two decoder sides run on two streams (or as grouped launches); a third stream encodes the NEXT frame in
        the frame-by-frame mode (each frame is encoded exactly once, spann3r/model.py:293-295)."""
        eng, mem, B, P, E = self.eng, self.mem, self.B, self.P, self.E
        main = torch.cuda.current_stream()
        st = eng.side_streams()
        # fragment-order bf16 copies of the features exist when the sequence was encoded up front (encode_sequence)
        packed = self.packed_feats and self.model.packed_features and self.batched and self.feats_p is not None
        p1 = p2 = f1p = None
        if packed:
            p1, p2 = self.featpair_p.at(0), self.featpair_p.at(1)
        if first:
            if not self.batched:
                eng.encode_image(self.img_pair, out=self.featpair)
            f1, f1p = self.feat1, p1
        else:
            if not self.batched:
                ops.copy2d(self.feat2, E, self.feat1, E, B * P, E)      # feat1 <- previous feat2 (:294)
                ops.copy2d(self.feat_pre, E, self.feat2, E, B * P, E)   # feat2 <- the frame encoded during the previous step
            # reads k2 (and its fragment-order copy / statistics) before the key MLP of this step overwrites them
            mem.memory_read(self.k2, self.fuse, *(self.k2_aux if B == 1 else (None, None)), defer_attn=True,
                            out_packed=self.fuse_p if packed else None)
            f1 = self.fuse
            f1p = self.fuse_p if (packed and mem.wrote_packed) else None
        if has_next:
            st[3].wait_stream(main)
            with torch.cuda.stream(st[3]):
                eng.encode_image(self.img_next, out=self.feat_pre, tag="_pre")
        if eng.packed_attn and self.model.grouped_decoder:
            # both decoder sides and both key MLPs as grouped launches on the main stream: no per-layer fork/join
            dec1, dec2 = eng.decoder_grouped(f1, self.feat2, B, self.nh, self.nw, f1p=f1p, f2p=p2 if f1p is not None else None)
            self.k2_aux = eng.encode_feat_keys_grouped(self.feat1, self.feat2, dec1[-1], dec2[-1], B * P, self.k1, self.k2,
                                                       feat1p=p1, feat2p=p2)
        else:
            if self.model.decoder_streams:
                dec1, dec2 = eng.decoder(f1, self.feat2, B, self.nh, self.nw, self.nh, self.nw, streams=st)   # joined on return
                st[2].wait_stream(main)
                with torch.cuda.stream(st[2]):
                    self.k2_aux = eng.encode_feat_key(self.feat2, dec2[-1], B * P, 2, self.k2, aux=True)
                eng.encode_feat_key(self.feat1, dec1[-1], B * P, 1, self.k1)
                main.wait_stream(st[2])
            else:
                # one stream, side after side: a graph fork/join pair costs more than the overlap of two 196-row launches buys
                dec1, dec2 = eng.decoder(f1, self.feat2, B, self.nh, self.nw, self.nh, self.nw, streams=None)
                self.k2_aux = eng.encode_feat_key(self.feat2, dec2[-1], B * P, 2, self.k2, aux=True)
                eng.encode_feat_key(self.feat1, dec1[-1], B * P, 1, self.k1)
        if not self.training and mem.sim_needed():
            mem.sim_scores(self.k1, to_host=self._scores_to_host and mem.state is not None)
        if has_next:
            main.wait_stream(st[3])
        self.dec = (dec1, dec2)
        self.dec2_hooks = [dec2[h] for h in self.model.cfg.hooks[1:]]

    def _part2(self):
        """DPT head(s) + value encoder + staged memory write (spann3r/model.py:505-521).  Nothing in here feeds the
        similarity gate, so the host reads the scores of _part1 while this part runs."""
        eng, mem, B = self.eng, self.mem, self.B
        dec1, dec2 = self.dec
        main = torch.cuda.current_stream()
        st = eng.side_streams()
        pts2 = conf2 = None
        if not self.defer2:
            st[2].wait_stream(main)
            with torch.cuda.stream(st[2]):
                pts2, conf2, _ = eng.dpt_head(dec2, B, self.hh, self.hw, 2)
        pts1, conf1, _ = eng.dpt_head(dec1, B, self.hh, self.hw, 1)
        # portrait results are handed on axis-swapped (landscape_only wrapper); the value encoder sees that view
        if eng.cfg.use_feat:                    # spann3r/model.py:312-314: the value comes from dec1[-1], not from the pointmap
            eng.encode_cur_value_feat(dec1[-1], self.v, self.k1,
                                      pos32=eng.positions(B, self.nh, self.nw)[1] if eng.cfg.mem_pos_enc else None)
        else:
            eng.encode_cur_value(pts1.swapaxes(1, 2) if self.swap else pts1, self.v, self.k1)   # v = cur_v + feat_k1
        mem.stage_write(self.k1, self.v)
        if not self.defer2:
            main.wait_stream(st[2])
        self.out = (pts1, conf1, pts2, conf2)

    def run(self, first, has_next, use_graphs, post_copies=()):
        """post_copies: (src, dst) pairs issued with the step's own result copies in ONE launch after the second graph
        (decoder hooks -> sequence slots, the next pair of encoder features): six eager launches per frame became one"""
        mem = self.mem
        has_next = has_next and not self.batched                # nothing to prefetch: the sequence is already encoded
        gk = mem.graph_key()
        per_length = gk[0] not in ("s", "p")               # (the bank length itself is in the key: see _graphed)
        key = gk + (mem._cur, has_next, self.batched, self.defer2, self.model.grouped_decoder, self.model.packed_features)
        if not use_graphs and ops._prof is not None:
            ops._prof.step_begin()
        # two graphs per step: the host fetches the similarity scores (async copy + event) as soon as the first one is
        # done and takes the memory decision while the second one (head, value encoder) still runs -> the next step's
        # launches are queued before the GPU runs dry.  The reference syncs at the same point of the data flow (:114).
        need_sim = not self.training and mem.sim_needed()
        if self.model.single_graph_step and use_graphs and ops._prof is None:
            # ONE graph per step: the score copy is a memcpy node between the two halves and the host polls the pinned buffer for
            # it while the second half runs (two graph launches per step left ~15 us of launch gaps around the copy)
            if need_sim:
                mem.arm_score_poll()

            def whole():
                self._scores_to_host = need_sim                 # (device-resident bank state: cos_mean stores into the pinned buffer itself)
                try:
                    self._part1(first, has_next)
                finally:
                    self._scores_to_host = False
                if need_sim and mem.state is None:
                    mem.copy_scores_in_graph()
                self._part2()
            self._graphed(("whole", first, need_sim) + key, whole, use_graphs, per_length)
            if not first:
                mem.note_deferred_read()        # (a replayed graph ran no Python)
        else:
            self._graphed(("first" if first else "step",) + key, lambda: self._part1(first, has_next), use_graphs, per_length)
            if not first:
                mem.note_deferred_read()        # (a replayed graph ran no Python)
            if need_sim:
                mem.fetch_scores_async()
            self._graphed(("tail",) + key, self._part2, use_graphs, per_length)
        pts1, conf1, pts2, conf2 = self.out
        srcs = (pts1, conf1) + (() if pts2 is None else (pts2, conf2))
        outs = self._step_outputs(srcs)
        copies = list(zip(srcs, outs))
        copies += list(post_copies() if callable(post_copies) else post_copies)     # (callable: the hook buffers exist only now)
        for c0 in range(0, len(copies), 8):
            ops.copy_multi(copies[c0:c0 + 8])
        res1 = {"pts3d": outs[0], "conf": outs[1]}
        res2 = {} if pts2 is None else {"pts3d": outs[2], "conf": outs[3]}                 # {}: filled by finish_head2
        if self.swap:                                           # landscape_only wrapper (dust3r/utils/misc.py:79-80)
            res1 = {k: v.swapaxes(1, 2) for k, v in res1.items()}
            res2 = {k: v.swapaxes(1, 2) for k, v in res2.items()}
        # memory policy (:518-521): the frame was staged inside the step; commit or drop it here
        if self.training:
            mem.commit()
            mem._push_state()
        else:
            mem.finish_staged(mem.sim_verdict() if need_sim else False)
        return res1, res2


class Spann3R(nn.Module):
    """MI355X drop-in for spann3r.model.Spann3R (spann3r/model.py:213-539)."""

    def __init__(self, dus3r_name="./checkpoints/DUSt3R_ViTLarge_BaseDecoder_512_dpt.pth",
                 use_feat=False, mem_pos_enc=False, memory_dropout=0.15, cfg: Spann3RConfig = None,
                 init_weights=True):
        super().__init__()
        self.use_feat = use_feat
        self.mem_pos_enc = mem_pos_enc
        # spann3r/model.py:248: only its .training flag and p matter for the forward-only build
        self.mem_dropout = nn.Dropout(memory_dropout)
        ckpt = None
        if dus3r_name is not None and not os.path.isfile(dus3r_name):
            # the reference fails here too (load_model / hub download, dust3r/model.py:83-90); dus3r_name=None is the
            # explicit "no checkpoint" switch (seeded synthetic weights if init_weights, else uninitialised for load_state_dict)
            raise FileNotFoundError("DUSt3R checkpoint %r not found (pass dus3r_name=None to build without one)" % dus3r_name)
        if dus3r_name is not None:
            # dust3r/model.py:27-51: the checkpoint carries its own constructor string
            torch.serialization.add_safe_globals([argparse.Namespace])
            ckpt = torch.load(dus3r_name, map_location="cpu", weights_only=True)
            cfg = Spann3RConfig.from_ctor_string(ckpt["args"].model)
        self.cfg = cfg or FULL
        if mem_pos_enc != self.cfg.mem_pos_enc or use_feat != self.cfg.use_feat:
            import dataclasses
            self.cfg = dataclasses.replace(self.cfg, mem_pos_enc=bool(mem_pos_enc), use_feat=bool(use_feat))
        self.add_module("dust3r", _Dust3RFacade(self))       # the parameter tree below hangs the DUSt3R weights into it
        self._params = _build_param_tree(self, param_spec(self.cfg))
        # no network: without a checkpoint file the weights are seeded synthetic ones
        if init_weights:
            with torch.no_grad():
                for k, v in synth_state_dict(0, self.cfg).items():
                    if alias_of(k) is None:
                        self._params[k].copy_(v)
        if ckpt is not None:
            sd = {("dust3r." + k): v for k, v in ckpt["model"].items()}
            if not any(k.startswith("dust3r.dec_blocks2") for k in sd):          # dust3r/model.py:94-101
                for k, v in list(sd.items()):
                    if k.startswith("dust3r.dec_blocks."):
                        sd[k.replace("dust3r.dec_blocks.", "dust3r.dec_blocks2.")] = v
            own = self.state_dict()
            # dust3r/model.py:47-49 prints the strict=False result; same report here
            missing = [k for k in own if k.startswith("dust3r.") and k not in sd and alias_of(k) is None]
            unexpected = [k for k in sd if k not in own]
            print("<checkpoint %s: missing keys %s, unexpected keys %s>" % (dus3r_name, missing, unexpected))
            with torch.no_grad():
                for k, v in sd.items():
                    if k in own:
                        own[k].copy_(v)
                # spann3r/model.py:241-242: pos_patch_embed starts as a copy of the DUSt3R patch embed
                if not self.cfg.use_feat:
                    self._params["pos_patch_embed.proj.weight"].copy_(self._params["dust3r.patch_embed.proj.weight"])
                    self._params["pos_patch_embed.proj.bias"].copy_(self._params["dust3r.patch_embed.proj.bias"])
        self.precision = "fp32"
        self._engine = None
        self._engine_key = None
        self._pinned = None
        self._runners = {}
        self.use_graphs = True       # capture each per-frame step in a hipGraph (False: same kernels, eager launches)
        self.batch_encode = True     # forward(): encode all frames of the sequence together (False: frame by frame)
        self.defer_head2 = True      # with batch_encode: run the view-2 DPT head once for all steps after the loop
        self.grouped_decoder = True  # bf16: the two decoder sides as grouped launches on one stream (False: two streams)
        self.single_graph_step = True   # one hipGraph per step, the similarity scores polled from pinned memory (False: two graphs + an event)
        self.packed_features = True  # bf16 + batch_encode + grouped_decoder: decoder_embed / key MLPs read fragment-order bf16 copies of the features (lean instances)
        self.decoder_streams = True  # ungrouped decoder (the fp32-operand modes): two streams with a fork/join per layer (False: one stream, side after side)
        self.force_general = False   # True: always take the reference-shaped eager loop (_forward_general; tests)
        self.max_runners = 4         # geometries (batch, H, W, policy, true_shape) kept with their buffers and graphs

    # ------------------------------------------------------------------ engine management
    f16x3_range_guard = True              # 'f16x3': raise if an activation left the fp16 range (non-finite outputs), see _forward_inference

    def set_precision(self, precision):
        """'fp32': fp32 operands, fp32 MFMA (exact fp32 products; the strictest parity mode).  'f32x3': fp32 operands, every
        GEMM product through three bf16 MFMAs of a (hi, lo) split (16 mantissa bits per product -- TF32, which the reference
        enables on its own GPUs, has 10 -- fp32 accumulation).  'f32x6': three-way split, six bf16 MFMAs per product: fp32-grade
        products (24 operand bits) at bf16 MFMA speed -- the fast PARITY mode: it holds the 1e-3 bar also on trained-like weight
        statistics, where 'f32x3' does not (tests/test_model_gpu.py stress fixture).  'bf16': bf16 operands (benchmark mode)."""
        assert precision in ("fp32", "f32x3", "f32x6", "f16x3", "bf16")
        self.precision = precision
        return self

    def _versions(self):
        return tuple(p._version for p in self._params.values())

    @property
    def engine(self) -> Engine:
        # every entry point (forward, the reference-shaped stage methods, offline_reconstruction, model.dust3r) fetches the
        # engine first; the product mode of the fp32 GEMMs is the ENGINE's (Engine.activate: per engine and per thread, so two
        # models of different precision in one process never share it)
        if self._pinned is not None:          # inside forward(): weights cannot change, skip the version scan
            return self._pinned.activate()
        dev = self._params["norm_q.weight"].device
        if dev.type != "cuda":
            raise RuntimeError("spann3r_amd.Spann3R runs on an MI355X only: call .to('cuda') first "
                               "(there is no CPU fallback; the CPU reference lives in oracle/ for tests)")
        # (ops.WEIGHTS_EPOCH: optimizers that update through raw pointers / a replayed hipGraph do not move the version counters)
        key = (dev, self.precision, self._versions(), ops.WEIGHTS_EPOCH)
        if self._engine is None or self._engine_key != key:
            self._engine = Engine(self.cfg, dict(self._params), dev, self.precision)
            self._engine_key = key
            self._runners = {}
        return self._engine.activate()

    def _runner(self, key, make):
        """LRU over geometries: a runner owns the static buffers, the memory arena and the hipGraphs of one geometry."""
        run = self._runners.pop(key, None)
        if run is None:
            run = make()
            while len(self._runners) >= self.max_runners:
                self._runners.pop(next(iter(self._runners)))
        self._runners[key] = run
        return run

    # ------------------------------------------------------------------ reference-shaped stage methods
    @staticmethod
    def _true_shape(view):
        img = view["img"]
        return view.get("true_shape", torch.tensor(img.shape[-2:])[None].repeat(img.shape[0], 1))

    def _tag_grid(self, pos, img):
        """remember the token raster a position tensor was built for (saves decode() a device read of pos.max())"""
        pos._sp3_grid = (img.shape[-2] // self.cfg.patch, img.shape[-1] // self.cfg.patch)
        return pos

    def encode_image(self, view):                                           # spann3r/model.py:263-270
        img = view["img"]
        feat, pos = self.engine.encode_image(img.float())
        return feat, self._tag_grid(pos[:], img), self._true_shape(view)

    def encode_image_pairs(self, view1, view2):                             # :272-287
        img = torch.cat((view1["img"], view2["img"]), dim=0).float()
        feat, pos = self.engine.encode_image(img)
        f1, f2 = feat.chunk(2, dim=0)
        p1, p2 = pos.chunk(2, dim=0)
        return (f1.contiguous(), f2.contiguous(), self._tag_grid(p1, img), self._tag_grid(p2, img),
                self._true_shape(view1), self._true_shape(view2))

    def encode_frames(self, view1, view2, feat1, feat2, pos1, pos2, shape1, shape2):   # :289-297
        if feat1 is None:
            return self.encode_image_pairs(view1, view2)
        feat1, pos1, shape1 = feat2, pos2, shape2
        feat2, pos2, shape2 = self.encode_image(view2)
        return feat1, feat2, pos1, pos2, shape1, shape2

    def decode(self, feat1, pos1, feat2, pos2):                             # :322-325
        """RoPE positions are the token raster of the IMAGE (pos comes from the patch embed, which ignores true_shape)."""
        B = feat1.shape[0]
        g1, g2 = self._grid_from_pos(pos1), self._grid_from_pos(pos2)
        return self.engine.decoder(feat1, feat2, B, g1[0], g1[1], g2[0], g2[1])

    @staticmethod
    def _grid_from_pos(pos):
        g = getattr(pos, "_sp3_grid", None)
        if g is not None:
            return g
        return int(pos[0, :, 0].max()) + 1, int(pos[0, :, 1].max()) + 1

    def encode_feat_key(self, feat1, feat2, num=1):                         # :299-303
        B, P = feat1.shape[:2]
        out = torch.empty(B, P, self.cfg.enc_dim, device=feat1.device)
        return self.engine.encode_feat_key(feat1, feat2, B * P, num, out)

    def downstream_head(self, dec, true_shape, num=1):                      # :327-331 + dust3r/utils/misc.py:54-96
        hs, ws = true_shape[:, 0], true_shape[:, 1]
        B = dec[-1].shape[0]
        H, W = int(true_shape[0, 0]), int(true_shape[0, 1])
        p = self.cfg.patch
        if not bool((true_shape == true_shape[0:1]).all()):
            # a batch that mixes landscape views and portraits the dataset rotated to landscape (dust3r/utils/misc.py:80-94): the
            # head runs once per orientation on its share of the batch -- portraits on the (W, H) token grid, results transposed --
            # and the results are scattered back into batch order
            H, W = int(true_shape.min()), int(true_shape.max())
            land = (ws >= hs).to(dec[-1].device)
            if not bool(((hs == H) & (ws == W) | (hs == W) & (ws == H)).all()):
                raise ValueError("true_shape: one image size per batch, in either orientation (got %s)" % true_shape.tolist())
            res = None
            for mask, (gh, gw), swap in ((land, (H, W), False), (~land, (W, H), True)):
                nb = int(mask.sum())
                if nb == 0:
                    continue
                sel = [d[mask].contiguous() for d in dec]
                pts, conf, _ = self.engine.dpt_head(sel, nb, gh // p, gw // p, num)
                part = {"pts3d": pts.clone(), "conf": conf.clone()}
                if swap:
                    part = {k: v.swapaxes(1, 2) for k, v in part.items()}
                if res is None:
                    res = {k: v.new_empty((B,) + tuple(v.shape[1:])) for k, v in part.items()}
                for k in res:
                    res[k][mask] = part[k]
            return res
        pts, conf, _ = self.engine.dpt_head(dec, B, H // p, W // p, num)
        res = {"pts3d": pts.clone(), "conf": conf.clone()}
        if bool((ws < hs).all()):        # landscape_only wrapper: portrait results come back axis-swapped
            res = {k: v.swapaxes(1, 2) for k, v in res.items()}
        return res

    def encode_cur_value(self, res1, dec1, pos1, shape1, add=None):         # :312-320
        if self.cfg.use_feat:                                               # :313-314: value from the last decoder output
            tok = dec1[-1]
            out = torch.empty(tok.shape[0], tok.shape[1], self.cfg.enc_dim, device=tok.device)
            pos32 = None
            if self.cfg.mem_pos_enc:                                        # pos1: int64 [B, P, 2] token positions (:313)
                pos32 = pos1.reshape(-1, 2).to(torch.int32).to(tok.device).contiguous()
            return self.engine.encode_cur_value_feat(tok, out, add, pos32=pos32)
        pts = res1["pts3d"]
        B = pts.shape[0]
        P = (pts.shape[1] // self.cfg.patch) * (pts.shape[2] // self.cfg.patch)
        out = torch.empty(B, P, self.cfg.enc_dim, device=pts.device)
        return self.engine.encode_cur_value(pts, out, add)

    # ------------------------------------------------------------------ offline reconstruction (:333-471)
    @staticmethod
    def find_initial_pair(graph, n_frames):
        """spann3r/model.py:333-358: the pair of the DUSt3R pair graph with the largest mean sigmoid-confidence."""
        view1, view2, pred1, pred2 = graph["view1"], graph["view2"], graph["pred1"], graph["pred2"]
        conf_matrix = torch.zeros(n_frames, n_frames)
        for i in range(len(view1["idx"])):
            c1, c2 = pred1["conf"][i].float(), pred2["conf"][i].float()
            conf_matrix[int(view1["idx"][i]), int(view2["idx"][i])] = float(((c1 - 1) / c1).mean() + ((c2 - 1) / c2).mean())
        flat = int(conf_matrix.argmax())
        pair_idx = (flat // n_frames, flat % n_frames)
        print("init pair:%s, conf: %s" % (pair_idx, float(conf_matrix.max())))
        return pair_idx

    NBV_CHUNK = 16       # candidate views decoded together per launch group

    def find_next_best_view(self, frames, idx_todo, feat_fuse, pos1, shape1, feats=None):
        """spann3r/model.py:360-392.  The reference encodes, decodes and regresses every remaining view one by one in
        every round (O(n^2) encoder passes); here each frame's encoder features are computed once (`feats` cache) and the
        candidates of a round go through the pair decoder and both heads as ONE batch (side 1 = the fused features
        repeated).  Returns the reference's tuple for the view with the largest mean sigmoid-confidence."""
        eng = self.engine
        feats = {} if feats is None else feats
        B = feat_fuse.shape[0]
        best = None
        todo = list(idx_todo)
        g1 = self._grid_from_pos(pos1)
        per = self.NBV_CHUNK if B == 1 else 1
        for c0 in range(0, len(todo), per):
            ids = todo[c0:c0 + per]
            for i in ids:
                if i not in feats:
                    f, pos, shp = self.encode_image(frames[i])
                    feats[i] = (f.clone(), pos, shp)
            f2 = torch.cat([feats[i][0] for i in ids], 0)
            f1 = feat_fuse.expand(len(ids), -1, -1).contiguous() if B == 1 else feat_fuse
            g2 = self._grid_from_pos(feats[ids[0]][1])
            nb = f2.shape[0]
            dec1, dec2 = eng.decoder(f1, f2, nb, g1[0], g1[1], g2[0], g2[1])
            sh1 = shape1.expand(nb, -1) if B == 1 else shape1
            sh2 = torch.cat([torch.as_tensor(feats[i][2]).reshape(-1, 2) for i in ids], 0)
            res1 = self.downstream_head(dec1, sh1, 1)
            res2 = self.downstream_head(dec2, sh2, 2)
            c1, c2 = res1["conf"], res2["conf"]
            score = (((c1 - 1) / c1).flatten(1).mean(1) + ((c2 - 1) / c2).flatten(1).mean(1)) if B == 1 else \
                (((c1 - 1) / c1).mean() + ((c2 - 1) / c2).mean()).reshape(1)
            score = score.cpu()
            for j, i in enumerate(ids):
                sc = float(score[j])
                if best is None and sc <= 0.0:
                    continue                                # the reference starts from best_conf = 0.0
                if best is None or sc > best[0]:
                    sl = slice(j, j + 1) if B == 1 else slice(None)
                    best = (sc, i, [d[sl].clone() for d in dec1], [d[sl].clone() for d in dec2],
                            {k: v[sl].clone() for k, v in res1.items()}, {k: v[sl].clone() for k, v in res2.items()})
        if best is None:
            raise RuntimeError("find_next_best_view: no candidate with positive confidence (the reference fails here too)")
        sc, i, d1, d2, r1, r2 = best
        return i, d1, d2, r1, r2, feats[i][0], feats[i][1], feats[i][2], sc

    @torch.no_grad()
    def offline_reconstruction(self, frames, graph):
        """spann3r/model.py:394-471: start from the most confident pair of the DUSt3R pair graph, then repeatedly read the
        memory, pick the next best view among the remaining frames, and write it.  Returns (preds, preds_all, idx_used)."""
        self._pinned = None
        eng = self.engine
        self._pinned = eng
        try:
            n_frames = len(frames)
            idx_todo, idx_used = list(range(n_frames)), []
            img0 = frames[0]["img"]
            B = img0.shape[0]
            p = self.cfg.patch
            P = (img0.shape[-2] // p) * (img0.shape[-1] // p)
            sp_mem = SpatialMemory(eng, B, P, capacity=4000 + 8 * P)
            pair_idx = self.find_initial_pair(graph, n_frames)
            f1, f2 = frames[pair_idx[0]], frames[pair_idx[1]]
            for i in pair_idx:
                idx_used.append(i)
                idx_todo.remove(i)
            feat1, feat2, pos1, pos2, shape1, shape2 = self.encode_image_pairs(f1, f2)
            feats = {pair_idx[0]: (feat1, pos1, shape1), pair_idx[1]: (feat2, pos2, shape2)}
            dec1, dec2 = self.decode(feat1, pos1, feat2, pos2)
            res1 = self.downstream_head(dec1, shape1, 1)
            res2 = self.downstream_head(dec2, shape2, 2)
            feat_k2, preds, preds_all = None, None, []
            while True:
                if feat_k2 is not None:
                    feat1, pos1, shape1 = feat2, pos2, shape2
                    feat_fuse = sp_mem.memory_read(feat_k2, torch.empty_like(feat1))
                    id_n, dec1, dec2, res1, res2, feat2, pos2, shape2, best_conf = \
                        self.find_next_best_view(frames, idx_todo, feat_fuse, pos2, shape2, feats)
                    idx_todo.remove(id_n)
                    idx_used.append(id_n)
                    print("next best view: %d, conf: %s" % (id_n, best_conf))
                feat_k1 = self.encode_feat_key(feat1, dec1[-1], 1)
                feat_k2 = self.encode_feat_key(feat2, dec2[-1], 2)
                cur_v = self.encode_cur_value(res1, dec1, pos1, shape1, add=feat_k1)       # = cur_v + feat_k1
                sp_mem.add_mem_check(feat_k1, cur_v)
                res2["pts3d_in_other_view"] = res2.pop("pts3d")
                if preds is None:
                    preds = [res1]
                else:
                    res1["pts3d_in_other_view"] = res1.pop("pts3d")
                    preds.append(res1)
                preds_all.append((res1, res2))
                if len(idx_todo) == 0:
                    break
            preds.append(res2)
            return preds, preds_all, idx_used
        finally:
            self._pinned = None

    # ------------------------------------------------------------------ forward (:473-539)
    def forward(self, frames, return_memory=False):
        if self.training and self.mem_dropout.training and (torch.is_grad_enabled() or self.mem_dropout.p > 0):
            # train mode (spann3r/model.py:474-475: attn_thresh = 0, dropout on the read, unconditional add_mem) through the
            # autograd-recording forward of spann3r_amd/train.py -- HIP forward AND backward kernels.  With gradients enabled it is
            # a training step (spann3r/training.py:216); under torch.no_grad() with active memory dropout (a validation pass that left
            # the model in train mode) the same ops run without a tape: the forward-only runner has no dropout mask.
            # (model.train() with model.mem_dropout.eval() keeps selecting the forward-only growing-bank policy below.)
            if not frames[0]["img"].is_cuda:
                raise RuntimeError("spann3r_amd runs on an MI355X (HIP kernels); there is no CPU path -- move the frames to the GPU")
            from . import train as T
            P = {k: v for k, v in self.state_dict(keep_vars=True).items() if v.is_floating_point()}
            return T.forward_train(P, frames, self.cfg, dropout_p=self.mem_dropout.p, return_memory=return_memory)
        with torch.no_grad():
            return self._forward_inference(frames, return_memory)

    def _forward_inference(self, frames, return_memory):
        self._pinned = None
        eng = self.engine
        self._pinned = eng
        try:                                 # (the engine fetch above set the product mode of the fp32 GEMMs from self.precision)
            out = self._forward(eng, frames, return_memory)
        finally:
            self._pinned = None
        if self.precision == "f16x3" and self.f16x3_range_guard:
            # the fp16 planes of the f16x3 products hold |x| < 65504: a larger activation converts to inf and
            # the products to NaN -- loud in the numbers, and made loud HERE: one device reduction over the last frame's outputs, read
            # back once per sequence (the sequence has already synchronised on its similarity verdicts)
            last = out[0][-1]
            ok = torch.isfinite(last["conf"]).all() & torch.isfinite(next(v for k, v in last.items() if k.startswith("pts3d"))).all()
            if not bool(ok):
                raise FloatingPointError("f16x3: non-finite outputs -- an activation left the fp16 range (|x| >= 65504) of the split "
                                         "products; run this checkpoint with set_precision('f32x6') or 'fp32'")
        return out

    def _uniform_true_hw(self, frames):
        """(true_h, true_w) if every frame carries the same image shape and the same true_shape for the whole batch
        (absent = the image shape, spann3r/model.py:266), else None.  Every reference caller supplies `true_shape`
        (dust3r/datasets/base/base_stereo_view_dataset.py:89, demo.py:109, eval.py:101), normally as a CPU int32
        tensor equal to the image shape, or its transpose for portrait images the dataset rotated to landscape (:215-220)."""
        img0 = frames[0]["img"]
        H, W = int(img0.shape[-2]), int(img0.shape[-1])
        p = self.cfg.patch
        if H % p or W % p or any(tuple(f["img"].shape) != tuple(img0.shape) for f in frames):
            return None
        ts = [f["true_shape"] for f in frames if f.get("true_shape") is not None]
        if not ts:
            return H, W
        try:
            t = torch.stack([torch.as_tensor(x).reshape(-1, 2) for x in ts])            # [n, B, 2]
        except RuntimeError:
            return None
        if t.is_cuda:
            t = t.cpu()                                      # one host read per forward; the callers keep it on the CPU
        th, tw = int(t[0, 0, 0]), int(t[0, 0, 1])
        if not bool((t == t[0, 0]).all()) or t.shape[1] not in (1, img0.shape[0]):
            return None
        if len(ts) != len(frames) and (th, tw) != (H, W):
            return None                                      # frames without the key default to the image shape
        if (th, tw) not in ((H, W), (W, H)):
            return None
        return th, tw

    def _forward(self, eng, frames, return_memory):
        n = len(frames)
        if n < 2:
            raise ValueError("need at least two frames")
        hw = None if self.force_general else self._uniform_true_hw(frames)
        if hw is not None:
            return self._forward_static(eng, frames, return_memory, hw)
        return self._forward_general(eng, frames, return_memory)

    def _forward_static(self, eng, frames, return_memory, true_hw):
        """Same-shape sequences with one true_shape (what demo / eval / app / training batches are): static buffers,
        whole-sequence encoder, two hipGraphs per step."""
        img0 = frames[0]["img"]
        B, _, H, W = img0.shape
        key = (B, H, W, bool(self.training), true_hw)
        run = self._runner(key, lambda: _SequenceRunner(self, eng, B, H, W, bool(self.training), true_hw))
        mem = run.ensure_memory(len(frames))
        preds, preds_all = None, []
        n = len(frames)
        run.begin_outputs(n - 1)
        if self.batch_encode and n > 2:
            run.encode_sequence(frames, self.use_graphs)
            if self.defer_head2:
                run.enable_deferred_head2(n)
            else:
                run.defer2 = False
        else:
            run.batched = run.defer2 = False
        for i in range(n - 1):
            if run.batched:
                if i == 0:
                    run.load_pair(0)
            else:
                if i == 0:
                    run.img_pair[:B].copy_(frames[0]["img"])
                    run.img_pair[B:].copy_(frames[1]["img"])
                if i + 2 < n:
                    run.img_next.copy_(frames[i + 2]["img"])
            has_next = i + 2 < n
            def post(i=i):
                # after this step's kernels: the next step's pair of encoder features, side 2's hook outputs to their slots
                nxt = run.pair_copies(i + 1) if (run.batched and i + 1 < n - 1) else []
                return nxt + (run.dec2_copies(i) if run.defer2 else [])
            res1, res2 = run.run(i == 0, has_next, self.use_graphs, post)
            if not run.defer2:
                res2["pts3d_in_other_view"] = res2.pop("pts3d")                  # :523
            if preds is None:
                preds = [res1]
            else:
                res1["pts3d_in_other_view"] = res1.pop("pts3d")
                preds.append(res1)
            preds_all.append((res1, res2))
        if run.defer2:
            for (_, r2), (pts2, conf2) in zip(preds_all, run.finish_head2(n, self.use_graphs)):
                if run.swap:                                    # landscape_only wrapper, as in _SequenceRunner.run
                    pts2, conf2 = pts2.swapaxes(1, 2), conf2.swapaxes(1, 2)
                r2["pts3d_in_other_view"], r2["conf"] = pts2, conf2
        preds.append(res2)
        if return_memory:
            return preds, preds_all, mem.snapshot()
        return preds, preds_all

    def _forward_general(self, eng, frames, return_memory):
        """Reference-shaped loop over the stage methods (mixed shapes / explicit true_shape): same kernels, eager."""
        n = len(frames)
        img0 = frames[0]["img"]
        B = img0.shape[0]
        p = self.cfg.patch
        P = (img0.shape[-2] // p) * (img0.shape[-1] // p)
        if self.training:
            sp_mem = SpatialMemory(eng, B, P, capacity=(n - 1) * P, attn_thresh=0.0)
        else:
            sp_mem = SpatialMemory(eng, B, P, capacity=4000 + 8 * P)
        feat1 = feat2 = pos1 = pos2 = shape1 = shape2 = None
        feat_k2 = None
        preds, preds_all = None, []
        for i in range(n - 1):
            view1, view2 = frames[i], frames[i + 1]
            feat1, feat2, pos1, pos2, shape1, shape2 = self.encode_frames(view1, view2, feat1, feat2, pos1, pos2,
                                                                          shape1, shape2)
            if feat_k2 is not None:
                feat_fuse = sp_mem.memory_read(feat_k2, torch.empty_like(feat1))
            else:
                feat_fuse = feat1
            dec1, dec2 = self.decode(feat_fuse, pos1, feat2, pos2)
            feat_k1 = self.encode_feat_key(feat1, dec1[-1], 1)
            feat_k2 = self.encode_feat_key(feat2, dec2[-1], 2)
            res1 = self.downstream_head(dec1, shape1, 1)
            res2 = self.downstream_head(dec2, shape2, 2)
            cur_v = self.encode_cur_value(res1, dec1, pos1, shape1, add=feat_k1)     # = cur_v + feat_k1
            if self.training:
                sp_mem.add_mem(feat_k1, cur_v)
            else:
                sp_mem.add_mem_check(feat_k1, cur_v)
            res2["pts3d_in_other_view"] = res2.pop("pts3d")
            if preds is None:
                preds = [res1]
            else:
                res1["pts3d_in_other_view"] = res1.pop("pts3d")
                preds.append(res1)
            preds_all.append((res1, res2))
        preds.append(res2)
        if return_memory:
            return preds, preds_all, sp_mem
        return preds, preds_all
