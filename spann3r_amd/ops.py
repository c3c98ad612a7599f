"""Operator-level host wrappers: torch tensors in, HIP kernels (via the C-ABI) do the work.

Each function mirrors one ATen-level operator of the reference's hot path (citations in
include/spann3r_hip.h).  Tensors are only used as typed device pointers; nothing here computes
with torch.  All launches go to torch's CURRENT stream, so the whole per-frame step can be
captured in a hipGraph with torch.cuda.graph().
"""
import ctypes as C

import torch

from . import lib as L
from .lib import GemmDesc, F32, BF16, ACT_NONE, ACT_GELU, ACT_RELU  # noqa: F401


# ----------------------------------------------------------------------------- optional per-launch profiler
_prof = None


class Profiler:
    """Brackets every launch with HIP events on the launch stream (torch's current stream) and books the
    ALGORITHMIC flops / bytes of the launch (bench.py's `roofline` object is computed from this)."""

    def __init__(self, blocker_ms=12.0):
        self.rec = []
        self.regions = []
        self.blocker_cycles = int(blocker_ms * 2.4e6)

    def step_begin(self):
        """Park the GPU on a spin kernel while the host enqueues the whole step, so that the event brackets below
        measure back-to-back kernel execution instead of host launch gaps (eager launches are host-bound)."""
        torch.cuda._sleep(self.blocker_cycles)

    def region_begin(self):
        return (self.begin(), len(self.rec))

    def region_end(self, key, e0, nbytes, info=None):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        e0, n0 = e0
        self.regions.append((key, e0, e1, float(nbytes), info, len(self.rec) - n0))

    def region_summary(self):
        """a region spans several bracketed launches: its own bracket and every inner one cost `bracket_ms` of GPU time
        each (calibrate()), which is subtracted; `raw_ms` keeps the uncorrected span"""
        torch.cuda.synchronize()
        ov = getattr(self, "bracket_ms", 0.0)
        out = []
        for k, e0, e1, b, i, inner in self.regions:
            raw = e0.elapsed_time(e1)
            out.append(dict(key=k, ms=max(raw - ov * (inner + 1), 1e-4), raw_ms=raw, launches=inner, bytes=b, info=i))
        return out

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, key, e0, flops, nbytes):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.rec.append((key, e0, e1, float(flops), float(nbytes)))

    def calibrate(self, n=40):
        """Cost of one event bracket AROUND A RUNNING KERNEL: brackets a spin kernel that reports its own duration
        (sp3_spin, 100 MHz device counter); cost = elapsed(bracket) - kernel duration, median over n launches.  (An EMPTY
        bracket overstates it -- two back-to-back event packets serialise -- which made round 1's per-kernel times ~2 us
        too short against rocprofv3.)  summary() subtracts it from every bracket."""
        ticks = torch.zeros(n, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        self.step_begin()
        ev = []
        for i in range(n):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(L.load().sp3_spin(12000, ticks[i:].data_ptr(), L.stream_ptr()), "sp3_spin")
            e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        dur_ms = (ticks.cpu().double() * 1e-5).tolist()            # 10 ns ticks -> ms
        t = sorted(a.elapsed_time(b) - d for (a, b), d in zip(ev, dur_ms))
        self.bracket_ms = max(t[len(t) // 2], 0.0)
        return self.bracket_ms

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        ov = getattr(self, "bracket_ms", 0.0)
        for key, e0, e1, fl, by in self.rec:
            a = agg.setdefault(key, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1
            a["ms"] += max(e0.elapsed_time(e1) - ov, 1e-4)
            a["flops"] += fl
            a["bytes"] += by
        return agg


def set_profiler(p):
    global _prof
    _prof = p


_TILE_NAMES = {0: "32x32xk4", 1: "64x64", 2: "64x128", 3: "64x64xk4", 4: "32x64xk4", 18: "16x64xk4", 5: "128x128lds", 6: "128x64lds", 7: "112x64lds",
               8: "208x64lds", 9: "112x32lds", 10: "112x64lds2w", 11: "112x64lds8w", 12: "208x64lds8w", 13: "112x64wreg8",
               14: "112x64wreg4", 15: "208x64wreg8", 16: "112x64wreg4l6", 17: "112x64wreg8l4",
               20: "256x128pipe", 21: "128x128pipe", 22: "128x64pipe", 23: "64x64pipe", 24: "64x32pipe", 25: "32x32pipe",
               # lean small-M instances (csrc/gemm_sm.hip): epilogue, K, tile, waves over K
               30: "lean-rope-k1024-48x64xk8", 31: "lean-rope-k768-32x32xk4", 32: "lean-packed-k1024-64x64xk8", 33: "lean-packed-k768-32x32xk4",
               34: "lean-stream-k1024-32x32xk8", 35: "lean-stream-k4096-32x32xk8r4", 36: "lean-stream-k768-48x32xk6",
               37: "lean-stream-k3072-48x32xk8", 38: "lean-stream-k1792-64x32xk7",
               39: "lean-stream-k1024-48x32xk8", 40: "lean-conv-16x16xk12", 41: "lean-conv-32x32xk8", 42: "lean-packed-splitA-k1792-64x32xk7", 43: "lean-score-k1024-32x32xk8", 44: "lean-softmax-pv-16x64xk8",
               45: "lean-prob-k1024-256x128", 46: "lean-prob-pv-256x128",
               # many-row lean instances (bm_kernel: LDS-staged operands, epilogue in registers)
               50: "lean-rope-k1024-256x128", 51: "lean-rope-k1024-128x128", 52: "lean-rope-k768-128x128", 53: "lean-packed-k1024-256x128",
               54: "lean-packed-k1024-128x128", 55: "lean-packed-k768-128x128", 56: "lean-stream-k1024-128x64", 57: "lean-stream-k4096-128x64",
               58: "lean-stream-k768-128x64", 59: "lean-stream-k3072-128x64", 60: "lean-stream-k1792-128x64",
               61: "lean-stream-k1024-256x128", 62: "lean-stream-k4096-256x128", 63: "lean-rope-k768-256x128", 64: "lean-packed-k768-256x128", 65: "lean-rope-k768-128x64", 66: "lean-packed-splitA-k1792-128x128", 67: "lean-packed-splitA-k1792-256x128"}


def pick_tile(M, N, batch=1, splitk=1, conv=False, K=0, packed_bf16=False, plain=False, ln_nt=0, pipe_ok=False, rope=False):
    """Mirror of the auto heuristic in csrc/gemm.hip (kept in sync so profiles can be keyed by instantiation)."""
    t64 = ((M + 63) // 64) * ((N + 63) // 64) * batch * splitk
    t128 = ((M + 63) // 64) * ((N + 127) // 128) * batch
    lds_ok = not conv and splitk == 1 and packed_bf16 and K % 64 == 0
    wide_legacy = lds_ok and M >= 1024 and N >= 2304 and N % 128 == 0 and N < 16384
    if pipe_ok and PIPE_TILES and M >= 512 and not rope and not wide_legacy:
        mt256, mt128, nt128 = (M + 255) // 256, (M + 127) // 128, (N + 127) // 128
        if mt256 * nt128 * batch * splitk >= 144:
            return 20
        if mt128 * nt128 * batch * splitk >= 200:
            return 21
        if mt128 * ((N + 63) // 64) * batch * splitk <= 192:
            return 24
        return 22
    if lds_ok and M >= 1024 and N >= 2304 and N % 128 == 0:
        return 5 if (N % 4096 == 0 or ((M + 127) // 128) * (N // 128) * batch >= 1024) else 6
    if not conv and M >= 1024:
        return 1
    if lds_ok and 112 < M <= 224 and plain and N >= 3072 and K <= 1024 and ln_nt <= 32:
        return 13
    if conv or M > 2048:
        if t128 >= 1024 and N % 128 == 0:
            return 2
        if t64 >= 512:
            return 1
        return 0
    return 0


import os as _os
import threading as _threading
PIPE_TILES = True                                               # (the pipelined LDS tiles 20-23 of csrc/gemm.hip: always on since round 6)
PROF_SHAPES = _os.environ.get("SP3_PROF_SHAPES", "0")[:1] == "1"
LEAN = _os.environ.get("SP3_LEAN_GEMM", "1")[:1] != "0"          # mirrors sm_enabled() in csrc/gemm_sm.hip: with the lean instances off the
                                                                  # engine must not pick the layouts only they serve (packed split-A, bf16 DPT maps)

_pair = None


class pair:
    """`with ops.pair():` -- the two GEMM launches issued inside go out as ONE launch (sp3_gemm2: two differently shaped
    groups of problems that depend on the same inputs; same dtypes / loader; the second runs on the first one's tile)."""

    def __enter__(self):
        global _pair
        assert _pair is None
        _pair = []
        return self

    def __exit__(self, et, ev, tb):
        global _pair
        descs, _pair = _pair, None
        if et is not None:
            return False
        if len(descs) != 2:
            raise RuntimeError("ops.pair(): expected exactly two GEMM launches, got %d" % len(descs))
        (a, na), (b, _) = descs
        if a.tile < 0 and b.tile < 0:
            ta, tb = L.load().sp3_gemm_plan(C.byref(a)), L.load().sp3_gemm_plan(C.byref(b))
            if ta >= 30 and ta == tb and a.epi == L.EPI_ROPE_VT and b.epi == L.EPI_ROPE_VT:   # both groups on one lean q/k/v instance (the only
                a.tile = b.tile = ta                # family that takes a second group; others pair on the general tiles)
        if a.tile < 0:
            _pick_general(a)                        # (a lean instance for one group only: both take the general tiles)
        if b.tile < 0:
            b.tile = a.tile
        if _prof is None:
            L.check(L.load().sp3_gemm2(C.byref(a), C.byref(b), L.stream_ptr()), "sp3_gemm2")
            return False
        e0 = _prof.begin()
        L.check(L.load().sp3_gemm2(C.byref(a), C.byref(b), L.stream_ptr()), "sp3_gemm2")
        fl, by = 0.0, 0.0
        for d in (a, b):
            f, n = _gemm_cost(d)
            fl, by = fl + f, by + n
        _prof.end("gemm2<A%s,W%s,%s,%s>" % ("bf16" if a.a_bf16 else "f32", "f32" if a.wdtype == F32 else "bf16", na, _TILE_NAMES[a.tile]), e0, fl, by)
        return False


def _pipe_ok(d):
    return bool(d.loader == L.LOAD_PLAIN and d.a_packed and d.w_packed and d.a_bf16 and not d.A2 and not d.sm_stats_out and d.K % 64 == 0
                and (max(d.splitk, 1) == 1 or d.epi == L.EPI_PARTIAL))


def _pick_general(d):
    d.tile = pick_tile(d.M, d.N, max(d.batch, 1), max(d.splitk, 1), d.loader == L.LOAD_CONV3X3, d.K,
                       bool(d.a_packed and d.w_packed and d.a_bf16 and not d.A2 and d.epi != L.EPI_PARTIAL
                            and not d.sm_stats_out),
                       d.epi == L.EPI_PLAIN, d.ln_nt if d.ln_stats else 0, _pipe_ok(d), d.epi == L.EPI_ROPE_VT)


def _pick(d):
    t = L.load().sp3_gemm_plan(C.byref(d))          # the lean small-M instances (tiles 30..) are chosen by the library
    if t >= 30:
        d.tile = t
        return
    _pick_general(d)


def _gemm_cost(d):
    b = max(d.batch, 1)
    wsz = 4 if d.wdtype == F32 else 2
    asz, csz = (2 if d.a_bf16 else 4), (2 if d.out_bf16 else 4)
    a_elems = d.M * d.K if d.loader != L.LOAD_CONV3X3 else d.M * d.conv_stride * d.conv_stride * d.conv_C   # conv: the map, once
    return 2.0 * d.M * d.N * d.K * b, b * (asz * a_elems + wsz * d.N * d.K + csz * d.M * d.N)      # algorithmic: every operand once


def _gemm_launch(d, what, loader_name):
    if _pair is not None:
        _pair.append((d, loader_name))
        return
    if d.loader == L.LOAD_SOFTMAX:
        if d.tile < 0:
            t = L.load().sp3_gemm_plan(C.byref(d))          # 44: the lean P.V instance of the memory read
            d.tile = t if t >= 30 else 0
        elif d.tile < 30:
            d.tile = 1 if d.tile == 1 else 0
    if d.tile < 0:
        _pick(d)
    if _prof is None:
        L.check(L.load().sp3_gemm(C.byref(d), L.stream_ptr()), what)
        return
    e0 = _prof.begin()
    L.check(L.load().sp3_gemm(C.byref(d), L.stream_ptr()), what)
    b = max(d.batch, 1)
    wsz = 4 if d.wdtype == F32 else 2
    flops = 2.0 * d.M * d.N * d.K * b
    asz, csz = (2 if d.a_bf16 else 4), (2 if d.out_bf16 else 4)
    a_elems = d.M * d.K if d.loader != L.LOAD_CONV3X3 else d.M * d.conv_stride * d.conv_stride * d.conv_C   # conv: the map, once
    nbytes = b * (asz * a_elems + wsz * d.N * d.K + csz * d.M * d.N)      # algorithmic: every operand once
    adt = "bf16" if d.a_bf16 else "f32"
    tname = ("16x64xk4" if d.tile == 0 else "32x32xk4") if (d.loader == L.LOAD_SOFTMAX and d.tile < 30) else _TILE_NAMES[d.tile]
    key = "gemm<A%s,W%s,%s,%s>" % (adt, "f32" if d.wdtype == F32 else "bf16", loader_name, tname)
    if PROF_SHAPES:                                 # (tools: one profile line per GEMM shape)
        key += " %dx%dx%d x%d%s" % (d.M, d.N, d.K, b, " splitA" if d.A2 else "")
    _prof.end(key, e0, flops, nbytes)


def _timed(key, flops, nbytes, fn, *args):
    if _prof is None:
        return fn(*args)
    e0 = _prof.begin()
    r = fn(*args)
    _prof.end(key, e0, flops, nbytes)
    return r


def packed_shape(rows, K, dtype):
    """shape of a fragment-order operand buffer (include/spann3r_hip.h: a_packed / w_packed)"""
    KB, CH = (64, 16) if dtype == torch.bfloat16 else (32, 8)
    return ((rows + 15) // 16, (K + KB - 1) // KB, 2, 4, 16, CH // 2)


class PackedAct:
    """An activation matrix [M, K] in fragment order; produced by kernels with out_packed, consumed as GEMM A."""

    def __init__(self, M, K, dtype, device, data=None):
        self.M, self.K, self.dtype = M, K, dtype
        self.data = torch.zeros(packed_shape(M, K, dtype), dtype=dtype, device=device) if data is None else data
        self.is_cuda = self.data.is_cuda

    @staticmethod
    def from_dense(x):
        M, K = x.shape
        KB, CH = (64, 16) if x.dtype == torch.bfloat16 else (32, 8)
        nb, nkb = (M + 15) // 16, (K + KB - 1) // KB
        pad = torch.zeros(nb * 16, nkb * KB, dtype=x.dtype, device=x.device)
        pad[:M, :K] = x
        # [nb, r, kb, g, h, e] -> [nb, kb, h, g, r, e]
        return PackedAct(M, K, x.dtype, x.device, pad.view(nb, 16, nkb, 4, 2, CH // 2).permute(0, 2, 4, 3, 1, 5).contiguous())

    def to_dense(self):
        nb, nkb, _, _, _, ch2 = self.data.shape
        return self.data.permute(0, 4, 1, 3, 2, 5).reshape(nb * 16, nkb * 8 * ch2)[:self.M, :self.K]

    def data_ptr(self):
        return self.data.data_ptr()

    @staticmethod
    def group(G, M, K, dtype, device, alloc=None):
        """G fragment-order [M, K] matrices in one allocation, each starting on a 16-row boundary (grouped launches):
        .stride = elements between consecutive problems, .rows_pad = padded rows per problem.
        alloc(shape, dtype, zero=True) -> tensor: where the storage comes from (Engine._alloc: the workspace arena)."""
        KB = 64 if dtype == torch.bfloat16 else 32
        Mp, Kp = (M + 15) // 16 * 16, (K + KB - 1) // KB * KB
        t = PackedAct(G * Mp, K, dtype, device, data=None if alloc is None else alloc(packed_shape(G * Mp, K, dtype), dtype, zero=True))
        t.M, t.G, t.rows_pad, t.stride = M, G, Mp, Mp * Kp
        return t

    def at(self, g):
        """problem g of a group as a PackedAct of its own (for launches that start at another problem)"""
        v = PackedAct(self.M, self.K, self.dtype, self.data.device, data=self.data.view(-1)[g * self.stride:])
        v.stride = self.stride
        return v


class PackedWeight:
    """A weight matrix [N, K] re-ordered once into MFMA-fragment order (include/spann3r_hip.h: w_packed) so that
    every operand load of the GEMM is a fully coalesced, contiguous wave read."""

    def __init__(self, w2d, halves=False):
        """halves (fp32 weights of an f16x3 engine): every lane's 8 fp32 values of a k-block are stored as their two fp16 planes
        (h = fp16(w) in the block's first KB, l = fp16((w - h) * 2^11) in the second; same bytes per block) -- the split the f16x3
        product takes per use, taken once here; the GEMM is told by w_packed = 2 (include/spann3r_hip.h)."""
        N, K = w2d.shape
        self.N, self.K, self.dtype = N, K, w2d.dtype
        KB, CH = (64, 16) if w2d.dtype == torch.bfloat16 else (32, 8)
        nb, nkb = (N + 15) // 16, (K + KB - 1) // KB
        pad = torch.zeros(nb * 16, nkb * KB, dtype=w2d.dtype, device=w2d.device)
        pad[:N, :K] = w2d
        # [nb, r, kb, g, h, e] -> [nb, kb, h, g, r, e]
        self.data = pad.view(nb, 16, nkb, 4, 2, CH // 2).permute(0, 2, 4, 3, 1, 5).contiguous()
        self.halves = bool(halves) and w2d.dtype == torch.float32
        if self.halves:
            x8 = self.data.permute(0, 1, 3, 4, 2, 5).reshape(nb, nkb, 4, 16, 8)                   # [.., g, r, the lane's 8 k]
            hi = x8.to(torch.float16)
            lo = ((x8 - hi.float()) * 2048.0).to(torch.float16)
            self.data = torch.stack([hi, lo], dim=2).contiguous().view(torch.float32).view(nb, nkb, 2, 4, 16, 4)

    @classmethod
    def wrap(cls, data, N, K):
        """a fragment-order [N, K] operand that already lives in `data` (e.g. a spatial-memory bank written by sp3_bank_write)"""
        w = object.__new__(cls)
        w.N, w.K, w.dtype, w.data, w.halves = N, K, data.dtype, data, False
        return w

    def data_ptr(self):
        return self.data.data_ptr()

    def element_size(self):
        return self.data.element_size()


class PackedWeightGroup(PackedWeight):
    """Several equally shaped PackedWeights in one allocation (grouped launches); .stride = elements per problem"""

    def __init__(self, items):
        self.N, self.K, self.dtype = items[0].N, items[0].K, items[0].dtype
        self.halves = items[0].halves
        assert all(w.halves == self.halves for w in items)
        self.data = torch.stack([w.data for w in items]).contiguous()
        self.stride = items[0].data.numel()


WEIGHTS_EPOCH = 0    # bumped by every optimizer step of spann3r_amd.train (train.invalidate_weight_cache): part of Spann3R.engine's key

# Product mode of the GEMMs on fp32 operands (sp3_gemm_desc.f32x3).  It belongs to whoever launches: an Engine activates ITS mode
# at every entry point (Engine.activate), the training ops theirs -- and it is kept per THREAD, so two models of different
# precision in one process (one after the other, or one per thread) never see each other's setting.
#   0 "fp32": exact fp32 products (v_mfma_f32_16x16x4_f32)          1 "f32x3": three bf16 MFMAs of a (hi, lo) split
#   2 "f32_bf16": operands rounded to bf16 inside the GEMM (bf16 training step)
#   3 "f32x6": six bf16 MFMAs of a three-way split (fp32-grade)      4 "f16x3": three fp16 MFMAs of a (h, l * 2^-11) split
PRODUCT_MODES = {"fp32": 0, "bf16": 0, "f32x3": 1, "f32_bf16": 2, "f32x6": 3, "f16x3": 4}
_tls = _threading.local()
_default_mode = 0        # what a thread without its own setting sees: the training ops' mode (autograd runs backward on its own threads)


def set_product_mode(mode, process_default=False):
    """mode of the calling thread; process_default=True (spann3r_amd.train): also of every thread that never set one -- the
    autograd engine executes backward nodes on its device threads"""
    global _default_mode
    _tls.f32_mode = PRODUCT_MODES[mode] if isinstance(mode, str) else int(mode)
    if process_default:
        _default_mode = _tls.f32_mode


def get_product_mode():
    return getattr(_tls, "f32_mode", _default_mode)


class product_mode:
    """`with ops.product_mode("f32x3"): ...` -- the fp32-operand GEMMs launched inside (by this thread) use that product mode"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_product_mode()
        set_product_mode(self.mode)
        return self

    def __exit__(self, et, ev, tb):
        set_product_mode(self.prev)
        return False


def _w(d, W):
    """fills the W fields of a GemmDesc from a tensor or a PackedWeight"""
    d.W = W.data_ptr()
    d.w_packed = (2 if getattr(W, "halves", False) else 1) if isinstance(W, PackedWeight) else 0
    d.wdtype = wdtype_of(W)
    d.f32x3 = get_product_mode() if d.wdtype == F32 else 0
    if d.w_packed == 2 and d.f32x3 != PRODUCT_MODES["f16x3"]:
        raise ValueError("a PackedWeight(halves=True) holds fp16 planes: only the f16x3 product mode reads it (mode %d is active)" % d.f32x3)


class LnFold:
    """Consumer-side arguments of a folded LayerNorm: statistics partials of x [M, C/32, 2], s_n = sum_k (gamma*W)_nk."""

    def __init__(self, stats, C_, s, eps=1e-6, sb_stats=0, sb_s=0):
        self.stats, self.C, self.s, self.eps = stats, C_, s, eps
        self.sb_stats, self.sb_s = sb_stats, sb_s        # grouped launches: byte offsets per problem


def _ln(d, ln):
    if ln is not None:
        d.ln_stats, d.ln_s, d.ln_nt, d.ln_C, d.ln_eps = L.ptr(ln.stats), ln.s.data_ptr(), ln.C // 32, ln.C, ln.eps
        d.sb_ln_stats, d.sb_ln_s = ln.sb_stats, ln.sb_s


def _group(d, batch, strideA, strideW, strideC, sb):
    """grouped launch (include/spann3r_hip.h "grouped launches"): element strides of A / W / C and byte offsets of the rest"""
    d.batch, d.strideA, d.strideW, d.strideC = batch, strideA, strideW, strideC
    for k, v in (sb or {}).items():
        setattr(d, "sb_" + k, v)


def wdtype_of(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("weights must be float32 or bfloat16, got %s" % t.dtype)


def _f32(t, name):
    if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
        raise TypeError("%s must be a float32 CUDA(HIP) tensor" % name)


def _is_packed(t):
    return int(isinstance(t, PackedAct))


def _act(t, name):
    """GEMM A operands: fp32, or bf16 in bf16 mode (dense tensor or PackedAct).  Returns the a_bf16 flag."""
    if not t.is_cuda or t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("%s must be a float32/bfloat16 CUDA(HIP) tensor" % name)
    return int(t.dtype == torch.bfloat16)


def gemm(A, W, out, *, M, N, K, lda, ldc, bias=None, res1=None, ldr1=0, res2=None, ldr2=0, act=ACT_NONE, alpha=1.0,
         relu_in=False, tile=-1, batch=1, strideA=0, strideW=0, strideC=0, ldw=0,
         A2=None, lda2=0, K1=0, splitk=0, ln=None, stats_out=None, c2=None, sb=None, trace=None,
         sm_stats_out=None, softmax=None, plan_only=False, dyn_n=None):
    """out[M,N] = act(alpha * A[M,K] @ W[N,K]^T + bias) (+ res1 + res2).  nn.Linear / 1x1 conv / einsum.
    dyn_n (int32 device tensor): the memory read's growing extent as device state -- the score GEMM (sm_stats_out) reads its N, the
    softmax-loader GEMM its K from dyn_n[0]; the N / K passed here only bound the launch (include/spann3r_hip.h dyn_n).
    plan_only: nothing is launched; returns the tile the library would take for this descriptor (sp3_gemm_plan: >= 30 = a lean
    instance, < 30 = the general kernel's tiles, which not every operand layout has).
    `out` may be fp32 or bf16.  splitk >= 1 selects the PARTIAL epilogue: out is an fp32 [splitk, M, ldc] workspace
    that sp3_reduce_ln finishes.
    sm_stats_out: fp32 [M, ceil(N/32), 2] gets the (max, sum exp) partials of the finished rows (score GEMM of the memory
    read).  softmax=(stats, thresh, zout): A is an fp32 score matrix whose rows become thresholded probabilities on load
    (loader SOFTMAX); out = (P @ W^T) / kept mass (+ res), zout[M, 4] = (kept mass, row max, 1/Z, -)."""
    d = GemmDesc()
    d.a_bf16 = _act(A, "A")
    d.a_packed = _is_packed(A)
    d.out_packed = _is_packed(out)
    out_bf16 = out.dtype == torch.bfloat16
    _w(d, W)
    d.A, d.A2, d.C = A.data_ptr(), L.ptr(A2), out.data_ptr()
    d.bias, d.res1, d.res2 = L.ptr(bias), L.ptr(res1), L.ptr(res2)
    d.M, d.N, d.K, d.batch, d.K1 = M, N, K, batch, K1
    d.lda, d.lda2, d.ldw, d.ldc, d.ldr1, d.ldr2 = lda, lda2, ldw, ldc, ldr1, ldr2
    d.strideA, d.strideW, d.strideC = strideA, strideW, strideC
    d.alpha, d.act, d.out_bf16, d.relu_in = alpha, act, int(out_bf16), int(relu_in)
    d.loader, d.epi, d.tile = L.LOAD_PLAIN, L.EPI_PLAIN, tile
    if splitk >= 1:
        d.epi, d.splitk = L.EPI_PARTIAL, splitk
    _ln(d, ln)
    d.stats_out, d.c2 = L.ptr(stats_out), L.ptr(c2)
    d.trace = L.ptr(trace)
    d.sm_stats_out = L.ptr(sm_stats_out)
    d.dyn_n = L.ptr(dyn_n)
    if softmax is not None:
        st, thresh, zout = softmax
        d.loader, d.sm_stats, d.sm_nt, d.sm_thresh, d.sm_zout = L.LOAD_SOFTMAX, st.data_ptr(), (K + 31) // 32, float(thresh), L.ptr(zout)
    if sb:
        _group(d, batch, strideA, strideW, strideC, sb)
    if plan_only:
        return int(L.load().sp3_gemm_plan(C.byref(d)))
    _gemm_launch(d, "sp3_gemm", "softmax" if softmax is not None else "plain")
    return out


def reduce_ln(partial, splits, rows, C_, *, bias=None, res=None, ldres=0, x_out=None, ldx=0,
              ln1=None, out1=None, ld1=0, ln2=None, out2=None, ld2=0, eps=1e-6, act=ACT_NONE, res2=None, ldres2=0):
    """Finish a split-K GEMM: x = sum(partials) + bias (+ res); x_out = x; out1/out2 = LayerNorm(x) with (gamma, beta)
    pairs ln1 / ln2.  One launch replaces `x = x + proj(...)` and the LayerNorm(s) that follow on the stream."""
    d = L.ReduceLnDesc()
    d.partial, d.split_stride, d.splits, d.rows, d.C = partial.data_ptr(), rows * C_, splits, rows, C_
    d.bias, d.res, d.ldres = L.ptr(bias), L.ptr(res), ldres or C_
    d.x_out, d.ldx = L.ptr(x_out), ldx or C_
    d.act, d.res2, d.ldres2 = act, L.ptr(res2), ldres2 or C_
    if out1 is not None:
        d.g1, d.b1, d.out1, d.ld1, d.out1_bf16 = ln1[0].data_ptr(), ln1[1].data_ptr(), out1.data_ptr(), ld1 or C_, int(out1.dtype == torch.bfloat16)
        d.out1_packed = _is_packed(out1)
    if out2 is not None:
        d.g2, d.b2, d.out2, d.ld2, d.out2_bf16 = ln2[0].data_ptr(), ln2[1].data_ptr(), out2.data_ptr(), ld2 or C_, int(out2.dtype == torch.bfloat16)
        d.out2_packed = _is_packed(out2)
    d.eps = eps
    nln = (out1 is not None) + (out2 is not None)
    _timed("reduce_ln", (splits + 8.0 * nln) * rows * C_, 4.0 * rows * C_ * (splits + 2 + nln),
           lambda: L.check(L.load().sp3_reduce_ln(C.byref(d), L.stream_ptr()), "sp3_reduce_ln"))


def conv3x3(x, Wp, out, *, B, H, W_, Cin, Cout, stride=1, bias=None, res1=None, res2=None, act=ACT_NONE,
            relu_in=False, tile=-1, force_tile_kernel=False, splitk_ws=None, tile_px=None):
    """3x3 Conv2d, padding 1, on an NHWC fp32 map [B,H,W,Cin] -> [B,OH,OW,Cout] (implicit GEMM).
    Wp is the weight packed as [Cout, 9*Cin] with k = (ky*3+kx)*Cin + ci.
    tile_px (LDS-tiled kernel only; tests and tools/bench_conv_tile.py): None = the library's size rule, "8x8" | "8x16" | "8x16n32" = that
    workgroup tile of sp3_conv3x3_tile (pixels, n32 = 32 instead of 64 output channels)."""
    OH, OW = (H - 1) // stride + 1, (W_ - 1) // stride + 1
    if (tile < 0 and stride == 1 and isinstance(Wp, PackedWeight) and Wp.dtype == torch.bfloat16
            and Cin % 64 == 0 and Cout % 64 == 0 and Cin <= 768 and (force_tile_kernel or B * H * W_ >= 2048)):
        # bf16 weights, stride 1: the LDS-tiled kernel (input halo tile staged once, 9 taps read it from LDS)
        _act(x, "x")
        rbf = [t.dtype == torch.bfloat16 for t in (res1, res2) if t is not None]
        if rbf and (any(rbf) != all(rbf) or (any(rbf) and out.dtype != torch.bfloat16)):
            raise TypeError("conv3x3: residual maps must share one dtype (bf16 only next to a bf16 output)")
        oflag = int(out.dtype == torch.bfloat16) | (2 if any(rbf) else 0) | ({None: 0, "8x8": 1, "8x16": 2, "8x16n32": 3}[tile_px] << 2)
        _timed("conv3x3_tile", 2.0 * B * H * W_ * Cout * 9 * Cin,
               B * H * W_ * (x.element_size() * Cin + out.element_size() * Cout) + 2.0 * 9 * Cin * Cout,
               lambda: L.check(L.load().sp3_conv3x3_tile(
                   x.data_ptr(), int(x.dtype == torch.bfloat16), Wp.data_ptr(), L.ptr(bias), L.ptr(res1), L.ptr(res2),
                   out.data_ptr(), oflag, B, H, W_, Cin, Cout, int(relu_in), act,
                   L.stream_ptr()), "sp3_conv3x3_tile"))
        return out
    d = GemmDesc()
    d.a_bf16 = _act(x, "x")
    d.out_bf16 = int(out.dtype == torch.bfloat16)
    _w(d, Wp)
    M = B * OH * OW
    lean = -1
    if tile < 0 and isinstance(Wp, PackedWeight):
        # small maps (<= 1024 pixels, bf16 weights): one lean launch, K split inside the workgroup -- no split-K workspace, no reduce launch
        p = GemmDesc()
        p.a_bf16, p.out_bf16 = d.a_bf16, d.out_bf16
        _w(p, Wp)
        p.A, p.C, p.bias, p.res1, p.res2 = x.data_ptr(), out.data_ptr(), L.ptr(bias), L.ptr(res1), L.ptr(res2)
        p.M, p.N, p.K, p.batch = M, Cout, 9 * Cin, 1
        p.lda, p.ldc, p.ldr1, p.ldr2 = Cin, Cout, Cout, Cout
        p.alpha, p.act, p.relu_in = 1.0, act, int(relu_in)
        p.loader, p.epi, p.tile = L.LOAD_CONV3X3, L.EPI_PLAIN, -1
        p.conv_H, p.conv_W, p.conv_C, p.conv_OH, p.conv_OW, p.conv_stride = H, W_, Cin, OH, OW, stride
        rbf = [t.dtype == torch.bfloat16 for t in (res1, res2) if t is not None]
        if rbf and any(rbf) != all(rbf):
            raise TypeError("conv3x3: residual maps must share one dtype")
        p.res_bf16 = int(any(rbf))                  # the lean instances read the residuals in the map dtype: the plan refuses a mismatch
        lean = L.load().sp3_gemm_plan(C.byref(p))
        if lean >= 30:
            p.tile = lean
            _gemm_launch(p, "sp3_gemm(conv3x3)", "conv3x3")
            return out
    if any(t is not None and t.dtype == torch.bfloat16 for t in (res1, res2)):
        raise TypeError("conv3x3: bf16 residual maps are served by the lean small-map and the LDS-tiled kernels only "
                        "(B*H*W=%d, Cin=%d, Cout=%d, stride %d)" % (B * H * W_, Cin, Cout, stride))
    S = conv_splitk(M, Cout, 9 * Cin, Wp.dtype) if (splitk_ws is not None and tile < 0 and out.dtype == torch.float32) else 1
    if S > 1 and splitk_ws.numel() < S * M * Cout:
        raise ValueError("conv3x3: splitk_ws holds %d floats, needs %d" % (splitk_ws.numel(), S * M * Cout))
    d.A, d.C = x.data_ptr(), (splitk_ws if S > 1 else out).data_ptr()
    if S == 1:
        d.bias, d.res1, d.res2 = L.ptr(bias), L.ptr(res1), L.ptr(res2)
    d.M, d.N, d.K, d.batch = M, Cout, 9 * Cin, 1
    d.lda, d.ldc, d.ldr1, d.ldr2 = Cin, Cout, Cout, Cout
    d.alpha, d.act, d.relu_in = 1.0, (act if S == 1 else ACT_NONE), int(relu_in)
    d.loader, d.epi, d.tile = L.LOAD_CONV3X3, (L.EPI_PLAIN if S == 1 else L.EPI_PARTIAL), (tile if S == 1 else 0)
    d.splitk = S if S > 1 else 0
    d.conv_H, d.conv_W, d.conv_C, d.conv_OH, d.conv_OW, d.conv_stride = H, W_, Cin, OH, OW, stride
    _gemm_launch(d, "sp3_gemm(conv3x3)", "conv3x3")
    if S > 1:
        # few output tiles, long K (small maps): K split over S workgroups per tile, finished by one reduce launch
        reduce_ln(splitk_ws, S, M, Cout, bias=bias, res=res1, ldres=Cout, x_out=out, ldx=Cout, act=act, res2=res2, ldres2=Cout)
    return out


def conv_splitk(M, N, K, wdtype):
    """K split of the implicit-GEMM convolution: only when the 32x32 tiles alone leave most CUs idle"""
    tiles = ((M + 31) // 32) * ((N + 31) // 32)
    nkb = K // (64 if wdtype == torch.bfloat16 else 32)
    s = 1
    while tiles * s * 2 <= 512 and nkb // (s * 2) >= 4 and s < 8:
        s *= 2
    return s


def conv_transpose_ks(x, Wp, out, *, B, H, W_, Cin, Cout, ks, bias=None, tile=-1):
    """ConvTranspose2d with kernel == stride == ks on NHWC [B,H,W,Cin] -> [B,ks*H,ks*W,Cout].
    Wp is packed [(ky*ks+kx)*Cout + co, ci]."""
    d = GemmDesc()
    d.a_bf16 = _act(x, "x")
    d.out_bf16 = int(out.dtype == torch.bfloat16)
    _w(d, Wp)
    d.A, d.C, d.bias = x.data_ptr(), out.data_ptr(), L.ptr(bias)
    d.M, d.N, d.K, d.batch = B * H * W_, ks * ks * Cout, Cin, 1
    d.lda, d.ldc = Cin, Cout
    d.alpha = 1.0
    d.loader, d.epi, d.tile = L.LOAD_PLAIN, L.EPI_PIXSHUF, tile
    d.ps_k, d.ps_H, d.ps_W, d.ps_C = ks, H, W_, Cout
    _gemm_launch(d, "sp3_gemm(conv_transpose)", "plain")
    return out


def proj_rope_vt(A, W, bias, out_qk, ldc, vt, vt_ld, *, M, N, K, lda, rope_cols, pos, cos, sin, tokens, heads, tile=-1,
                 qkv_packed=False, ln=None, batch=1, strideA=0, strideW=0, strideC=0, sb=None):
    """Fused q/k(/v) projection of an attention layer: bias + 2-D RoPE on columns [0, rope_cols)
    (stored row-major to out_qk) and per-head transposed store of the V columns to vt."""
    d = GemmDesc()
    d.a_bf16 = _act(A, "A")
    d.a_packed = _is_packed(A)
    _w(d, W)
    d.A, d.C, d.bias = A.data_ptr(), L.ptr(out_qk), L.ptr(bias)
    d.M, d.N, d.K, d.batch = M, N, K, 1
    d.lda, d.ldc = lda, ldc
    d.alpha = 1.0
    d.loader, d.epi, d.tile = L.LOAD_PLAIN, L.EPI_ROPE_VT, tile
    d.rope_cos, d.rope_sin, d.pos, d.rope_cols = cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), rope_cols
    d.vt, d.vt_ld, d.tokens, d.heads = L.ptr(vt), vt_ld, tokens, heads
    d.qkv_packed = int(qkv_packed)
    _ln(d, ln)
    if out_qk is None:
        d.C = vt.data_ptr()           # unused by the kernel when rope_cols == 0, but must be non-null
    if batch > 1:
        _group(d, batch, strideA, strideW, strideC, sb)
    _gemm_launch(d, "sp3_gemm(rope_vt)", "plain")


def layernorm(x, gamma, beta, eps, out, *, rows, C_, ldx=None, ldo=None, transposed=False, dual=None, group_rows=0):
    """dual: a bf16 PackedAct(.group) that receives a fragment-order copy in the same launch, rows taken in groups of group_rows,
    each group starting at a multiple of dual.rows_pad packed rows (sp3_layernorm_dual)"""
    _f32(x, "x")
    ldx = C_ if ldx is None else ldx
    ldo = C_ if ldo is None else ldo
    if dual is not None:
        assert dual.dtype == torch.bfloat16 and not transposed and not _is_packed(out)
        pad = getattr(dual, "rows_pad", (group_rows + 15) // 16 * 16)
        _timed("layernorm", 8.0 * rows * C_, rows * C_ * 10.0,
               lambda: L.check(L.load().sp3_layernorm_dual(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), eps, out.data_ptr(), ldo,
                                                           dual.data_ptr(), rows, C_, group_rows or rows, pad, L.stream_ptr()),
                               "sp3_layernorm_dual"))
        return out
    if _is_packed(out):
        _timed("layernorm", 8.0 * rows * C_, rows * C_ * 6.0,
               lambda: L.check(L.load().sp3_layernorm_packed(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), eps,
                                                             out.data_ptr(), int(out.dtype == torch.bfloat16), rows, C_,
                                                             L.stream_ptr()), "sp3_layernorm_packed"))
        return out
    fn = L.load().sp3_layernorm_t if transposed else L.load().sp3_layernorm
    _timed("layernorm_t" if transposed else "layernorm", 8.0 * rows * C_, rows * C_ * (4.0 + out.element_size()),
           lambda: L.check(fn(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), eps, out.data_ptr(), ldo,
                              int(out.dtype == torch.bfloat16), rows, C_, L.stream_ptr()), "sp3_layernorm"))
    return out


def rope_2d(tokens, positions, base, fwd):
    """In-place 2-D RoPE on a [B,N,H,D] view (curope.rope_2d drop-in; curope.cpp:49-69)."""
    if tokens.dim() != 4:
        raise RuntimeError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise RuntimeError("positions must have 3 dimensions")
    if tokens.size(0) != positions.size(0):
        raise RuntimeError("batch size differs between tokens & positions")
    if tokens.size(1) != positions.size(1):
        raise RuntimeError("seq_length differs between tokens & positions")
    if positions.size(2) != 2:
        raise RuntimeError("positions.shape[2] must be equal to 2")
    if tokens.is_cuda != positions.is_cuda:
        raise RuntimeError("tokens and positions are not on the same device")
    if not tokens.is_cuda:
        raise RuntimeError("spann3r_amd rope_2d is the MI355X kernel: tokens must be on the GPU (no CPU fallback)")
    if tokens.stride(3) != 1:
        raise RuntimeError("tokens are not contiguous")
    if not positions.is_contiguous() or positions.dtype != torch.int64:
        raise RuntimeError("positions are not contiguous int64")
    B, N, H, D = tokens.shape
    dt = wdtype_of(tokens)
    L.check(L.load().sp3_rope_2d(tokens.data_ptr(), dt, B, N, H, D, tokens.stride(0), tokens.stride(1), tokens.stride(2),
                                 positions.data_ptr(), float(base), float(fwd), L.stream_ptr()), "sp3_rope_2d")
    return tokens


def attention(q, sq, ldq, k, sk, ldk, vt, vt_ld, out, ldo, *, B, heads, Nq, Nk, scale):
    es = vt.element_size()
    _timed("attention<%s>" % ("f32" if es == 4 else "bf16"), 4.0 * B * heads * Nq * Nk * 64,
           B * heads * 64.0 * (es * (Nq + 2 * Nk) + 4 * Nq),
           lambda: L.check(L.load().sp3_attention_ex(q.data_ptr(), sq, ldq, k.data_ptr(), sk, ldk, vt.data_ptr(), vt_ld,
                                                     out.data_ptr(), ldo, int(out.dtype == torch.bfloat16), _is_packed(out),
                                                     B, heads, Nq, Nk, float(scale), (2 if (get_product_mode() == 1 and es == 4) else 3 if (get_product_mode() == 4 and es == 4) else wdtype_of(vt)),
                                                     L.stream_ptr()),
                           "sp3_attention"))
    return out


def attention_packed(qp, q_cols, q_col0, npad_q, kp, k_cols, k_col0, npad_k, vtp, out, ldo, *, B, heads, Nq, Nk, scale,
                     o_group=0, o_group_rows=0):
    """bf16 attention on the fragment-order q/k and PV-order V written by proj_rope_vt(qkv_packed=True)."""
    _timed("attention_packed<bf16>", 4.0 * B * heads * Nq * Nk * 64, B * heads * 64.0 * (2 * (Nq + 2 * Nk) + 4 * Nq),
           lambda: L.check(L.load().sp3_attention_packed(qp.data_ptr(), q_cols, q_col0, npad_q, kp.data_ptr(), k_cols, k_col0,
                                                         npad_k, vtp.data_ptr(), out.data_ptr(), ldo,
                                                         int(out.dtype == torch.bfloat16), _is_packed(out), B, heads, Nq, Nk,
                                                         float(scale), o_group, o_group_rows, L.stream_ptr()),
                           "sp3_attention_packed"))
    return out


def attention_packed_qproj(xq, stats, Wq, ln_s, bias, pos, cos, sin, kp, k_cols, k_col0, npad_k, vtp, out, ldo, *, B, heads, Nq, Nk, scale,
                           o_group, o_group_rows, eps=1e-6, stats_group_stride=0, vec_group_stride=0):
    """Cross-attention with its query projection inside the launch (include/spann3r_hip.h sp3_attention_packed_qproj):
    q = RoPE2D(LN(x) Wq^T + b) per (16 query rows, head) workgroup from xq (PackedAct.group: fragment-order bf16 x of both decoder
    sides), the producer's LayerNorm partials `stats`, the fragment-order weight group Wq with its folded column sums / bias."""
    d = L.AttnQProjDesc()
    D = heads * 64
    d.x_packed, d.x_group_stride = xq.data_ptr(), getattr(xq, "stride", 0)
    d.ln_stats, d.stats_group_stride = stats.data_ptr(), stats_group_stride
    d.w_packed, d.w_group_stride = Wq.data_ptr(), getattr(Wq, "stride", 0)
    d.ln_s, d.bias, d.vec_group_stride = ln_s.data_ptr(), bias.data_ptr(), vec_group_stride
    d.pos, d.rope_cos, d.rope_sin, d.ln_eps, d.D = pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), eps, D
    d.kp, d.k_cols, d.k_col0, d.npad_k, d.vtp = kp.data_ptr(), k_cols, k_col0, npad_k, vtp.data_ptr()
    d.out, d.ldo, d.out_bf16, d.out_packed = out.data_ptr(), ldo, int(out.dtype == torch.bfloat16), _is_packed(out)
    d.B, d.heads, d.Nq, d.Nk, d.scale, d.o_group, d.o_group_rows = B, heads, Nq, Nk, float(scale), o_group, o_group_rows
    # algorithmic work: the attention's + the projection GEMM's (2 Nq D D per image); bytes: q is never materialised
    _timed("attention_packed_qproj<bf16>", 4.0 * B * heads * Nq * Nk * 64 + 2.0 * B * Nq * D * D,
           B * heads * 64.0 * (2 * (2 * Nk) + 4 * Nq) + 2.0 * B * Nq * D + 2.0 * (B // max(o_group, 1)) * D * D,
           lambda: L.check(L.load().sp3_attention_packed_qproj(C.byref(d), L.stream_ptr()), "sp3_attention_packed_qproj"))
    return out


def softmax_thresh(S, P, *, ld, rows, M, Mpad, thresh, batch=1, strideS=0, packed=None, stride_packed=0):
    """P: optional fp32 row-major output; packed: optional bf16 / fp32 buffer that receives the probabilities in fragment
    order [rows, M rounded up to a k-block] (the P.V GEMM's packed A)"""
    pbf = packed is not None and packed.dtype == torch.bfloat16
    _timed("softmax_thresh", 8.0 * batch * rows * M, (4.0 + (0.0 if P is None else 4.0) + (0.0 if packed is None else packed.element_size())) * batch * rows * M,
           lambda: L.check(L.load().sp3_softmax_thresh(S.data_ptr(), L.ptr(P), ld, strideS, rows, M, Mpad, float(thresh),
                                                       batch, L.ptr(packed), stride_packed, int(pbf), L.stream_ptr()), "sp3_softmax_thresh"))


def softmax_pack(S, packed, rowstat, *, ld, rows, M, thresh, batch=1, strideS=0, stride_packed=0):
    """long banks: rows of S -> thresholded, renormalised probabilities as the P.V GEMM's fragment-order operand (two streaming
    launches, sp3_softmax_pack); rowstat: fp32 [batch * rows * 4] workspace"""
    _timed("softmax_pack", 8.0 * batch * rows * M, (4.0 + packed.element_size()) * batch * rows * M,
           lambda: L.check(L.load().sp3_softmax_pack(S.data_ptr(), ld, strideS, rows, M, float(thresh), batch, packed.data_ptr(), stride_packed,
                                                     int(packed.dtype == torch.bfloat16), rowstat.data_ptr(), L.stream_ptr()), "sp3_softmax_pack"))


def colsum_accum(P, ld, rows, M, mem_attn):
    L.check(L.load().sp3_colsum_accum(P.data_ptr(), ld, rows, M, mem_attn.data_ptr(), L.stream_ptr()), "sp3_colsum_accum")


def colsum_packed(packed, rows, M, mem_attn):
    _timed("colsum_packed", 1.0 * rows * M, 1.0 * packed.element_size() * rows * M,
           lambda: L.check(L.load().sp3_colsum_packed(packed.data_ptr(), int(packed.dtype == torch.bfloat16), rows, M, mem_attn.data_ptr(),
                                                      L.stream_ptr()), "sp3_colsum_packed"))


def copy_multi(pairs):
    """[(src, dst), ...] (up to 8, contiguous device tensors of equal byte size per pair) copied in one launch"""
    n = len(pairs)
    if not 1 <= n <= 8:
        raise ValueError("copy_multi: 1..8 copies")
    S, D, Bt = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int64 * n)()
    for i, (s_, d_) in enumerate(pairs):
        nb = s_.numel() * s_.element_size()
        if not (s_.is_contiguous() and d_.is_contiguous()) or nb != d_.numel() * d_.element_size():
            raise ValueError("copy_multi: pair %d must be contiguous and of equal size" % i)
        S[i], D[i], Bt[i] = s_.data_ptr(), d_.data_ptr(), nb
    _timed("copy_multi", 0.0, 2.0 * sum(Bt), lambda: L.check(L.load().sp3_copy_multi(n, S, D, Bt, L.stream_ptr()), "sp3_copy_multi"))


def colsum_softmax(S, ld, rows, M, rowz, thresh, mem_attn, mem_count=None, append_P=0):
    """mem_attn[j] += column sums of the thresholded, renormalised softmax of S[:rows, :M]; rowz [rows, 4] = the softmax-loader
    GEMM's zout (kept mass, max, 1/Z, -) (two-launch memory read).  append_P > 0: the same launch also does
    mem_append(mem_count, mem_attn, M, append_P)."""
    _timed("colsum_softmax", 4.0 * rows * M, 4.0 * rows * M,
           lambda: L.check(L.load().sp3_colsum_softmax(S.data_ptr(), ld, rows, M, rowz.data_ptr(), float(thresh), mem_attn.data_ptr(),
                                                       L.ptr(mem_count), append_P, L.stream_ptr()), "sp3_colsum_softmax"))


def bank_write(feat_k, feat_v, bank, M, P, C_, cap, norms, alpha, eps=1e-5, state=None):
    """bank: dict with k_raw, v_raw, k_hat, v_hat_t, s_bank, b_bank (one batch element); norms: (gk, bk, gv, bv, gq, bq).
    state (int32 device tensor, bank_state_set): the first row is read from state[0] on the device instead of M."""
    d = L.BankWriteDesc()
    d.state = L.ptr(state)
    d.feat_k, d.feat_v = feat_k.data_ptr(), feat_v.data_ptr()
    d.k_raw, d.v_raw, d.k_hat, d.v_hat_t = bank["k_raw"].data_ptr(), bank["v_raw"].data_ptr(), bank["k_hat"].data_ptr(), bank["v_hat_t"].data_ptr()
    d.s_bank, d.b_bank = bank["s_bank"].data_ptr(), bank["b_bank"].data_ptr()
    d.gamma_k, d.beta_k, d.gamma_v, d.beta_v, d.gamma_q, d.beta_q = [t.data_ptr() for t in norms]
    d.eps, d.alpha, d.M, d.P, d.C, d.cap, d.wdtype = eps, alpha, M, P, C_, cap, wdtype_of(bank["k_hat"])
    es = bank["k_hat"].element_size()
    _timed("bank_write", 20.0 * P * C_, P * C_ * (16.0 + 2 * es),
           lambda: L.check(L.load().sp3_bank_write(C.byref(d), L.stream_ptr()), "sp3_bank_write"))


def pack_stats(x, packed, stats, *, rows, C_, ldx=None):
    _f32(x, "x")
    L.check(L.load().sp3_pack_stats(x.data_ptr(), C_ if ldx is None else ldx, rows, C_, packed.data_ptr(),
                                    int(packed.dtype == torch.bfloat16), stats.data_ptr(), L.stream_ptr()), "sp3_pack_stats")


_pack_ws = {}


def _pack_colsum_ws(device, nby, nbx):
    """partial column sums of sp3_pack_bf16_colsum, grown on demand; one launch at a time uses them (the training step is one stream)"""
    ws = _pack_ws.get(str(device))
    need = nby * nbx * 64
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 18), device=device)
        _pack_ws[str(device)] = ws
    return ws


def pack_bf16(x2d, want=True, want_t=False, colsum=None, accumulate=False, act=0):
    """fp32 row-major [rows, cols] (any row stride) -> (PackedAct [rows, cols] or None, PackedAct [cols, rows] or None) in bf16
    fragment order, one launch (sp3_pack_bf16); the pads of both are zeros.  colsum (fp32 [cols], rows <= 8192): x2d.sum(0) is written /
    added there by the same launch.  act (1 GELU, 2 ReLU): applied to every element as it is loaded (sp3_pack_bf16_act)."""
    rows, cols = x2d.shape
    _f32(x2d, "x")
    if x2d.stride(1) != 1:
        x2d = x2d.contiguous()
    alloc = lambda r, c: torch.empty(packed_shape(r, c, torch.bfloat16), dtype=torch.bfloat16, device=x2d.device)
    a = PackedAct(rows, cols, torch.bfloat16, x2d.device, data=alloc(rows, cols)) if want else None
    t = PackedAct(cols, rows, torch.bfloat16, x2d.device, data=alloc(cols, rows)) if want_t else None
    if act:
        assert colsum is None
        _timed("pack_bf16", 0.0, rows * cols * (4.0 + 2.0 * (bool(want) + bool(want_t))),
               lambda: L.check(L.load().sp3_pack_bf16_act(x2d.data_ptr(), x2d.stride(0), rows, cols, L.ptr(a), L.ptr(t), int(act), L.stream_ptr()), "sp3_pack_bf16_act"))
        return a, t
    if colsum is not None:
        pw = _pack_colsum_ws(x2d.device, (rows + 63) // 64, (cols + 63) // 64)
        _timed("pack_bf16", 0.0, rows * cols * (4.0 + 2.0 * (bool(want) + bool(want_t))),
               lambda: L.check(L.load().sp3_pack_bf16_colsum(x2d.data_ptr(), x2d.stride(0), rows, cols, L.ptr(a), L.ptr(t), colsum.data_ptr(), int(accumulate),
                                                             pw.data_ptr(), L.stream_ptr()), "sp3_pack_bf16_colsum"))
        return a, t
    _timed("pack_bf16", 0.0, rows * cols * (4.0 + 2.0 * (bool(want) + bool(want_t))),
           lambda: L.check(L.load().sp3_pack_bf16(x2d.data_ptr(), x2d.stride(0), rows, cols, L.ptr(a), L.ptr(t), L.stream_ptr()), "sp3_pack_bf16"))
    return a, t


def pack_bf16_conv3x3(x, stride, want=True, want_t=False, act=0):
    """NHWC fp32 map [B, H, W, C] -> the bf16 fragment-order copies of its 3x3 / pad 1 im2col matrix [B*OH*OW, 9C] (and / or its transpose),
    gathered inside the pack launch (sp3_pack_bf16_conv3x3): the column matrix itself is never materialised"""
    B, H, W_, C_ = x.shape
    _f32(x, "x")
    assert x.is_contiguous()
    OH, OW = (H - 1) // stride + 1, (W_ - 1) // stride + 1
    rows, cols = B * OH * OW, 9 * C_
    alloc = lambda r, c: torch.empty(packed_shape(r, c, torch.bfloat16), dtype=torch.bfloat16, device=x.device)
    a = PackedAct(rows, cols, torch.bfloat16, x.device, data=alloc(rows, cols)) if want else None
    t = PackedAct(cols, rows, torch.bfloat16, x.device, data=alloc(cols, rows)) if want_t else None
    _timed("pack_bf16_conv3x3", 0.0, rows * cols * 2.0 * (bool(want) + bool(want_t)) + x.numel() * 4.0,
           lambda: L.check(L.load().sp3_pack_bf16_conv3x3(x.data_ptr(), B, H, W_, C_, stride, int(act), L.ptr(a), L.ptr(t), L.stream_ptr()), "sp3_pack_bf16_conv3x3"))
    return a, t


def gather_packed_rows(src, dst, sel, n_sel, C_):
    L.check(L.load().sp3_gather_packed_rows(src.data_ptr(), dst.data_ptr(), sel.data_ptr(), n_sel, C_, src.element_size(),
                                            L.stream_ptr()), "sp3_gather_packed_rows")


def gather_packed_cols(src, dst, sel, n_sel, n_fill, C_, cap):
    L.check(L.load().sp3_gather_packed_cols(src.data_ptr(), dst.data_ptr(), sel.data_ptr(), n_sel, n_fill, C_, cap, src.element_size(),
                                            L.stream_ptr()), "sp3_gather_packed_cols")


def prob_merge(stats, scale, rows, M, cap, dyn_n=None):
    """group statistics of the long-bank read's score stage -> scale[group][row] = exp(m_g - m_row) / Z_row (include/spann3r_hip.h)"""
    L.check(L.load().sp3_prob_merge(stats.data_ptr(), scale.data_ptr(), rows, M, cap, L.ptr(dyn_n), L.stream_ptr()), "sp3_prob_merge")


def colsum_prob(p_packed, scale, rows, M, cap, mem_attn, dyn_n=None):
    """mem_attn[:M] += column sums of softmax = p~ * scale (long-bank read without a score matrix)"""
    _timed("colsum_prob", 2.0 * rows * M, 2.0 * rows * M,
           lambda: L.check(L.load().sp3_colsum_prob(p_packed.data_ptr(), scale.data_ptr(), rows, M, cap, L.ptr(dyn_n), mem_attn.data_ptr(), L.stream_ptr()),
                           "sp3_colsum_prob"))


def bank_state_set(state, M, wm):
    """state[0] = M, state[1] = wm on the device (the fill level the kernels of a step read: their hipGraph does not depend on it)"""
    L.check(L.load().sp3_bank_state_set(state.data_ptr(), int(M), int(wm), L.stream_ptr()), "sp3_bank_state_set")


def cos_sim_state(k, k_raw, Tmax, P, C_, state, score, scratch):
    """cos_sim against the last state[1] (<= Tmax) frames of the bank's k_raw, rows [state[0] - state[1] P, state[0])"""
    assert scratch.numel() >= Tmax * P and scratch.dtype == torch.float32 and state.dtype == torch.int32
    L.check(L.load().sp3_cos_sim_state(k.data_ptr(), k_raw.data_ptr(), Tmax, P, C_, state.data_ptr(), scratch.data_ptr(), score.data_ptr(),
                                       L.stream_ptr()), "sp3_cos_sim_state")


def cos_sim(k, wm, T, P, C_, score, scratch):
    """score[t] = mean_p cos(k[p], wm[t, p]); scratch: fp32 [>= T*P]"""
    assert scratch.numel() >= T * P and scratch.dtype == torch.float32
    L.check(L.load().sp3_cos_sim(k.data_ptr(), wm.data_ptr(), T, P, C_, scratch.data_ptr(), score.data_ptr(), L.stream_ptr()),
            "sp3_cos_sim")


def mem_append(count, attn, M, P):
    L.check(L.load().sp3_mem_append(count.data_ptr(), attn.data_ptr(), M, P, L.stream_ptr()), "sp3_mem_append")


def prune_select(attn, count, M, protect, top_k, sel):
    L.check(L.load().sp3_prune_select(attn.data_ptr(), count.data_ptr(), M, float(protect), top_k, sel.data_ptr(),
                                      L.stream_ptr()), "sp3_prune_select")


def gather_rows(src, dst, sel, n_sel, C_):
    L.check(L.load().sp3_gather_rows(src.data_ptr(), dst.data_ptr(), sel.data_ptr(), n_sel, C_, src.element_size(),
                                     L.stream_ptr()), "sp3_gather_rows")


def gather_cols(src, ld_src, dst, ld_dst, sel, n_sel, n_fill, C_):
    L.check(L.load().sp3_gather_cols(src.data_ptr(), ld_src, dst.data_ptr(), ld_dst, sel.data_ptr(), n_sel, n_fill, C_,
                                     src.element_size(), L.stream_ptr()), "sp3_gather_cols")


def gather_1d(src, dst, sel, n_sel):
    L.check(L.load().sp3_gather_1d(src.data_ptr(), dst.data_ptr(), sel.data_ptr(), n_sel, L.stream_ptr()), "sp3_gather_1d")


def im2col_patch(img, out, *, B, C_, H, W_, p, strides):
    _f32(img, "img")
    sb, sc, sy, sx = strides
    L.check(L.load().sp3_im2col_patch(img.data_ptr(), sb, sc, sy, sx, B, C_, H, W_, p, out.data_ptr(),
                                      int(out.dtype == torch.bfloat16), _is_packed(out), L.stream_ptr()), "sp3_im2col_patch")
    return out


def upsample2x(x, out, *, B, H, W_, C_, outH=None, outW=None):
    outH = 2 * H if outH is None else outH
    outW = 2 * W_ if outW is None else outW
    if x.dtype != out.dtype or x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("upsample2x: fp32 or bf16 maps, input and output alike")
    fn = L.load().sp3_upsample2x_bf16 if x.dtype == torch.bfloat16 else L.load().sp3_upsample2x
    _timed("upsample2x", 8.0 * B * outH * outW * C_, float(x.element_size()) * B * C_ * (H * W_ + outH * outW),
           lambda: L.check(fn(x.data_ptr(), out.data_ptr(), B, H, W_, C_, outH, outW, L.stream_ptr()), "sp3_upsample2x"))
    return out


def head_final(feat, w, b, pixels, C_, pts, conf, raw=None):
    fn = L.load().sp3_head_final_bf16 if feat.dtype == torch.bfloat16 else L.load().sp3_head_final
    _timed("head_final", 8.0 * pixels * C_, pixels * (float(feat.element_size()) * C_ + 16.0),
           lambda: L.check(fn(feat.data_ptr(), w.data_ptr(), b.data_ptr(), pixels, C_, pts.data_ptr(),
                              conf.data_ptr(), L.ptr(raw), L.stream_ptr()), "sp3_head_final"))


def fill(t, v):
    L.check(L.load().sp3_fill_f32(t.data_ptr(), float(v), t.numel(), L.stream_ptr()), "sp3_fill_f32")


def copy2d(src, lds, dst, ldd, rows, cols):
    L.check(L.load().sp3_copy2d_f32(src.data_ptr(), lds, dst.data_ptr(), ldd, rows, cols, L.stream_ptr()), "sp3_copy2d_f32")
