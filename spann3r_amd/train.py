"""Training-side pieces of the hot path: the spatial-memory read as a differentiable op whose forward
AND backward run in HIP kernels -- `SpatialMemory.memory_read` of the reference in train mode (spann3r/model.py:145-183
with attn_thresh = 0 and nn.Dropout on the attention, :475):

    out = dropout(softmax(LN_q(feat) . LN_k(mem_k)^T / sqrt(C))) . LN_v(mem_v) + feat

`memory_read_train` returns `out` with an autograd edge to feat, mem_k, mem_v and the six LayerNorm parameters; the
backward is four sp3_gemm launches with the transposes, the softmax / dropout backward and the LayerNorm backward of
csrc/train.hip in between.  The dropout mask is an input (0 or 1/(1-p), drawn by the caller: the reference's comes from
torch's RNG stream and cannot be matched bit for bit by any other implementation).

Below it: every other stage of the train-mode forward (ViT blocks, decoder blocks, key MLPs, DPT heads, value encoder) as
autograd ops with HIP forward and backward, `forward_train` (= Spann3R.forward in train mode), and the optimizer side of
spann3r/training.py:216-231: `FlatAdamW` (bucket-wide AdamW launches on the flat layout of `runner.GradReducer`, with the
global gradient-norm clip of croco/utils/misc.py:262-288 computed on the device).  Gradient averaging across ranks:
`spann3r_amd.runner.GradReducer`; the criterion: `spann3r_amd.loss`.

Attention (croco/models/blocks.py:94-112, :149-169) is `_FlashMHA` for heads of 64: one sp3_head_shuffle launch (RoPE from the
reference's own fp32 tables, per-head transposes), the flash forward sp3_attention_train_fwd, and in the backward
sp3_attention_train_bwd (dq | dk, dv with the probabilities recomputed) between two shuffles -- the attention matrix is never
written.  `_MHA` (other head widths, or FLASH_ATTENTION = False) keeps the shuffles around grouped GEMMs + softmax kernels;
`_Attention` on ATen-reshaped heads (FUSED_HEADS = False) is the round-2 path the tests compare both against.

Precision (`set_precision`): "fp32" = fp32 MFMA everywhere (the parity build: gradients within 1e-5 of float64 autograd,
8e-7 of the reference's own fp32 step at full depth); "bf16" = what the reference's bf16 autocast would compute: every Linear
runs on bf16 fragment-order copies of its operands (sp3_pack_bf16 makes X, X^T, dY, dY^T and the bias gradient's tile sums; W and
W^T are packed once per optimizer step) through the pipelined bf16 GEMM tiles with fp32 accumulation -- forward, dX = dY . W and
dW = dY^T . X are all plain A . W^T launches, no fp32 transposes -- the attention kernels round their fp32 operands into one bf16
MFMA per product, and the memory read's fp32 GEMMs do the same (sp3_gemm f32x3 = 2).  Master weights, gradients (flat buckets the
kernels accumulate into directly), LayerNorm, softmax statistics, the residual stream and the optimizer stay fp32."""
import ctypes
import math

import torch

from . import lib as L
from . import ops


PRECISION = "fp32"
FLASH_ATTENTION = True   # heads of 64: sp3_attention_train_fwd / _bwd (no attention matrix in memory); False: GEMMs + softmax kernels (_MHA)
ACT_ON_LOAD = True    # bf16: GELU / ReLU in front of a Linear or 3x3 convolution applied by its pack launch (False: separate activation launches)
CONV_GATHER = True    # bf16 3x3 convolutions: im2col gathered inside the pack launch (False: sp3_im2col3x3 + pack)
FUSED_HEADS = True    # attention through _MHA (one shuffle launch each way); False: the separate ATen reshapes + _Attention (tests compare the two)
_wcache = {}          # (id(weight), derivation) -> (stamp, packed W, packed W^T): one entry per weight, see _packed_weight


def set_precision(p):
    """"fp32" | "bf16" for every op of this module (forward and backward), see the module docstring"""
    global PRECISION
    if p not in ("fp32", "bf16"):
        raise ValueError("training precision must be 'fp32' or 'bf16'")
    PRECISION = p
    ops.set_product_mode("f32_bf16" if p == "bf16" else "fp32", process_default=True)


def invalidate_weight_cache():
    """Every optimizer of this module calls this after it has written the parameters (also through raw pointers or a replayed
    hipGraph, which leave the version counters alone): the bf16 weight copies are re-packed at their next use and every
    inference Engine built from the parameters is rebuilt (ops.WEIGHTS_EPOCH is part of Spann3R.engine's key)."""
    _wcache.clear()
    ops.WEIGHTS_EPOCH += 1


def _packed_weight(W, key):
    """(W, W^T) of a [N, K] fp32 matrix as bf16 fragment-order GEMM operands.  key = (id of the parameter, what was derived from
    it, *stamp) or None (no caching): ONE entry per (parameter, derivation), replaced when the stamp (version counter, address)
    has moved -- a torch optimizer bumps the version every step, so stale entries cannot pile up; optimizers that write through
    raw pointers call invalidate_weight_cache()."""
    if key is None:
        slot = stamp = None
        ent = None
    else:
        slot, stamp = key[:2], key[2:]
        ent = _wcache.get(slot)
        if ent is not None and ent[0] != stamp:
            ent = None
    if ent is None:
        a, t = ops.pack_bf16(W.detach(), True, True)
        N, K = W.shape
        ent = (stamp, ops.PackedWeight.wrap(a.data, N, K), ops.PackedWeight.wrap(t.data, K, N))
        if slot is not None:
            _wcache[slot] = ent
    return ent[1], ent[2]


def _colsum(dy, out=None):
    """db = dY.sum(0) of a contiguous fp32 [R, N] (two deterministic stages); `out` given: accumulated into it"""
    R, N = dy.shape
    acc = out is not None
    ws = torch.empty(int(L.load().sp3_colsum_rows_ws(R, N)) if R > 8192 else 4, device=dy.device)
    if out is None:
        out = torch.empty(N, device=dy.device)
    L.check(L.load().sp3_colsum_rows(dy.data_ptr(), dy.stride(0), R, N, out.data_ptr(), int(acc), ws.data_ptr(), L.stream_ptr()), "sp3_colsum_rows")
    return out


def _into_grad(p, compute):
    """Accumulate a parameter gradient straight into the flat-bucket view that IS p.grad (runner.GradReducer): `compute(out, acc)`
    writes (acc = False) or adds (acc = True) the gradient in place; the post-accumulate-grad hooks autograd would have run (the
    reducer's bucket bookkeeping) are run by hand.  Returns True when done (the caller then returns None to autograd)."""
    if not _direct_ok(p):
        return False
    compute(p.grad, True)
    _contributed(p)
    return True


def _contributed(p):
    p._sp3_pending = getattr(p, "_sp3_pending", 1) - 1      # a parameter used k times in the forward gets k contributions:
    if p._sp3_pending <= 0:                                 # the hooks run once, behind the last one (as AccumulateGrad does)
        for h in (p._post_accumulate_grad_hooks or {}).values():
            h(p)


def _direct_ok(p):
    return isinstance(p, torch.nn.Parameter) and p.grad is not None and p.grad.is_contiguous() and getattr(p, "_sp3_direct_grad", False)


def _expect(p):
    """forward side of _into_grad: one more contribution to p.grad is owed by a backward of this tape"""
    if isinstance(p, torch.nn.Parameter):
        p._sp3_pending = getattr(p, "_sp3_pending", 0) + 1


def reset_pending(params):
    """forget contributions owed by tapes that never ran their backward (a skipped step, a branch that missed the loss): called by
    GradReducer.prepare() / zero_grad() so that the hand-run post-accumulate hooks of the NEXT step fire again"""
    for p in params:
        p._sp3_pending = 0


def _r8(n):
    return (n + 7) // 8 * 8


def _r64(n):
    return (n + 63) // 64 * 64


def _transpose(src, rows, cols, ld_src, ld_dst=None):
    ld_dst = _r8(rows) if ld_dst is None else ld_dst
    if ld_dst - rows >= 8:
        dst = torch.zeros(cols, ld_dst, device=src.device)
        L.check(L.load().sp3_transpose(src.data_ptr(), ld_src, dst.data_ptr(), ld_dst, rows, cols, L.stream_ptr()), "sp3_transpose")
    else:                       # the kernel writes the (< 8) pad columns itself: no fill launch
        dst = torch.empty(cols, ld_dst, device=src.device)
        L.check(L.load().sp3_transpose_pad(src.data_ptr(), ld_src, 0, dst.data_ptr(), ld_dst, 0, rows, cols, 1, ld_dst, L.stream_ptr()), "sp3_transpose_pad")
    return dst


def _ln_bwd(x, gamma, dy, dx_add, eps, direct=None):
    """direct = (gamma parameter, beta parameter): their gradients are ADDED to the flat-bucket views (-> dx, None, None)"""
    rows, C = x.shape
    dx = torch.empty_like(x)
    acc = direct is not None and _direct_ok(direct[0]) and _direct_ok(direct[1])
    dg, db = (direct[0].grad, direct[1].grad) if acc else (torch.empty(C, device=x.device), torch.empty(C, device=x.device))
    scratch = torch.empty((rows + 3) // 4 * 2 * C, device=x.device)
    L.check(L.load().sp3_layernorm_bwd(x.data_ptr(), C, gamma.data_ptr(), dy.data_ptr(), dy.stride(0), L.ptr(dx_add), C, dx.data_ptr(), C,
                                       dg.data_ptr(), db.data_ptr(), int(acc), scratch.data_ptr(), rows, C, eps, L.stream_ptr()), "sp3_layernorm_bwd")
    if acc:
        _contributed(direct[0])
        _contributed(direct[1])
        return dx, None, None
    return dx, dg, db


class _MemoryReadTrain(torch.autograd.Function):
    """one batch element: feat [P,C], mem_k / mem_v [T,C], mask [P,T] or None"""

    @staticmethod
    def forward(ctx, feat, mem_k, mem_v, gq, bq, gk, bk, gv, bv, mask, eps, attn_accum=None):
        P, C = feat.shape
        T = mem_k.shape[0]
        Tp = _r8(T)
        dev = feat.device
        f32 = lambda *s: torch.empty(*s, device=dev)
        qh, kh = f32(P, C), f32(T, C)
        ops.layernorm(feat, gq, bq, eps, qh, rows=P, C_=C)
        ops.layernorm(mem_k, gk, bk, eps, kh, rows=T, C_=C)
        vht = torch.zeros(C, Tp, device=dev)                          # LN_v(mem_v)^T, k-extent padded to 8 with zeros
        ops.layernorm(mem_v, gv, bv, eps, vht, rows=T, C_=C, ldo=Tp, transposed=True)
        S = f32(P, Tp)
        alpha = 1.0 / (C ** 0.5)
        ops.gemm(qh, kh, S, M=P, N=T, K=C, lda=C, ldc=Tp, alpha=alpha)
        A = f32(P, Tp)
        ops.softmax_thresh(S, A, ld=Tp, rows=P, M=T, Mpad=Tp, thresh=0.0)          # columns [T, Tp) = 0
        Ad = A
        if mask is not None:
            if mask.shape != (P, T) or mask.dtype != torch.float32:
                raise ValueError("mask must be float32 [P, T] (0 or 1 / (1 - p))")
            mp = torch.zeros(P, Tp, device=dev)
            mp[:, :T] = mask
            Ad = f32(P, Tp)
            L.check(L.load().sp3_mul(A.data_ptr(), mp.data_ptr(), Ad.data_ptr(), P * Tp, L.stream_ptr()), "sp3_mul")
            mask = mp
        out = f32(P, C)
        ops.gemm(Ad, vht, out, M=P, N=C, K=Tp, lda=Tp, ldc=C, res1=feat, ldr1=C)
        if attn_accum is not None:             # spann3r/model.py:180-181: mem_attn += column sums of the (dropped-out) attention
            ops.colsum_accum(Ad, Tp, P, T, attn_accum)
        ctx.save_for_backward(feat, mem_k, mem_v, gq, gk, gv, qh, kh, vht, A, Ad, mask if mask is not None else torch.empty(0, device=dev))
        ctx.eps, ctx.alpha = eps, alpha
        return out

    @staticmethod
    def backward(ctx, dO):
        feat, mem_k, mem_v, gq, gk, gv, qh, kh, vht, A, Ad, mask = ctx.saved_tensors
        mask = mask if mask.numel() else None
        P, C = feat.shape
        T, Tp, Pp = mem_k.shape[0], A.shape[1], _r8(P)
        dev = feat.device
        dO = dO.contiguous().float()
        f32 = lambda *s: torch.empty(*s, device=dev)
        # dAd = dO . V_hat^T  ->  W[N = T, K = C] = V_hat (row-major): the transpose of the stored V_hat^T
        vh = _transpose(vht, C, T, Tp, ld_dst=C)                       # [T, C]
        dAd = f32(P, Tp)                                                # (columns [T, Tp) are never read)
        ops.gemm(dO, vh, dAd, M=P, N=T, K=C, lda=C, ldc=Tp)
        # dV_hat = Ad^T . dO  ->  A = Ad^T [T, Pp], W[N = C, K = Pp] = dO^T
        AdT = _transpose(Ad, P, T, Tp)                                  # [T, Pp] (pad columns zero)
        dOT = _transpose(dO, P, C, C)                                   # [C, Pp]
        dVh = f32(T, C)
        ops.gemm(AdT, dOT, dVh, M=T, N=C, K=Pp, lda=Pp, ldc=C)
        # softmax (+ dropout) backward, scaled by 1 / sqrt(C)
        dS = f32(P, Tp)                                                 # (pad columns zeroed by the kernel)
        L.check(L.load().sp3_softmax_bwd_pad(A.data_ptr(), dAd.data_ptr(), L.ptr(mask), dS.data_ptr(), Tp, P, T, Tp, ctx.alpha, L.stream_ptr()),
                "sp3_softmax_bwd_pad")
        # dq_hat = dS . K_hat  ->  W[N = C, K = Tp] = K_hat^T ;  dK_hat = dS^T . q_hat  ->  A = dS^T [T, Pp], W[N = C, K = Pp] = q_hat^T
        khT = _transpose(kh, T, C, C)                                   # [C, Tp]
        dqh = f32(P, C)
        ops.gemm(dS, khT, dqh, M=P, N=C, K=Tp, lda=Tp, ldc=C)
        dST = _transpose(dS, P, T, Tp)                                  # [T, Pp]
        qhT = _transpose(qh, P, C, C)                                   # [C, Pp]
        dKh = f32(T, C)
        ops.gemm(dST, qhT, dKh, M=T, N=C, K=Pp, lda=Pp, ldc=C)
        # LayerNorm backwards; the residual `+ feat` adds dO to d feat
        dfeat, dgq, dbq = _ln_bwd(feat, gq, dqh, dO, ctx.eps)
        dk, dgk, dbk = _ln_bwd(mem_k, gk, dKh, None, ctx.eps)
        dv, dgv, dbv = _ln_bwd(mem_v, gv, dVh, None, ctx.eps)
        return dfeat, dk, dv, dgq, dbq, dgk, dbk, dgv, dbv, None, None, None


def memory_read_train(feat, mem_k, mem_v, norm_q, norm_k, norm_v, mask=None, eps=1e-5, attn_accum=None):
    """feat [B,P,C], mem_k / mem_v [B,T,C] (fp32, device); norm_* = (weight, bias) of the three LayerNorms; mask [B,P,T]
    float32 dropout mask (0 or 1/(1-p)) or None.  -> out [B,P,C] = memory_read(feat, res=True) of the reference in train mode.
    attn_accum fp32 [B,T] (contiguous; no gradient): += the column sums of the attention after dropout (the reference's mem_attn)."""
    if not feat.is_cuda:
        raise RuntimeError("memory_read_train runs on the GPU (HIP kernels); there is no CPU path")
    outs = []
    for b in range(feat.shape[0]):
        outs.append(_MemoryReadTrain.apply(feat[b].contiguous().float(), mem_k[b].contiguous().float(), mem_v[b].contiguous().float(),
                                           norm_q[0], norm_q[1], norm_k[0], norm_k[1], norm_v[0], norm_v[1],
                                           None if mask is None else mask[b].contiguous(), eps,
                                           None if attn_accum is None else attn_accum[b]))
    return torch.stack(outs)


# =====================================================================================================================
# Train-mode building blocks with HIP forward AND backward (fp32 MFMA): every matrix product of a backward pass is an
# sp3_gemm launch (C = A . B^T only, so the other operand orders go through sp3_transpose / sp3_transpose_batched with the
# contraction extent zero-padded to 8), the elementwise pieces are the kernels of csrc/train.hip.  torch is the autograd
# tape, the reshapes between [B, N, C] and per-head layouts, and the accumulation of gradients at fan-outs.
# Reference modules: croco/models/blocks.py:73-79 (Mlp), :94-112 (Attention), :127-130 (Block), :149-169 (CrossAttention),
# :186-191 (DecoderBlock).
def _nt(A, Bm, M, N, K, lda, ldb, out, ldc, bias=None, res=None, alpha=1.0, batch=1, sA=0, sB=0, sC=0):
    """out[M, N] (row stride ldc) = alpha * A[M, K] . Bm[N, K]^T (+ bias) (+ res); K % 8 == 0 (operands zero-padded)"""
    # short contractions (the per-head products of attention: K = 64 or ~200): the 32x32 tile splits K over its 4 waves and would
    # leave half of them without a k-block; the 64x64 tile gives every wave its own 32x32 sub-tile
    tile = 1 if (K <= 256 and batch > 1 and M >= 64) else -1
    ops.gemm(A, Bm, out, M=M, N=N, K=K, lda=lda, ldc=ldc, ldw=ldb, bias=bias, res1=res, ldr1=ldc, alpha=alpha,
             batch=batch, strideA=sA, strideW=sB, strideC=sC, tile=tile)
    return out


def _tb(src, rows, cols, ld_src, batch=1, stride_src=0):
    """batched transpose with the new row length padded to 8 (zeros): [batch, rows, cols] -> [batch, cols, r8(rows)]"""
    ldd = _r8(rows)
    dst = torch.empty(batch, cols, ldd, device=src.device)          # (pad columns zeroed by the kernel)
    L.check(L.load().sp3_transpose_pad(src.data_ptr(), ld_src, stride_src, dst.data_ptr(), ldd, cols * ldd, rows, cols, batch, ldd,
                                       L.stream_ptr()), "sp3_transpose_pad")
    return dst


def _bf16_tile(M, N):
    """Tile of a training Linear's bf16 GEMMs (forward, dX, dW at ~800 rows; tools/bench_gemm2.py --train on MI355X): the 128x128
    pipelined tile as soon as it yields ~100 workgroups, else the 32x32 pipelined one (600-1000 workgroups, 7-9 us instead of the
    128x64 tile's 13 with 72-112 workgroups on 256 CUs); large problems keep the library's own choice."""
    if M < 512:
        return -1
    mt256, mt128, nt128 = (M + 255) // 256, (M + 127) // 128, (N + 127) // 128
    if mt256 * nt128 >= 144:
        return -1
    return 21 if mt128 * nt128 >= 100 else 25


def _pad8(x):
    """[R, N] -> contiguous [R, r8(N)] with zero columns (no copy when N % 8 == 0)"""
    R, N = x.shape
    if N % 8 == 0 and x.is_contiguous():
        return x
    y = torch.zeros(R, _r8(N), device=x.device)
    y[:, :N] = x
    return y


class _Linear(torch.autograd.Function):
    """y = x W^T + b (+ res) (+ res2); x [R, K], W [N, K].
    conv = stride (bf16 mode only): x is an NHWC map [B, H, W, Cin] and the product runs on its 3x3 / pad 1 im2col matrix
    [B*OH*OW, 9*Cin] -- gathered inside the pack launch (sp3_pack_bf16_conv3x3), never materialised; the backward's d col goes through
    sp3_col2im3x3 and comes back as the gradient of the map.
    act_in = 1 (GELU) / 2 (ReLU), bf16 mode only: the product runs on act(x), applied as the pack launch loads x (no activation launch, no
    fp32 act(x)); the backward multiplies d act(x) by act'(x) (sp3_gelu_bwd / sp3_relu_bwd on the saved x)."""

    @staticmethod
    def forward(ctx, x, W, b, res, res2, wkey, conv=None, act_in=0):
        N = W.shape[0]
        ctx.conv = None
        ctx.act_in = act_in
        if conv is not None:
            assert PRECISION == "bf16"
            Bc, Hc, Wc, Cin = x.shape
            OH, OW = (Hc - 1) // conv + 1, (Wc - 1) // conv + 1
            R, K = Bc * OH * OW, 9 * Cin
            ctx.conv = (Bc, Hc, Wc, Cin, conv)
        else:
            R, K = x.shape
        y = torch.empty(R, N, device=x.device)
        ctx.has = (b is not None, res is not None, res2 is not None)
        ctx.bf16 = PRECISION == "bf16"
        ctx.bparam = b if (b is not None and ctx.needs_input_grad[2] and isinstance(b, torch.nn.Parameter)) else None
        if ctx.bparam is not None:
            _expect(b)
        if ctx.bf16:
            # bf16 operands in fragment order; X^T is made in the same pass and is all the backward keeps of x
            need_t = ctx.needs_input_grad[1]
            xp, xT = ops.pack_bf16_conv3x3(x, conv, True, need_t, act=act_in) if conv is not None else ops.pack_bf16(x, True, need_t, act=act_in)
            if act_in and ctx.needs_input_grad[0]:
                ctx.save_for_backward(x)
            Wp, WT = _packed_weight(W, wkey)
            ops.gemm(xp, Wp, y, M=R, N=N, K=_r64(K), lda=K, ldc=N, bias=b, res1=res, ldr1=N, res2=res2, ldr2=N, tile=_bf16_tile(R, N))
            ctx.xT, ctx.WT, ctx.shape = xT, WT, (R, K, N)
            ctx.Wparam = W if (isinstance(W, torch.nn.Parameter) and W.dim() == 2) else None
            if ctx.Wparam is not None and need_t:
                _expect(W)
            return y
        assert not act_in, "act_in is a bf16-mode fusion"
        xp, Wp = _pad8(x), _pad8(W)
        ops.gemm(xp, Wp, y, M=R, N=N, K=xp.shape[1], lda=xp.shape[1], ldc=N, ldw=Wp.shape[1], bias=b, res1=res, ldr1=N, res2=res2, ldr2=N)
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.bf16 and ctx.WT is None:
            raise RuntimeError("spann3r_amd.train: second backward through a bf16 Linear (its packed operands are released by the "
                               "first one; retain_graph=True is not supported)")
        dy = dy.contiguous()
        dev = dy.device
        dx = dW = db = None
        bias_done = False
        if ctx.bf16:
            R, K, N = ctx.shape
            need_w = ctx.needs_input_grad[1]
            need_b = ctx.has[0] and ctx.needs_input_grad[2]
            dyp = dyT = None
            if ctx.needs_input_grad[0] or need_w:
                if need_b and R <= 8192:            # the bias gradient rides in the pack launch (straight into the bucket when it may)
                    direct = _direct_ok(ctx.bparam)
                    tgt = ctx.bparam.grad if direct else torch.empty(N, device=dev)
                    dyp, dyT = ops.pack_bf16(dy, ctx.needs_input_grad[0], need_w, colsum=tgt, accumulate=direct)
                    if direct:
                        _contributed(ctx.bparam)
                    else:
                        db = tgt
                    bias_done = True
                else:
                    dyp, dyT = ops.pack_bf16(dy, ctx.needs_input_grad[0], need_w)
            if ctx.needs_input_grad[0]:           # dX = dY . W = dY . (W^T)^T: contraction over N
                dx = torch.empty(R, K, device=dev)
                ops.gemm(dyp, ctx.WT, dx, M=R, N=K, K=_r64(N), lda=N, ldc=K, tile=_bf16_tile(R, K))
                if ctx.conv is not None:          # d col -> d map (the adjoint of the gather)
                    Bc, Hc, Wc, Cin, stride = ctx.conv
                    dmap = torch.empty(Bc, Hc, Wc, Cin, device=dev)
                    L.check(L.load().sp3_col2im3x3(dx.data_ptr(), dmap.data_ptr(), Bc, Hc, Wc, Cin, stride, L.stream_ptr()), "sp3_col2im3x3")
                    dx = dmap
                if ctx.act_in:                    # d act(x) -> d x, in place (element i reads and writes only its own slot)
                    (xraw,) = ctx.saved_tensors
                    fn = L.load().sp3_gelu_bwd if ctx.act_in == 1 else L.load().sp3_relu_bwd
                    L.check(fn(xraw.data_ptr(), dx.data_ptr(), dx.data_ptr(), dx.numel(), L.stream_ptr()), "sp3_gelu_bwd / sp3_relu_bwd")
            if need_w:                            # dW = dY^T . X = (dY^T) . (X^T)^T: contraction over the rows
                xTw = ops.PackedWeight.wrap(ctx.xT.data, K, R)
                def dw_into(out, acc):
                    ops.gemm(dyT, xTw, out, M=N, N=K, K=_r64(R), lda=R, ldc=K, res1=out if acc else None, ldr1=K, tile=_bf16_tile(N, K))
                if not _into_grad(ctx.Wparam, dw_into):
                    dW = torch.empty(N, K, device=dev)
                    dw_into(dW, False)
            ctx.xT = ctx.WT = None
        else:
            x, W = ctx.saved_tensors
            R, K = x.shape
            N = W.shape[0]
            if ctx.needs_input_grad[0]:
                WT = _tb(W, N, K, K)[0]                                      # [K, r8(N)]
                dyp = _pad8(dy)
                dx = _nt(dyp, WT, R, K, dyp.shape[1], dyp.shape[1], WT.shape[1], torch.empty(R, K, device=dev), K)
            if ctx.needs_input_grad[1]:
                dyT, xT = _tb(dy, R, N, N)[0], _tb(x, R, K, K)[0]            # [N, r8(R)], [K, r8(R)]
                dW = _nt(dyT, xT, N, K, dyT.shape[1], dyT.shape[1], xT.shape[1], torch.empty(N, K, device=dev), K)
        if ctx.has[0] and ctx.needs_input_grad[2] and not bias_done:
            if not _into_grad(ctx.bparam, lambda out, acc: _colsum(dy, out=out)):
                db = _colsum(dy)
        return dx, dW, db, (dy if ctx.has[1] else None), (dy if ctx.has[2] else None), None, None, None


def linear(x, W, b=None, res=None, res2=None, wkey=None, act_in=0):
    """wkey: cache key of the packed bf16 copies of W (default: the parameter itself when W is a leaf; derived matrices -- a
    permuted convolution weight -- pass (id(parameter), tag))"""
    sh = x.shape
    flat = lambda t: None if t is None else t.reshape(-1, W.shape[0]).contiguous()
    if wkey is None and W.is_leaf:
        wkey = (id(W), "linear", W._version, W.data_ptr())
    y = _Linear.apply(x.reshape(-1, sh[-1]).contiguous(), W.contiguous(), b, flat(res), flat(res2), wkey, None, act_in)
    return y.reshape(*sh[:-1], W.shape[0])


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        R, C_ = x.shape
        y = torch.empty_like(x)
        ops.layernorm(x, g, b, eps, y, rows=R, C_=C_)
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        ctx.direct = None
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and isinstance(g, torch.nn.Parameter) and isinstance(b, torch.nn.Parameter):
            ctx.direct = (g, b)
            _expect(g)
            _expect(b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dx, dg, db = _ln_bwd(x, g, dy.contiguous(), None, ctx.eps, ctx.direct)
        return dx, dg, db, None


def layer_norm(x, g, b, eps):
    sh = x.shape
    return _LayerNorm.apply(x.reshape(-1, sh[-1]).contiguous(), g, b, eps).reshape(sh)


class _LayerNormRes(torch.autograd.Function):
    """x -> (x, LN(x)) for the pre-LN residual pattern x + f(LN(x)): the gradient arriving on the pass-through x is added inside the
    LayerNorm backward kernel (dx_add) instead of by a separate accumulation launch"""

    @staticmethod
    def forward(ctx, x, g, b, eps):
        R, C_ = x.shape
        y = torch.empty_like(x)
        ops.layernorm(x, g, b, eps, y, rows=R, C_=C_)
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        ctx.direct = None
        ctx.set_materialize_grads(False)
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and isinstance(g, torch.nn.Parameter) and isinstance(b, torch.nn.Parameter):
            ctx.direct = (g, b)
            _expect(g)
            _expect(b)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dxp, dy):
        x, g = ctx.saved_tensors
        if dy is None:
            if ctx.direct is not None:
                _contributed(ctx.direct[0])
                _contributed(ctx.direct[1])
            return dxp, None, None, None
        dx, dg, db = _ln_bwd(x, g, dy.contiguous(), None if dxp is None else dxp.contiguous(), ctx.eps, ctx.direct)
        return dx, dg, db, None


def layer_norm_res(x, g, b, eps):
    """-> (x, LN(x)); use the returned x as the residual operand"""
    sh = x.shape
    xp, y = _LayerNormRes.apply(x.reshape(-1, sh[-1]).contiguous(), g, b, eps)
    return xp.reshape(sh), y.reshape(sh)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        L.check(L.load().sp3_gelu(x.data_ptr(), y.data_ptr(), x.numel(), L.stream_ptr()), "sp3_gelu")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        L.check(L.load().sp3_gelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), L.stream_ptr()), "sp3_gelu_bwd")
        return dx


class _Attention(torch.autograd.Function):
    """softmax(q k^T * scale) v on [BH, Nq, D] x [BH, Nk, D] (D % 8 == 0), heads as the batch of grouped GEMM launches"""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        BH, Nq, D = q.shape
        Nk = k.shape[1]
        Nkp = _r8(Nk)
        dev = q.device
        S = torch.empty(BH, Nq, Nkp, device=dev)                          # (the softmax reads Nk columns and zeroes the pad of P)
        _nt(q, k, Nq, Nk, D, D, D, S, Nkp, alpha=scale, batch=BH, sA=Nq * D, sB=Nk * D, sC=Nq * Nkp)
        P = torch.empty_like(S)
        ops.softmax_thresh(S, P, ld=Nkp, rows=BH * Nq, M=Nk, Mpad=Nkp, thresh=0.0)
        vT = _tb(v, Nk, D, D, BH, Nk * D)                                # [BH, D, Nkp]
        O = torch.empty(BH, Nq, D, device=dev)
        _nt(P, vT, Nq, D, Nkp, Nkp, Nkp, O, D, batch=BH, sA=Nq * Nkp, sB=D * Nkp, sC=Nq * D)
        ctx.save_for_backward(q, k, v, P)
        ctx.scale = scale
        return O

    @staticmethod
    def backward(ctx, dO):
        q, k, v, P = ctx.saved_tensors
        BH, Nq, D = q.shape
        Nk, Nkp, Nqp = k.shape[1], P.shape[2], _r8(q.shape[1])
        dev = q.device
        dO = dO.contiguous()
        dP = torch.empty(BH, Nq, Nkp, device=dev)
        _nt(dO, v, Nq, Nk, D, D, D, dP, Nkp, batch=BH, sA=Nq * D, sB=Nk * D, sC=Nq * Nkp)
        PT = _tb(P, Nq, Nk, Nkp, BH, Nq * Nkp)                           # [BH, Nk, Nqp]
        dOT = _tb(dO, Nq, D, D, BH, Nq * D)                              # [BH, D, Nqp]
        dV = torch.empty(BH, Nk, D, device=dev)
        _nt(PT, dOT, Nk, D, Nqp, Nqp, Nqp, dV, D, batch=BH, sA=Nk * Nqp, sB=D * Nqp, sC=Nk * D)
        dS = torch.empty(BH, Nq, Nkp, device=dev)
        L.check(L.load().sp3_softmax_bwd_pad(P.data_ptr(), dP.data_ptr(), None, dS.data_ptr(), Nkp, BH * Nq, Nk, Nkp, ctx.scale, L.stream_ptr()),
                "sp3_softmax_bwd_pad")
        kT = _tb(k, Nk, D, D, BH, Nk * D)                                # [BH, D, Nkp]
        dq = torch.empty(BH, Nq, D, device=dev)
        _nt(dS, kT, Nq, D, Nkp, Nkp, Nkp, dq, D, batch=BH, sA=Nq * Nkp, sB=D * Nkp, sC=Nq * D)
        dST = _tb(dS, Nq, Nk, Nkp, BH, Nq * Nkp)                         # [BH, Nk, Nqp]
        qT = _tb(q, Nq, D, D, BH, Nq * D)                                # [BH, D, Nqp]
        dk = torch.empty(BH, Nk, D, device=dev)
        _nt(dST, qT, Nk, D, Nqp, Nqp, Nqp, dk, D, batch=BH, sA=Nk * Nqp, sB=D * Nqp, sC=Nk * D)
        return dq, dk, dV, None


_rope_tabs = {}


def _rope_table(base, hd, device, n=256):
    """cos / sin tables [n, hd/4] of RoPE2D exactly as the reference builds them (croco/models/pos_embed.py:118-129 get_cos_sin: fp32
    inv_freq = 1 / base^(2i / D), fp32 angles t * inv_freq, fp32 cos / sin), so the rotation uses the reference's own rounded angles"""
    key = (float(base), hd, str(device))
    if key not in _rope_tabs:
        D = hd // 2
        inv_freq = 1.0 / (float(base) ** (torch.arange(0, D, 2).float() / D))
        ang = torch.einsum("i,j->ij", torch.arange(n, dtype=torch.float32), inv_freq)
        _rope_tabs[key] = (ang.cos().to(device).contiguous(), ang.sin().to(device).contiguous(), n)
    return _rope_tabs[key]


def _head_shuffle(parts, B, H, hd, base):
    """parts: up to three dicts (src, s = (s_b, s_n, s_h), N, and dst + d = (d_b, d_n, d_h) and / or dstT, pos, fwd): sp3_head_shuffle"""
    arr = (L.HeadPart * len(parts))()
    for a, p in zip(arr, parts):
        a.src, (a.s_b, a.s_n, a.s_h), a.N = p["src"].data_ptr(), p["s"], p["N"]
        dst, dstT, pos = p.get("dst"), p.get("dstT"), p.get("pos")
        a.dst = L.ptr(dst)
        if dst is not None:
            a.d_b, a.d_n, a.d_h = p["d"]
        a.dstT, a.pos, a.fwd, a.ldT = L.ptr(dstT), L.ptr(pos), float(p.get("fwd", 1.0)), int(p.get("ldT", 0))
    ct, st, n = _rope_table(base, hd, parts[0]["src"].device) if any(p.get("pos") is not None for p in parts) else (None, None, 0)
    L.check(L.load().sp3_head_shuffle(arr, len(parts), B, H, hd, float(base), L.ptr(ct), L.ptr(st), n, L.stream_ptr()), "sp3_head_shuffle")


class _MHA(torch.autograd.Function):
    """Multi-head attention between the projections (croco/models/blocks.py:100-108 and :160-166): RoPE2D on q and k, the per-head
    products, the head merge -- on the projections' own outputs.  Self-attention: a = qkv [B, N, 3C] (b = c = None); cross-attention:
    a = q [B, Nq, C], b = k, c = v [B, Nk, C].  Forward: 1 shuffle + S GEMM + softmax + P.V GEMM + 1 shuffle; backward: 1 shuffle
    (dO per head and transposed) + 4 GEMMs + 2 transposes + softmax backward + 1 shuffle (inverse RoPE, head merge into d qkv).
    Saved for the backward: v, q^T, k^T per head and the probabilities."""

    @staticmethod
    def forward(ctx, a, b, c, posq, posk, heads, scale, base):
        self_mode = b is None
        a = a.contiguous()
        if not self_mode:
            b, c = b.contiguous(), c.contiguous()
        B, Nq = a.shape[0], a.shape[1]
        C_ = a.shape[2] // 3 if self_mode else a.shape[2]
        Nk = Nq if self_mode else b.shape[1]
        H, hd = heads, C_ // heads
        BH, Nqp, Nkp = B * H, _r8(Nq), _r8(Nk)
        dev = a.device
        f32 = lambda *sh: torch.empty(*sh, device=dev)
        need = any(ctx.needs_input_grad[:3])
        q, k, v = f32(BH, Nq, hd), f32(BH, Nk, hd), f32(BH, Nk, hd)
        qT, kT, vT = (f32(BH, hd, Nqp) if need else None), (f32(BH, hd, Nkp) if need else None), f32(BH, hd, Nkp)
        if self_mode:
            srcs = [(a, 0), (a, C_), (a, 2 * C_)]
        else:
            srcs = [(a, 0), (b, 0), (c, 0)]
        parts = []
        for (t, off), N, dst, dstT, pos in zip(srcs, (Nq, Nk, Nk), (q, k, v), (qT, kT, vT), (posq, posk, None)):
            ld = t.shape[2]
            parts.append(dict(src=t[0, 0, off:], s=(N * ld, ld, hd), N=N, dst=dst, d=(H * N * hd, hd, N * hd), dstT=dstT, pos=pos, fwd=1.0))
        _head_shuffle(parts, B, H, hd, base)
        S = f32(BH, Nq, Nkp)
        _nt(q, k, Nq, Nk, hd, hd, hd, S, Nkp, alpha=scale, batch=BH, sA=Nq * hd, sB=Nk * hd, sC=Nq * Nkp)
        P = S                                                              # in place: a thread reads and writes its own columns
        ops.softmax_thresh(S, P, ld=Nkp, rows=BH * Nq, M=Nk, Mpad=Nkp, thresh=0.0)
        O = f32(BH, Nq, hd)
        _nt(P, vT, Nq, hd, Nkp, Nkp, Nkp, O, hd, batch=BH, sA=Nq * Nkp, sB=hd * Nkp, sC=Nq * hd)
        out = f32(B, Nq, C_)
        _head_shuffle([dict(src=O, s=(H * Nq * hd, hd, Nq * hd), N=Nq, dst=out, d=(Nq * C_, C_, hd))], B, H, hd, base)
        if need:
            ctx.save_for_backward(v, qT, kT, P, posq if posq is not None else torch.empty(0, device=dev), posk if posk is not None else torch.empty(0, device=dev))
        ctx.geo = (self_mode, B, Nq, Nk, C_, H, hd, scale, base, a.shape[2], (None if self_mode else (b.shape[2], c.shape[2])))
        return out

    @staticmethod
    def backward(ctx, dOm):
        v, qT, kT, P, posq, posk = ctx.saved_tensors
        self_mode, B, Nq, Nk, C_, H, hd, scale, base, lda, ldbc = ctx.geo
        posq = posq if posq.numel() else None
        posk = posk if posk.numel() else None
        BH, Nqp, Nkp = B * H, _r8(Nq), _r8(Nk)
        dev = dOm.device
        f32 = lambda *sh: torch.empty(*sh, device=dev)
        dOm = dOm.contiguous()
        dO, dOT = f32(BH, Nq, hd), f32(BH, hd, Nqp)
        _head_shuffle([dict(src=dOm, s=(Nq * C_, C_, hd), N=Nq, dst=dO, d=(H * Nq * hd, hd, Nq * hd), dstT=dOT)], B, H, hd, base)
        dP = f32(BH, Nq, Nkp)
        _nt(dO, v, Nq, Nk, hd, hd, hd, dP, Nkp, batch=BH, sA=Nq * hd, sB=Nk * hd, sC=Nq * Nkp)
        PT = _tb(P, Nq, Nk, Nkp, BH, Nq * Nkp)                           # [BH, Nk, Nqp]
        dV = f32(BH, Nk, hd)
        _nt(PT, dOT, Nk, hd, Nqp, Nqp, Nqp, dV, hd, batch=BH, sA=Nk * Nqp, sB=hd * Nqp, sC=Nk * hd)
        dS = dP                                                            # in place (same reasoning as the forward softmax)
        L.check(L.load().sp3_softmax_bwd_pad(P.data_ptr(), dP.data_ptr(), None, dS.data_ptr(), Nkp, BH * Nq, Nk, Nkp, scale, L.stream_ptr()),
                "sp3_softmax_bwd_pad")
        dq = f32(BH, Nq, hd)
        _nt(dS, kT, Nq, hd, Nkp, Nkp, Nkp, dq, hd, batch=BH, sA=Nq * Nkp, sB=hd * Nkp, sC=Nq * hd)
        dST = _tb(dS, Nq, Nk, Nkp, BH, Nq * Nkp)                         # [BH, Nk, Nqp]
        dk = f32(BH, Nk, hd)
        _nt(dST, qT, Nk, hd, Nqp, Nqp, Nqp, dk, hd, batch=BH, sA=Nk * Nqp, sB=hd * Nqp, sC=Nk * hd)
        if self_mode:
            da = f32(B, Nq, 3 * C_)
            dsts = [(da, 0), (da, C_), (da, 2 * C_)]
        else:
            da, db, dc = f32(B, Nq, C_), f32(B, Nk, C_), f32(B, Nk, C_)
            dsts = [(da, 0), (db, 0), (dc, 0)]
        parts = []
        for (t, off), N, src, pos in zip(dsts, (Nq, Nk, Nk), (dq, dk, dV), (posq, posk, None)):
            ld = t.shape[2]
            parts.append(dict(src=src, s=(H * N * hd, hd, N * hd), N=N, dst=t[0, 0, off:], d=(N * ld, ld, hd), pos=pos, fwd=-1.0))
        _head_shuffle(parts, B, H, hd, base)
        if self_mode:
            return da, None, None, None, None, None, None, None
        return da, db, dc, None, None, None, None, None


class _FlashMHA(torch.autograd.Function):
    """The same op as _MHA (heads of 64) without the attention matrix: sp3_attention_train_fwd / _bwd recompute the probabilities
    from the saved log-sum-exp.  Forward: 1 shuffle (RoPE'd q, k in token-major layout, V^T, and q^T / k^T for the backward) + 1
    attention launch writing the merged output; backward: 1 shuffle (dO^T) + 2 launches (dq | dk, dv, written straight into the
    gradient of the projections' output) + 1 in-place shuffle (inverse RoPE on dq, dk).  v is read in place from the projection
    output.  Saved: the projection output(s), RoPE'd q / k, q^T, k^T and lse."""

    @staticmethod
    def forward(ctx, a, b, c, posq, posk, heads, scale, base):
        self_mode = b is None
        a = a.contiguous()
        if not self_mode:
            b, c = b.contiguous(), c.contiguous()
        B, Nq = a.shape[0], a.shape[1]
        C_ = a.shape[2] // 3 if self_mode else a.shape[2]
        Nk = Nq if self_mode else b.shape[1]
        H, hd = heads, C_ // heads
        assert hd == 64
        BH, Tq, Tk = B * H, _r64(Nq), _r64(Nk)
        dev = a.device
        f32 = lambda *sh: torch.empty(*sh, device=dev)
        need = any(ctx.needs_input_grad[:3])
        # sources: (tensor, column offset, row length)
        qs, ks, vs = ((a, 0, 3 * C_), (a, C_, 3 * C_), (a, 2 * C_, 3 * C_)) if self_mode else ((a, 0, C_), (b, 0, C_), (c, 0, C_))
        vT = f32(BH, hd, Tk)
        qT, kT = (f32(BH, hd, Tq), f32(BH, hd, Tk)) if need else (None, None)
        parts = [dict(src=vs[0][0, 0, vs[1]:], s=(Nk * vs[2], vs[2], hd), N=Nk, dstT=vT, ldT=Tk)]
        rope = posq is not None
        if rope:            # rotated copies in token-major layout [B, N, C]
            qr, kr = f32(B, Nq, C_), f32(B, Nk, C_)
            parts.append(dict(src=qs[0][0, 0, qs[1]:], s=(Nq * qs[2], qs[2], hd), N=Nq, dst=qr, d=(Nq * C_, C_, hd), dstT=qT, ldT=Tq, pos=posq, fwd=1.0))
            parts.append(dict(src=ks[0][0, 0, ks[1]:], s=(Nk * ks[2], ks[2], hd), N=Nk, dst=kr, d=(Nk * C_, C_, hd), dstT=kT, ldT=Tk, pos=posk, fwd=1.0))
            qv, kv = (qr, 0, C_), (kr, 0, C_)
        else:               # no rotation: q and k are read where the projection left them
            if need:
                parts.append(dict(src=qs[0][0, 0, qs[1]:], s=(Nq * qs[2], qs[2], hd), N=Nq, dstT=qT, ldT=Tq))
                parts.append(dict(src=ks[0][0, 0, ks[1]:], s=(Nk * ks[2], ks[2], hd), N=Nk, dstT=kT, ldT=Tk))
            qv, kv = qs, ks
        _head_shuffle(parts, B, H, hd, base)
        out, lse = f32(B, Nq, C_), f32(BH, Nq, 2)            # lse: (row max, 1 / row sum) per query
        bf = int(PRECISION == "bf16")
        qp, kp = qv[0][0, 0, qv[1]:], kv[0][0, 0, kv[1]:]
        L.check(L.load().sp3_attention_train_fwd(qp.data_ptr(), Nq * qv[2], qv[2], kp.data_ptr(), Nk * kv[2], kv[2], vT.data_ptr(), Tk,
                                                 out.data_ptr(), C_, lse.data_ptr(), B, H, Nq, Nk, float(scale), bf, L.stream_ptr()), "sp3_attention_train_fwd")
        if need:
            none = torch.empty(0, device=dev)
            ctx.save_for_backward(qv[0], kv[0], vs[0], qT, kT, lse, posq if rope else none, posk if rope else none)
        ctx.geo = (self_mode, rope, B, Nq, Nk, C_, H, scale, base, bf, qv[1:], kv[1:], vs[1:])
        return out

    @staticmethod
    def backward(ctx, dOm):
        qt, kt, vt_, qT, kT, lse, posq, posk = ctx.saved_tensors
        self_mode, rope, B, Nq, Nk, C_, H, scale, base, bf, (qo, qld), (ko, kld), (vo, vld) = ctx.geo
        hd, BH, Tq, Tk = 64, B * H, _r64(Nq), _r64(Nk)
        dev = dOm.device
        f32 = lambda *sh: torch.empty(*sh, device=dev)
        dOm = dOm.contiguous()
        doT = f32(BH, hd, Tq)
        _head_shuffle([dict(src=dOm, s=(Nq * C_, C_, hd), N=Nq, dstT=doT, ldT=Tq)], B, H, hd, base)
        if self_mode:
            da = f32(B, Nq, 3 * C_)
            gq, gk, gv = (da, 0, 3 * C_), (da, C_, 3 * C_), (da, 2 * C_, 3 * C_)
        else:
            da, db, dc = f32(B, Nq, C_), f32(B, Nk, C_), f32(B, Nk, C_)
            gq, gk, gv = (da, 0, C_), (db, 0, C_), (dc, 0, C_)
        D = f32(BH, Nq)
        d = L.AttnBwdDesc()
        P = lambda t, off: t[0, 0, off:].data_ptr()
        d.q, d.sq, d.ldq = P(qt, qo), Nq * qld, qld
        d.k, d.sk, d.ldk = P(kt, ko), Nk * kld, kld
        d.v, d.sv, d.ldv = P(vt_, vo), Nk * vld, vld
        d.dout, d.sdo, d.lddo = dOm.data_ptr(), Nq * C_, C_
        d.qT, d.kT, d.doT, d.ldTq, d.ldTk = qT.data_ptr(), kT.data_ptr(), doT.data_ptr(), Tq, Tk
        d.lse, d.D = lse.data_ptr(), D.data_ptr()
        d.dq, d.sdq, d.lddq = P(gq[0], gq[1]), Nq * gq[2], gq[2]
        d.dk, d.sdk, d.lddk = P(gk[0], gk[1]), Nk * gk[2], gk[2]
        d.dv, d.sdv, d.lddv = P(gv[0], gv[1]), Nk * gv[2], gv[2]
        d.B, d.heads, d.Nq, d.Nk, d.scale, d.bf16_products = B, H, Nq, Nk, float(scale), bf
        L.check(L.load().sp3_attention_train_bwd(ctypes.byref(d), L.stream_ptr()), "sp3_attention_train_bwd")
        if rope:            # the gradients of the rotated q, k -> of the projections' outputs: R^T in place
            parts = []
            for (t, off, ld), N, pos in ((gq, Nq, posq), (gk, Nk, posk)):
                v = t[0, 0, off:]
                parts.append(dict(src=v, s=(N * ld, ld, hd), N=N, dst=v, d=(N * ld, ld, hd), pos=pos, fwd=-1.0))
            _head_shuffle(parts, B, H, hd, base)
        if self_mode:
            return da, None, None, None, None, None, None, None
        return da, db, dc, None, None, None, None, None


def _mha(a, b, c, posq, posk, heads, scale, base):
    C_ = a.shape[2] // 3 if b is None else a.shape[2]
    fn = _FlashMHA if (FLASH_ATTENTION and C_ // heads == 64) else _MHA
    return fn.apply(a, b, c, posq, posk, heads, scale, base)


def _heads(t, B, N, H, pos, base):
    """[B, N, H*hd] -> [B*H, N, hd] contiguous, RoPE applied first if pos is given (in place on the [B, N, H, hd] copy)"""
    hd = t.shape[-1] // H
    t = t.reshape(B, N, H, hd).contiguous()
    if pos is not None:
        from .curope import cuRoPE2D_func
        t = cuRoPE2D_func.apply(t, pos, base, 1.0)
    return t.permute(0, 2, 1, 3).reshape(B * H, N, hd).contiguous()


def self_attention(x, pos, P, pre, heads, base=100.0, use_rope=True, res=None):
    """croco/models/blocks.py:94-112; P: dict of parameters, pre: key prefix ('...attn.'); res is added by the projection"""
    B, N, C_ = x.shape
    qkv = linear(x, P[pre + "qkv.weight"], P[pre + "qkv.bias"])
    rp = pos if use_rope else None
    if FUSED_HEADS:
        o = _mha(qkv, None, None, rp, rp, heads, (C_ // heads) ** -0.5, base)
    else:
        qkv = qkv.reshape(B, N, 3, C_)
        q, k, v = _heads(qkv[:, :, 0], B, N, heads, rp, base), _heads(qkv[:, :, 1], B, N, heads, rp, base), _heads(qkv[:, :, 2], B, N, heads, None, base)
        o = _Attention.apply(q, k, v, (C_ // heads) ** -0.5)
        o = o.reshape(B, heads, N, C_ // heads).permute(0, 2, 1, 3).reshape(B, N, C_)
    return linear(o, P[pre + "proj.weight"], P[pre + "proj.bias"], res)


def cross_attention(xq, y, qpos, kpos, P, pre, heads, base=100.0, res=None):
    """croco/models/blocks.py:149-169"""
    B, Nq, C_ = xq.shape
    Nk = y.shape[1]
    q = linear(xq, P[pre + "projq.weight"], P[pre + "projq.bias"])
    k = linear(y, P[pre + "projk.weight"], P[pre + "projk.bias"])
    v = linear(y, P[pre + "projv.weight"], P[pre + "projv.bias"])
    if FUSED_HEADS:
        o = _mha(q, k, v, qpos, kpos, heads, (C_ // heads) ** -0.5, base)
    else:
        q, k, v = _heads(q, B, Nq, heads, qpos, base), _heads(k, B, Nk, heads, kpos, base), _heads(v, B, Nk, heads, None, base)
        o = _Attention.apply(q, k, v, (C_ // heads) ** -0.5)
        o = o.reshape(B, heads, Nq, C_ // heads).permute(0, 2, 1, 3).reshape(B, Nq, C_)
    return linear(o, P[pre + "proj.weight"], P[pre + "proj.bias"], res)


def mlp(x, P, pre, res=None):
    """croco/models/blocks.py:73-79"""
    h = linear(x, P[pre + "fc1.weight"], P[pre + "fc1.bias"])
    if PRECISION == "bf16" and ACT_ON_LOAD:           # GELU applied as fc2's pack launch loads fc1's output
        return linear(h, P[pre + "fc2.weight"], P[pre + "fc2.bias"], res, act_in=1)
    return linear(_Gelu.apply(h), P[pre + "fc2.weight"], P[pre + "fc2.bias"], res)


def block(x, pos, P, pre, heads, base=100.0, use_rope=True, eps=1e-6):
    """Pre-LN ViT block, croco/models/blocks.py:127-130 (the residual adds ride in the GEMM epilogues)"""
    x, h = layer_norm_res(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], eps)
    x = self_attention(h, pos, P, pre + "attn.", heads, base, use_rope, res=x)
    x, h = layer_norm_res(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"], eps)
    return mlp(h, P, pre + "mlp.", res=x)


def decoder_block(x, y, xpos, ypos, P, pre, heads, base=100.0, eps=1e-6):
    """croco/models/blocks.py:186-191"""
    x, h = layer_norm_res(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], eps)
    x = self_attention(h, xpos, P, pre + "attn.", heads, base, res=x)
    yn = layer_norm(y, P[pre + "norm_y.weight"], P[pre + "norm_y.bias"], eps)
    x, h = layer_norm_res(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"], eps)
    x = cross_attention(h, yn, xpos, ypos, P, pre + "cross_attn.", heads, base, res=x)
    x, h = layer_norm_res(x, P[pre + "norm3.weight"], P[pre + "norm3.bias"], eps)
    return mlp(h, P, pre + "mlp.", res=x)


# ---------------------------------------------------------------------------------------------------------------------
# DPT head in train mode (dust3r/heads/dpt_head.py:34-65, croco/models/dpt_block.py:120-218,356-410, postprocess.py:10-58)
# on NHWC maps: every convolution is im2col (or a reshape) + the linear op above, so its backward is two GEMMs + col2im.
class _Im2col3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride):
        B, H, W, C_ = x.shape
        x = x.contiguous()
        OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
        col = torch.empty(B * OH * OW, 9 * C_, device=x.device)
        L.check(L.load().sp3_im2col3x3(x.data_ptr(), col.data_ptr(), B, H, W, C_, stride, L.stream_ptr()), "sp3_im2col3x3")
        ctx.geo = (B, H, W, C_, stride)
        return col

    @staticmethod
    def backward(ctx, dcol):
        B, H, W, C_, stride = ctx.geo
        dcol = dcol.contiguous()
        dx = torch.empty(B, H, W, C_, device=dcol.device)
        L.check(L.load().sp3_col2im3x3(dcol.data_ptr(), dx.data_ptr(), B, H, W, C_, stride, L.stream_ptr()), "sp3_col2im3x3")
        return dx, None


class _Relu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        L.check(L.load().sp3_relu(x.data_ptr(), y.data_ptr(), x.numel(), L.stream_ptr()), "sp3_relu")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        L.check(L.load().sp3_relu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), L.stream_ptr()), "sp3_relu_bwd")
        return dx


class _Upsample2x(torch.autograd.Function):
    """F.interpolate(scale_factor=2, bilinear, align_corners=True) on NHWC, optionally cropped to (outH, outW)"""

    @staticmethod
    def forward(ctx, x, outH, outW):
        B, H, W, C_ = x.shape
        x = x.contiguous()
        y = torch.empty(B, outH, outW, C_, device=x.device)
        ops.upsample2x(x, y, B=B, H=H, W_=W, C_=C_, outH=outH, outW=outW)
        ctx.geo = (B, H, W, C_, outH, outW)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C_, outH, outW = ctx.geo
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, C_, device=dy.device)
        L.check(L.load().sp3_upsample2x_bwd(dy.data_ptr(), dx.data_ptr(), B, H, W, C_, outH, outW, L.stream_ptr()), "sp3_upsample2x_bwd")
        return dx, None, None


class _Postprocess(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw):
        raw = raw.contiguous()
        M = raw.numel() // 4
        pts, conf = torch.empty(M, 3, device=raw.device), torch.empty(M, device=raw.device)
        L.check(L.load().sp3_postprocess(raw.data_ptr(), pts.data_ptr(), conf.data_ptr(), M, L.stream_ptr()), "sp3_postprocess")
        ctx.save_for_backward(raw)
        return pts.reshape(*raw.shape[:-1], 3), conf.reshape(raw.shape[:-1])

    @staticmethod
    def backward(ctx, dpts, dconf):
        (raw,) = ctx.saved_tensors
        M = raw.numel() // 4
        dpts = (torch.zeros(M, 3, device=raw.device) if dpts is None else dpts.contiguous())
        dconf = (torch.zeros(M, device=raw.device) if dconf is None else dconf.contiguous())
        draw = torch.empty_like(raw)
        L.check(L.load().sp3_postprocess_bwd(raw.data_ptr(), dpts.data_ptr(), dconf.data_ptr(), draw.data_ptr(), M, L.stream_ptr()),
                "sp3_postprocess_bwd")
        return draw


def conv3x3(x, W, b=None, stride=1, res=None, res2=None, relu_in=False):
    """x NHWC [B,H,W,Cin], W [Cout,Cin,3,3] (the reference's Conv2d layout), padding 1 -> [B,OH,OW,Cout]; relu_in: the convolution of relu(x)"""
    B, H, Wd, Cin = x.shape
    OH, OW = (H - 1) // stride + 1, (Wd - 1) // stride + 1
    Wm = W.permute(0, 2, 3, 1).reshape(W.shape[0], 9 * Cin)
    wkey = (id(W), "conv3x3", W._version)
    if PRECISION == "bf16" and CONV_GATHER and Cin % 4 == 0:
        flat = lambda t: None if t is None else t.reshape(-1, W.shape[0]).contiguous()
        fuse = relu_in and ACT_ON_LOAD
        y = _Linear.apply((x if fuse or not relu_in else _Relu.apply(x)).contiguous(), Wm.contiguous(), b, flat(res), flat(res2), wkey, stride, 2 if fuse else 0)
    else:
        y = linear(_Im2col3x3.apply(_Relu.apply(x) if relu_in else x, stride), Wm, b, res, res2, wkey=wkey)
    return y.reshape(B, OH, OW, W.shape[0])


def conv1x1(x, W, b=None):
    return linear(x, W.reshape(W.shape[0], -1), b, wkey=(id(W), "conv1x1", W._version))


def conv_transpose_ks(x, W, b, k):
    """ConvTranspose2d(kernel = stride = k): x NHWC [B,H,W,Cin], W [Cin,Cout,k,k] -> [B,kH,kW,Cout]"""
    B, H, Wd, Cin = x.shape
    Co = W.shape[1]
    Wm = W.permute(2, 3, 1, 0).reshape(k * k * Co, Cin)
    y = linear(x, Wm, b.repeat(k * k), wkey=(id(W), "convT", W._version))
    return y.reshape(B, H, Wd, k, k, Co).permute(0, 1, 3, 2, 4, 5).reshape(B, H * k, Wd * k, Co)


def _rcu(x, P, p, res2=None):
    """ResidualConvUnit_custom (dpt_block.py:120-142): conv2(relu(conv1(relu(x)))) + x (+ res2)"""
    o = conv3x3(x, P[p + "conv1.weight"], P[p + "conv1.bias"], relu_in=True)
    return conv3x3(o, P[p + "conv2.weight"], P[p + "conv2.bias"], res=x, res2=res2, relu_in=True)


def _fusion(P, p, x0, x1=None, crop=None):
    """FeatureFusionBlock_custom (dpt_block.py:190-218); the 1x1 out_conv is applied before the bilinear x2 (they commute)"""
    out = x0 if x1 is None else _rcu(x1, P, p + "resConfUnit1.", res2=x0)
    out = _rcu(out, P, p + "resConfUnit2.")
    out = conv1x1(out, P[p + "out_conv.weight"], P[p + "out_conv.bias"])
    H, W = out.shape[1:3]
    oh, ow = (2 * H, 2 * W) if crop is None else crop
    return _Upsample2x.apply(out, oh, ow)


def dpt_head(dec, nh, nw, P, cfg, num):
    """dec: list of dec_depth+1 token tensors [B, nh*nw, C] -> (pts3d [B,H,W,3], conf [B,H,W]); P holds the reference's
    parameters under their own names (dust3r.downstream_head{num}.dpt.*)"""
    p = "dust3r.downstream_head%d.dpt." % num
    B = dec[0].shape[0]
    Lr = [dec[h].reshape(B, nh, nw, -1) for h in cfg.hooks]
    a = p + "act_postprocess."
    l0 = conv_transpose_ks(conv1x1(Lr[0], P[a + "0.0.weight"], P[a + "0.0.bias"]), P[a + "0.1.weight"], P[a + "0.1.bias"], 4)
    l1 = conv_transpose_ks(conv1x1(Lr[1], P[a + "1.0.weight"], P[a + "1.0.bias"]), P[a + "1.1.weight"], P[a + "1.1.bias"], 2)
    l2 = conv1x1(Lr[2], P[a + "2.0.weight"], P[a + "2.0.bias"])
    l3 = conv3x3(conv1x1(Lr[3], P[a + "3.0.weight"], P[a + "3.0.bias"]), P[a + "3.1.weight"], P[a + "3.1.bias"], stride=2)
    rn = [conv3x3(l, P[p + "scratch.layer%d_rn.weight" % (i + 1)]) for i, l in enumerate((l0, l1, l2, l3))]
    r = p + "scratch.refinenet"
    p4 = _fusion(P, r + "4.", rn[3], crop=(nh, nw))
    p3 = _fusion(P, r + "3.", p4, rn[2])
    p2 = _fusion(P, r + "2.", p3, rn[1])
    p1 = _fusion(P, r + "1.", p2, rn[0])
    h = conv3x3(p1, P[p + "head.0.weight"], P[p + "head.0.bias"])
    h = _Upsample2x.apply(h, 2 * h.shape[1], 2 * h.shape[2])
    h = _Relu.apply(conv3x3(h, P[p + "head.2.weight"], P[p + "head.2.bias"]))
    raw = conv1x1(h, P[p + "head.4.weight"], P[p + "head.4.bias"])
    return _Postprocess.apply(raw)


# ---------------------------------------------------------------------------------------------------------------------
# Spann3R.forward in train mode (spann3r/model.py:473-539 with self.training: memory attn_thresh = 0, dropout on the read,
# unconditional add_mem) assembled from the ops above: every arithmetic step of the forward and of the backward is a HIP
# kernel; torch records the tape and moves tensors between layouts.
def _positions(B, nh, nw, device):
    ys, xs = torch.meshgrid(torch.arange(nh, device=device), torch.arange(nw, device=device), indexing="ij")
    return torch.stack((ys.reshape(-1), xs.reshape(-1)), -1)[None].expand(B, -1, -1).contiguous()


def _patchify(img, patch):
    """[B, C, H, W] -> [B, nh*nw, C*patch*patch] in the element order of a Conv2d(k = s = patch) weight"""
    B, C_, H, W = img.shape
    nh, nw = H // patch, W // patch
    return img.reshape(B, C_, nh, patch, nw, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, nh * nw, C_ * patch * patch), nh, nw


def encode_image(img, P, cfg):
    """dust3r/model.py:131-154"""
    x, nh, nw = _patchify(img.float(), cfg.patch)
    W = P["dust3r.patch_embed.proj.weight"]
    x = linear(x, W.reshape(W.shape[0], -1), P["dust3r.patch_embed.proj.bias"], wkey=(id(W), "patch", W._version))
    pos = _positions(img.shape[0], nh, nw, img.device)
    for i in range(cfg.enc_depth):
        x = block(x, pos, P, "dust3r.enc_blocks.%d." % i, cfg.enc_heads, cfg.rope_base)
    return layer_norm(x, P["dust3r.enc_norm.weight"], P["dust3r.enc_norm.bias"], 1e-6), pos, (nh, nw)


def decoder(f1, pos1, f2, pos2, P, cfg):
    """dust3r/model.py:186-205"""
    outs1, outs2 = [f1], [f2]
    a = linear(f1, P["dust3r.decoder_embed.weight"], P["dust3r.decoder_embed.bias"])
    b = linear(f2, P["dust3r.decoder_embed.weight"], P["dust3r.decoder_embed.bias"])
    for i in range(cfg.dec_depth):
        na = decoder_block(a, b, pos1, pos2, P, "dust3r.dec_blocks.%d." % i, cfg.dec_heads, cfg.rope_base)
        nb = decoder_block(b, a, pos2, pos1, P, "dust3r.dec_blocks2.%d." % i, cfg.dec_heads, cfg.rope_base)
        a, b = na, nb
        outs1.append(a)
        outs2.append(b)
    outs1[-1] = layer_norm(outs1[-1], P["dust3r.dec_norm.weight"], P["dust3r.dec_norm.bias"], 1e-6)
    outs2[-1] = layer_norm(outs2[-1], P["dust3r.dec_norm.weight"], P["dust3r.dec_norm.bias"], 1e-6)
    return outs1, outs2


def encode_feat_key(feat, dec_last, P, num):
    """spann3r/model.py:299-303"""
    p = "attn_head_%d." % num
    x = torch.cat((feat, dec_last), dim=-1)
    h = linear(x, P[p + "0.weight"], P[p + "0.bias"])
    if PRECISION == "bf16" and ACT_ON_LOAD:
        return linear(h, P[p + "2.weight"], P[p + "2.bias"], act_in=1)
    return linear(_Gelu.apply(h), P[p + "2.weight"], P[p + "2.bias"])


def encode_cur_value(pts3d, feat_k, P, cfg, dec_last=None, pos1=None):
    """spann3r/model.py:305-320 -> cur_v + feat_k (the sum add_mem stores, :519); use_feat: the value encoder runs on dec1[-1]"""
    if cfg.use_feat:
        x, pos = dec_last, pos1
    else:
        x, nh, nw = _patchify(pts3d.permute(0, 3, 1, 2), cfg.patch)
        W = P["pos_patch_embed.proj.weight"]
        x = linear(x, W.reshape(W.shape[0], -1), P["pos_patch_embed.proj.bias"], wkey=(id(W), "patch", W._version))
        pos = _positions(pts3d.shape[0], nh, nw, pts3d.device)
    for i in range(cfg.val_depth):
        x = block(x, pos, P, "value_encoder.%d." % i, cfg.enc_heads, cfg.rope_base, use_rope=cfg.mem_pos_enc)
    x = layer_norm(x, P["value_norm.weight"], P["value_norm.bias"], 1e-6)
    return linear(x, P["value_out.weight"], P["value_out.bias"], res=feat_k)


def _orientation(view, H, W):
    """per-sample landscape flags of a view from its `true_shape` (CPU int tensor [B, 2], dust3r/datasets/base/
    base_stereo_view_dataset.py:89,215-220: the image shape, or its transpose for a portrait the dataset rotated to landscape);
    None: every sample is a landscape / square image in its own orientation"""
    ts = view.get("true_shape")
    if ts is None or H == W:
        return None
    ts = torch.as_tensor(ts).reshape(-1, 2).cpu()
    land = ts[:, 1] >= ts[:, 0]
    ok = ((ts[:, 0] == H) & (ts[:, 1] == W)) | ((ts[:, 0] == W) & (ts[:, 1] == H))
    if not bool(ok.all()):
        raise ValueError("true_shape must be the image shape or its transpose (got %s for %dx%d images)" % (ts.tolist(), H, W))
    return None if bool(land.all()) else land


def head_by_orientation(dec, land, nh, nw, P, cfg, num):
    """downstream_head through the landscape_only wrapper (dust3r/utils/misc.py:66-94): portrait samples (land[b] False) run the
    DPT head on the transposed token grid and their results are transposed back, a mixed batch runs the head once per orientation
    and the results are scattered into batch order (index_put on the tape)."""
    if land is None:
        return dpt_head(dec, nh, nw, P, cfg, num)
    if not bool(land.any()):
        pts, conf = dpt_head(dec, nw, nh, P, cfg, num)
        return pts.transpose(1, 2), conf.transpose(1, 2)
    dev = dec[0].device
    B = dec[0].shape[0]
    out = None
    for mask, (gh, gw), swap in ((land, (nh, nw), False), (~land, (nw, nh), True)):
        idx = mask.nonzero().reshape(-1).to(dev)
        pts, conf = dpt_head([d.index_select(0, idx) for d in dec], gh, gw, P, cfg, num)
        if swap:
            pts, conf = pts.transpose(1, 2), conf.transpose(1, 2)
        if out is None:
            out = [pts.new_zeros((B,) + tuple(pts.shape[1:])), conf.new_zeros((B,) + tuple(conf.shape[1:]))]
        out = [out[0].index_copy(0, idx, pts.contiguous()), out[1].index_copy(0, idx, conf.contiguous())]
    return out[0], out[1]


class TrainMemory:
    """What forward(..., return_memory=True) hands out in train mode: the visible fields of the reference's SpatialMemory after the
    sequence (spann3r/model.py:80-95,180-181) -- mem_k / mem_v are the tape tensors themselves."""
    mem_k = mem_v = mem_attn = mem_count = None
    wm = lm = 0
    num_patches = None


def forward_train(P, frames, cfg, dropout_p=0.0, generator=None, return_memory=False):
    """-> (preds, preds_all[, memory]) of Spann3R.forward in train mode.  P: {reference parameter name: tensor} (leaves of the tape);
    frames: list of dicts with img [B,3,H,W] on the device and, optionally, `true_shape` [B, 2] (portraits the dataset rotated to
    landscape, also mixed with landscape samples in one batch: head_by_orientation); dropout_p: spann3r/model.py:229
    memory_dropout (0.15 in training), drawn from `generator`."""
    ops.set_product_mode("f32_bf16" if PRECISION == "bf16" else "fp32", process_default=True)   # (an inference call in between activated its engine's own on this thread)
    mem_k = mem_v = mem_attn = mem_count = None
    feat2 = pos2 = feat_k2 = None
    preds, preds_all = None, []
    nq, nk, nv = (P["norm_q.weight"], P["norm_q.bias"]), (P["norm_k.weight"], P["norm_k.bias"]), (P["norm_v.weight"], P["norm_v.bias"])
    # the encoder sees every frame once and does not depend on the memory (spann3r/model.py:293-295): all n frames go through it
    # as ONE batch (rows n*B*P per GEMM instead of five passes of B*P rows); the reference encodes frame by frame, same values
    same = all(f["img"].shape == frames[0]["img"].shape for f in frames)
    if same:
        f_all, p_all, grid = encode_image(torch.cat([f["img"] for f in frames], 0), P, cfg)
        feats, poss = f_all.chunk(len(frames), 0), p_all.chunk(len(frames), 0)
    for i in range(len(frames) - 1):
        v1, v2 = frames[i], frames[i + 1]
        if same:
            feat1, pos1, feat2, pos2 = feats[i], poss[i], feats[i + 1], poss[i + 1]
        elif feat2 is None:
            f, p, grid = encode_image(torch.cat((v1["img"], v2["img"]), 0), P, cfg)
            (feat1, feat2), (pos1, pos2) = f.chunk(2, 0), p.chunk(2, 0)
        else:
            feat1, pos1 = feat2, pos2
            feat2, pos2, grid = encode_image(v2["img"], P, cfg)
        if feat_k2 is not None:
            mask = None
            if dropout_p > 0:
                keep = torch.rand(feat_k2.shape[0], feat_k2.shape[1], mem_k.shape[1], device=feat_k2.device, generator=generator) >= dropout_p
                mask = keep.float() / (1.0 - dropout_p)
            feat_fuse = memory_read_train(feat_k2, mem_k, mem_v, nq, nk, nv, mask, attn_accum=mem_attn if return_memory else None)
        else:
            feat_fuse = feat1
        dec1, dec2 = decoder(feat_fuse, pos1, feat2, pos2, P, cfg)
        feat_k1 = encode_feat_key(feat1, dec1[-1], P, 1)
        feat_k2 = encode_feat_key(feat2, dec2[-1], P, 2)
        Hi, Wi = v1["img"].shape[-2:]
        pts1, conf1 = head_by_orientation(dec1, _orientation(v1, Hi, Wi), grid[0], grid[1], P, cfg, 1)
        pts2, conf2 = head_by_orientation(dec2, _orientation(v2, Hi, Wi), grid[0], grid[1], P, cfg, 2)
        v = encode_cur_value(pts1, feat_k1, P, cfg, dec1[-1], pos1)
        mem_k = feat_k1 if mem_k is None else torch.cat((mem_k, feat_k1), 1)
        mem_v = v if mem_v is None else torch.cat((mem_v, v), 1)
        if return_memory:                                             # add_mem's bookkeeping (spann3r/model.py:84-90)
            z = torch.zeros(feat_k1.shape[0], feat_k1.shape[1], device=feat_k1.device)
            mem_count = z if mem_count is None else torch.cat((mem_count + 1, z), 1)
            mem_attn = z.clone() if mem_attn is None else torch.cat((mem_attn, z), 1).contiguous()
        res2 = {"pts3d_in_other_view": pts2, "conf": conf2}
        if preds is None:
            res1 = {"pts3d": pts1, "conf": conf1}
            preds = [res1]
        else:
            res1 = {"pts3d_in_other_view": pts1, "conf": conf1}
            preds.append(res1)
        preds_all.append((res1, res2))
    preds.append(res2)
    if return_memory:
        mem = TrainMemory()
        mem.mem_k, mem.mem_v, mem.mem_attn, mem.mem_count = mem_k, mem_v, mem_attn[..., None], mem_count[..., None]
        mem.num_patches = mem_k.shape[1] // (len(frames) - 1)
        return preds, preds_all, mem
    return preds, preds_all


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW(params, lr, betas, eps, weight_decay) with the update in a HIP kernel (sp3_adamw), the optimizer of
    spann3r/training.py:327.  `step(grad_scale=...)` multiplies every gradient first (gradient clipping / accumulation)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        lib = L.load()
        for gr in self.param_groups:
            b1, b2 = gr["betas"]
            for p in gr["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("AdamW (HIP): fp32 contiguous parameters on the GPU")
                st = self.state[p]
                if not st:
                    st["step"], st["m"], st["v"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                g = p.grad.contiguous()
                L.check(lib.sp3_adamw(p.data_ptr(), g.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(), p.numel(), float(gr["lr"]), b1, b2,
                                      float(gr["eps"]), float(gr["weight_decay"]), st["step"], float(grad_scale), L.stream_ptr()), "sp3_adamw")
        invalidate_weight_cache()       # sp3_adamw writes through data_ptr(): version counters do not move


def parameter_groups(model, weight_decay):
    """The two groups croco/utils/misc.py:404-455 get_parameter_groups builds at layer_decay = 1 (spann3r/training.py:327): no weight
    decay for 1-D parameters and biases, `weight_decay` for the rest."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.dim() == 1 or name.endswith(".bias")) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0, "lr_scale": 1.0}, {"params": decay, "weight_decay": weight_decay, "lr_scale": 1.0}]


def scheduled_lr(epoch, args):
    """learning rate at a (fractional) epoch, croco/utils/misc.py:464-471: linear warm-up over args.warmup_epochs, then a half cosine
    from args.lr down to args.min_lr at args.epochs"""
    if epoch < args.warmup_epochs:
        return args.lr * epoch / args.warmup_epochs
    return args.min_lr + (args.lr - args.min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)))


class TrainStep:
    """One optimisation step of spann3r/training.py:216-231 on one rank: train-mode forward (HIP autograd ops), ConfLoss_t,
    backward with the bucket all-reduces launched from inside it (RCCL when a process group is up), global-norm clip and AdamW on
    the flat buckets.  `run(frames, gts)` returns (loss, gradient norm) as device scalars (no host sync).

    graph=True: the whole step -- ~16 000 kernel launches at batch 4, the bucket all-reduces of a multi-rank job included -- is
    captured once into a hipGraph and replayed; the batch is copied into static buffers, the optimizer's step count and learning rate live on the device
    (`set_lr`).  Shapes must not change between steps (training batches of one resolution)."""

    def __init__(self, model, lr=5e-5, weight_decay=0.05, betas=(0.9, 0.95), clip_grad=1.0, precision="bf16", bucket_mb=64.0, force_collectives=False,
                 graph=False, accum_iter=1):
        from .loss import ConfLoss_t, Regr3D_t, L21
        from .runner import GradReducer
        set_precision(precision)
        self.model = model.train()
        self.reducer = GradReducer(list(model.parameters()), bucket_mb=bucket_mb, overlap=True, force=force_collectives)
        self.opt = FlatAdamW(parameter_groups(model, weight_decay), self.reducer, lr=lr, betas=betas, weight_decay=weight_decay)
        self.crit = ConfLoss_t(Regr3D_t(L21, norm_mode="avg_dis", fix_first=False), alpha=0.4)      # training.py:37
        self.clip_grad = clip_grad
        # multi-rank too: the bucket all-reduces (RCCL through torch.distributed's NCCL backend is capturable) are part of the
        # captured step; SP3_TRAIN_GRAPH=0 forces the eager step
        import os
        self.graph = bool(graph)
        self._g = self._static = self._out = None
        # gradient accumulation (spann3r/training.py:228-233): run() is ONE iteration of the loop; the gradients of accum_iter
        # consecutive iterations (each loss divided by accum_iter) add up in the flat buckets and the last one reduces them over the
        # ranks, clips and updates.  (DDP all-reduces after every backward; the mean over ranks is linear, so one reduction of the
        # accumulated buckets gives the same gradient with accum_iter times fewer collectives.)
        self.accum_iter = int(accum_iter)
        assert self.accum_iter >= 1
        self.data_iter_step = 0                      # iterations since the last reset_iteration(): position inside the window
        self._graphs = {}                            # (first, last) of the window -> (hipGraph, outputs, orientation pattern)

    def set_lr(self, lr):
        for g in self.opt.param_groups:
            g["lr"] = lr * g.get("lr_scale", 1.0)

    def adjust_learning_rate(self, epoch, args):
        """croco/utils/misc.py:464-479 (the per-iteration hook of spann3r/training.py:205-207): linear warm-up over
        args.warmup_epochs, then half-cycle cosine from args.lr to args.min_lr at args.epochs; every group gets lr * its lr_scale.
        `epoch` is fractional.  In graph mode the value reaches the captured update kernels through device memory (sync_lr)."""
        lr = scheduled_lr(epoch, args)
        self.set_lr(lr)
        return lr

    def reset_iteration(self):
        """start of an epoch (training.py:200 `optimizer.zero_grad()`): an unfinished accumulation window is dropped"""
        self.data_iter_step = 0

    def _window(self):
        i = self.data_iter_step % self.accum_iter
        return i == 0, i == self.accum_iter - 1

    def _body(self, frames, gts, monitor, first=True, last=True):
        if first:
            self.reducer.zero_grad()
        self.reducer.prepare(arm=last)               # (also resets the direct-to-bucket bookkeeping, train.reset_pending)
        preds, preds_all = self.model(frames)
        loss, details, factor = self.crit.compute_frame_loss(gts, preds_all, monitor=monitor)
        total = loss + factor
        (total if self.accum_iter == 1 else total * (1.0 / self.accum_iter)).backward()          # training.py:228 `loss /= accum_iter`
        if not last:
            return total.detach(), None
        self.reducer.finish()
        # parameters without a gradient in the whole window keep .grad None in the reference and AdamW passes over them (no weight
        # decay either): across ranks the used-set exchange of the reducer, on one rank its own bookkeeping
        skip = self.reducer.unused_everywhere() if self.reducer.active() else self.reducer.untouched()
        norm = self.opt.step(max_norm=self.clip_grad, skip=skip)
        return total.detach(), norm

    @staticmethod
    def _pattern(frames):
        """orientation pattern of a batch (per view: which samples are rotated portraits): host-side control flow of the forward"""
        pat = []
        for f in frames:
            H, W = f["img"].shape[-2:]
            land = _orientation(f, H, W)
            pat.append(None if land is None else tuple(bool(x) for x in land))
        return tuple(pat)

    def run(self, frames, gts):
        """one iteration of the training loop (spann3r/training.py:216-233): returns (loss + factor, gradient norm) as device
        scalars; the norm is None on iterations that only accumulate (`update_grad=False`, croco/utils/misc.py:268-282)."""
        first, last = self._window()
        self.data_iter_step += 1
        if not self.graph:
            return self._body(frames, gts, monitor=False, first=first, last=last)
        pat = self._pattern(frames)
        key = (first, last)
        held = self._graphs.get(key)
        if held is not None and pat != held[2]:
            # a captured step holds ONE orientation pattern (the per-orientation head passes are host-side control flow): a batch
            # with another pattern takes the eager step on the same buffers / optimizer state
            self.opt.sync_lr()
            out = self._body(frames, gts, monitor=False, first=first, last=last)
            invalidate_weight_cache()
            return out
        if held is None:
            clone = lambda d: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}
            if self._static is None:
                self._static = ([clone(f) for f in frames], [clone(g) for g in gts])
            else:
                self._copy_in(frames, gts)
            # the warm-up steps (allocator pools, caches, chunk tables) and the capture must not move the trajectory: parameters,
            # both moments, the step count and (inside an accumulation window) the gradients gathered so far are put back, so the
            # first REPLAY is this iteration, as in eager mode
            snap = self.opt.snapshot()
            grads = None if first else [g.clone() for g, _ in self.reducer.flat_buffers()]
            touched = list(self.reducer._touched)
            invalidate_weight_cache()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._body(*self._static, monitor=False, first=first, last=last)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.opt.capture_mode(True)
            self.reducer.capture = True                  # collectives stay in the graph; the used-parameter exchange is host-free
            g = torch.cuda.CUDAGraph()
            # no cyclic garbage collection while the stream captures: the collector may finalise hipGraphs of models dropped
            # earlier (an inference runner <-> model cycle), and destroying a graph is not permitted during a capture
            import gc
            gc.collect()
            gc_was = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(g):
                    out = self._body(*self._static, monitor=False, first=first, last=last)
            finally:
                if gc_was:
                    gc.enable()
            self._graphs[key] = held = (g, out, pat)
            self._g, self._out, self._captured_pattern = g, out, pat
            self.opt.restore(snap)
            if grads is not None:
                for (dst, _), src in zip(self.reducer.flat_buffers(), grads):
                    dst.copy_(src)
                self.reducer._touched = touched
            del snap, grads
        else:
            self._copy_in(frames, gts)
        self.opt.sync_lr()
        held[0].replay()
        if last:
            self.opt.step_count += 1                 # the device-side count advanced inside the replay
        invalidate_weight_cache()                    # the replay rewrote the packed copies (and, on `last`, the parameters)
        return held[1]

    def _copy_in(self, frames, gts):
        for dst, src in zip(self._static[0] + self._static[1], list(frames) + list(gts)):
            for k, v in src.items():
                if torch.is_tensor(v):
                    dst[k].copy_(v, non_blocking=True)


def train_one_epoch(step, data_loader, epoch, args, on_iteration=None):
    """The iteration schedule of the reference's train_one_epoch (spann3r/training.py:168-262) around TrainStep.run: per-iteration
    learning rate at the first iteration of every accumulation window (:205-207, fractional epoch = epoch + i / len(loader)), the
    update on the last one (:229-233), a non-finite loss stops the run (:223-226).  `data_loader` yields batches (lists of view
    dicts on the device or the host; `img`, `pts3d`, `valid_mask`, `camera_pose` are moved, :209-213); args: accum_iter (must equal
    step.accum_iter), lr, min_lr, warmup_epochs, epochs.  Returns the averaged statistics {loss, lr, norm}; `on_iteration(i, loss,
    norm, lr)` sees every iteration.  Data-set curriculum (`set_ratio`), logging and checkpoints are the caller's (out of scope)."""
    assert int(getattr(args, "accum_iter", 1)) == step.accum_iter, "args.accum_iter and TrainStep(accum_iter=...) differ"
    step.model.train()
    step.reset_iteration()
    dev = next(step.model.parameters()).device
    n = len(data_loader)
    losses, norms, lr = [], [], None
    for i, batch in enumerate(data_loader):
        if i % step.accum_iter == 0:
            lr = step.adjust_learning_rate(epoch + i / n, args)
        batch = [{k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) and k in ("img", "pts3d", "valid_mask", "camera_pose") else v)
                  for k, v in view.items()} for view in batch]
        loss, norm = step.run(batch, batch)
        loss_value = float(loss)
        if not math.isfinite(loss_value):
            raise FloatingPointError("Loss is %r, stopping training" % loss_value)
        losses.append(loss_value)
        if norm is not None:
            norms.append(float(norm))
        if on_iteration is not None:
            on_iteration(i, loss_value, None if norm is None else norms[-1], step.opt.param_groups[0]["lr"])
    mean = lambda v: sum(v) / len(v) if v else float("nan")
    return {"loss": mean(losses), "norm": mean(norms), "lr": lr}


class FlatAdamW:
    """AdamW + global gradient-norm clipping on the flat bucket layout of `runner.GradReducer` -- the optimizer side of
    spann3r/training.py:216-231 (`loss_scaler(loss, optimizer, clip_grad=1.0, parameters=model.parameters())`,
    croco/utils/misc.py:262-288; AdamW(param_groups, lr, betas=(0.9, 0.95)) of :327) in a handful of launches per step:

      * every parameter becomes a view of a flat fp32 buffer with the element layout of its gradient bucket (one buffer per
        bucket; moments m, v likewise), so the update of a bucket is ONE sp3_adamw_flat launch; weight decay and the
        group's lr scale are looked up per 1024-element chunk (parameters start on chunk boundaries);
      * `step(max_norm=...)` first reduces sum(g^2) of every bucket in a fixed order (sp3_sumsq_partial, sp3_clip_coef) and
        leaves clip_grad_norm_'s coefficient on the device; the update kernels read it from there (no host sync; the norm is
        returned as a device scalar like the reference's `norm`);
      * parameters that received no gradient on any rank (`reducer.unused_everywhere()`; torch leaves their .grad None and
        AdamW skips them) are skipped through the chunk table.

    groups: torch-style parameter groups ([{"params": [...], "weight_decay": w, "lr_scale": s}, ...], as
    croco/utils/misc.py get_parameter_groups builds them) or a plain parameter list.  `param_groups[i]["lr"]` may be changed
    between steps (misc.adjust_learning_rate)."""

    def __init__(self, groups, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        groups = list(groups)
        if groups and not isinstance(groups[0], dict):
            groups = [{"params": groups}]
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g["params"] = [p for p in g["params"] if p.requires_grad]
            g.setdefault("lr", lr * g.get("lr_scale", 1.0))
            g.setdefault("weight_decay", weight_decay)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            self.param_groups.append(g)
        self.reducer = reducer
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self._group_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        missing = [p for p in reducer.params if id(p) not in self._group_of]
        if missing or len(self._group_of) != len(reducer.params):
            raise ValueError("FlatAdamW: the parameter groups and the GradReducer must hold the same parameters")
        self.flat_p, self.flat_m, self.flat_v, self._tables, self._table_key = [], [], [], [], None
        with torch.no_grad():
            for gbuf, items in reducer.flat_buffers():
                fp = torch.zeros_like(gbuf)
                for p, o in items:
                    if not (p.is_cuda and p.dtype == torch.float32):
                        raise RuntimeError("FlatAdamW: fp32 parameters on the GPU")
                    v = fp[o:o + p.numel()].view_as(p)
                    v.copy_(p.data)
                    p.data = v                      # the parameter now lives in the flat buffer (state_dict etc. unchanged)
                self.flat_p.append(fp)
                self.flat_m.append(torch.zeros_like(gbuf))
                self.flat_v.append(torch.zeros_like(gbuf))
        nblk = [int(L.load().sp3_sumsq_blocks(g.numel())) for g, _ in reducer.flat_buffers()]
        self._nblk = nblk
        dev = self.flat_p[0].device
        self._partials = torch.zeros(sum(nblk), dtype=torch.float64, device=dev)
        self._coef = torch.ones(2, dtype=torch.float32, device=dev)
        self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)      # step count on the device (incremented by sp3_clip_coef)
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._capture = False
        for p in reducer.params:
            p._sp3_direct_grad = True               # _Linear.backward may accumulate dW straight into the bucket view
        invalidate_weight_cache()

    def capture_mode(self, on):
        """on: step count and learning rate are read from device memory by the update kernels (a captured step replays them)"""
        self._capture = bool(on)
        self._step_dev.fill_(self.step_count)

    def _base_lr(self):
        """the learning rate every group's lr is a multiple of: group 0's lr over its own scale"""
        g0 = self.param_groups[0]
        return float(g0["lr"]) / float(g0.get("lr_scale", 1.0) or 1.0)

    def _lr_ratio(self, g, base):
        """lr of group g as a multiple of the base lr: its lr_scale when the group carries one (croco/utils/misc.py:404-455 groups
        do; valid at lr = 0, i.e. at the first iteration of a warm-up), else lr / base"""
        if "lr_scale" in g:
            return float(g["lr_scale"])
        return float(g["lr"]) / base if base else 1.0

    def sync_lr(self):
        self._lr_dev.fill_(self._base_lr())

    def snapshot(self):
        """parameters, moments and step count (TrainStep restores them behind its warm-up / capture steps)"""
        return ([t.clone() for t in self.flat_p], [t.clone() for t in self.flat_m], [t.clone() for t in self.flat_v], self.step_count)

    @torch.no_grad()
    def restore(self, snap):
        ps, ms, vs, n = snap
        for dst, src in zip(self.flat_p + self.flat_m + self.flat_v, ps + ms + vs):
            dst.copy_(src)
        self.step_count = n
        self._step_dev.fill_(n)
        invalidate_weight_cache()

    def zero_grad(self, set_to_none=False):
        self.reducer.zero_grad()

    def _chunk_tables(self, skip):
        """(weight decay, lr / lr of group 0; -1 = skip) per 1024-element chunk and bucket; rebuilt only when the groups' lr ratios,
        weight decays or the skipped set change"""
        base = self._base_lr()
        key = (tuple((g["weight_decay"], self._lr_ratio(g, base)) for g in self.param_groups), tuple(sorted(id(p) for p in skip)))
        if key == self._table_key:
            return self._tables
        skip_ids = {id(p) for p in skip}
        tabs = []
        for gbuf, items in self.reducer.flat_buffers():
            t = torch.zeros(gbuf.numel() // 1024, 2)
            t[:, 1] = -1.0                          # alignment gaps between parameters: nothing to update
            for p, o in items:
                g = self.param_groups[self._group_of[id(p)]]
                c0, c1 = o // 1024, (o + p.numel() + 1023) // 1024
                t[c0:c1, 0] = g["weight_decay"]
                t[c0:c1, 1] = -1.0 if id(p) in skip_ids else self._lr_ratio(g, base)
            tabs.append(t.to(gbuf.device))
        if self._capture and self._tables:
            # a captured step holds the tables' addresses: new scales / weight decays / skips are written INTO them
            for old_t, new_t in zip(self._tables, tabs):
                old_t.copy_(new_t)
            tabs = self._tables
        self._tables, self._table_key = tabs, key
        return tabs

    @torch.no_grad()
    def step(self, grad_scale=1.0, max_norm=None, skip=()):
        """One AdamW update of every bucket.  The gradients are multiplied by grad_scale (1 / accum_iter, 1 / world ...) and, when
        max_norm is given, by clip_grad_norm_'s coefficient min(1, max_norm / (||grad_scale * g|| + 1e-6)).  Returns the gradient norm
        (device scalar) when max_norm is given."""
        lib = L.load()
        self.step_count += 1
        bufs = self.reducer.flat_buffers()
        coef_ptr, gs = None, float(grad_scale)
        if max_norm is not None or self._capture:
            off = 0
            for (g, _), nb in zip(bufs, self._nblk):
                L.check(lib.sp3_sumsq_partial(g.data_ptr(), g.numel(), self._partials[off:].data_ptr(), L.stream_ptr()), "sp3_sumsq_partial")
                off += nb
            L.check(lib.sp3_clip_coef(self._partials.data_ptr(), off, float(max_norm if max_norm is not None else 0.0), float(grad_scale),
                                      self._coef.data_ptr(), self._step_dev.data_ptr(), L.stream_ptr()), "sp3_clip_coef")
            coef_ptr, gs = self._coef.data_ptr(), 1.0
        tabs = self._chunk_tables(skip)
        base = self._base_lr()
        b1, b2 = self.betas
        for (g, _), fp, fm, fv, tab in zip(bufs, self.flat_p, self.flat_m, self.flat_v, tabs):
            L.check(lib.sp3_adamw_flat(fp.data_ptr(), g.data_ptr(), fm.data_ptr(), fv.data_ptr(), g.numel(), tab.data_ptr(), float(base), b1, b2,
                                       float(self.eps), self.step_count, coef_ptr, gs, self._step_dev.data_ptr() if self._capture else None,
                                       self._lr_dev.data_ptr() if self._capture else None, L.stream_ptr()), "sp3_adamw_flat")
            if not self._capture:
                fp[:0].zero_()                      # bumps the version counter the parameter views share: weight caches notice
        invalidate_weight_cache()
        return self._coef[1] if max_norm is not None else None
