"""Training-side pieces of the hot path: the spatial-memory read as a differentiable op whose forward
AND backward run in HIP kernels -- `SpatialMemory.memory_read` of the reference in train mode (spann3r/model.py:145-183
with attn_thresh = 0 and nn.Dropout on the attention, :475):

    out = dropout(softmax(LN_q(feat) . LN_k(mem_k)^T / sqrt(C))) . LN_v(mem_v) + feat

`memory_read_train` returns `out` with an autograd edge to feat, mem_k, mem_v and the six LayerNorm parameters; the
backward is four sp3_gemm launches (fp32 MFMA) with the transposes, the softmax / dropout backward and the LayerNorm
backward of csrc/train.hip in between.  The dropout mask is an input (0 or 1/(1-p), drawn by the caller: the reference's
comes from torch's RNG stream and cannot be matched bit for bit by any other implementation).  Gradient averaging across
ranks: `spann3r_amd.runner.GradReducer`; the criterion: `spann3r_amd.loss`.  What is NOT here: backward passes of the ViT
encoder / decoder / DPT heads (the model's forward refuses train mode with active dropout for that reason)."""
import torch

from . import lib as L
from . import ops


def _r8(n):
    return (n + 7) // 8 * 8


def _transpose(src, rows, cols, ld_src, ld_dst=None):
    ld_dst = _r8(rows) if ld_dst is None else ld_dst
    dst = torch.zeros(cols, ld_dst, device=src.device)
    L.check(L.load().sp3_transpose(src.data_ptr(), ld_src, dst.data_ptr(), ld_dst, rows, cols, L.stream_ptr()), "sp3_transpose")
    return dst


def _ln_bwd(x, gamma, dy, dx_add, eps):
    rows, C = x.shape
    dx = torch.empty_like(x)
    dg, db = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
    scratch = torch.empty((rows + 3) // 4 * 2 * C, device=x.device)
    L.check(L.load().sp3_layernorm_bwd(x.data_ptr(), C, gamma.data_ptr(), dy.data_ptr(), dy.stride(0), L.ptr(dx_add), C, dx.data_ptr(), C,
                                       dg.data_ptr(), db.data_ptr(), 0, scratch.data_ptr(), rows, C, eps, L.stream_ptr()), "sp3_layernorm_bwd")
    return dx, dg, db


class _MemoryReadTrain(torch.autograd.Function):
    """one batch element: feat [P,C], mem_k / mem_v [T,C], mask [P,T] or None"""

    @staticmethod
    def forward(ctx, feat, mem_k, mem_v, gq, bq, gk, bk, gv, bv, mask, eps):
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
        dAd = torch.zeros(P, Tp, device=dev)
        ops.gemm(dO, vh, dAd, M=P, N=T, K=C, lda=C, ldc=Tp)
        # dV_hat = Ad^T . dO  ->  A = Ad^T [T, Pp], W[N = C, K = Pp] = dO^T
        AdT = _transpose(Ad, P, T, Tp)                                  # [T, Pp] (pad columns zero)
        dOT = _transpose(dO, P, C, C)                                   # [C, Pp]
        dVh = f32(T, C)
        ops.gemm(AdT, dOT, dVh, M=T, N=C, K=Pp, lda=Pp, ldc=C)
        # softmax (+ dropout) backward, scaled by 1 / sqrt(C)
        dS = torch.zeros(P, Tp, device=dev)
        L.check(L.load().sp3_softmax_bwd(A.data_ptr(), dAd.data_ptr(), L.ptr(mask), dS.data_ptr(), Tp, P, T, ctx.alpha, L.stream_ptr()),
                "sp3_softmax_bwd")
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
        return dfeat, dk, dv, dgq, dbq, dgk, dbk, dgv, dbv, None, None


def memory_read_train(feat, mem_k, mem_v, norm_q, norm_k, norm_v, mask=None, eps=1e-5):
    """feat [B,P,C], mem_k / mem_v [B,T,C] (fp32, device); norm_* = (weight, bias) of the three LayerNorms; mask [B,P,T]
    float32 dropout mask (0 or 1/(1-p)) or None.  -> out [B,P,C] = memory_read(feat, res=True) of the reference in train mode."""
    if not feat.is_cuda:
        raise RuntimeError("memory_read_train runs on the GPU (HIP kernels); there is no CPU path")
    outs = []
    for b in range(feat.shape[0]):
        outs.append(_MemoryReadTrain.apply(feat[b].contiguous().float(), mem_k[b].contiguous().float(), mem_v[b].contiguous().float(),
                                           norm_q[0], norm_q[1], norm_k[0], norm_k[1], norm_v[0], norm_v[1],
                                           None if mask is None else mask[b].contiguous(), eps))
    return torch.stack(outs)
