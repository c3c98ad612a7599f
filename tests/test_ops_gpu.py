"""Kernel-level parity tests (need an MI355X): every C-ABI op against a CPU fp32/fp64 restatement
(oracle/spann3r_oracle.py where the op exists there, plain torch-CPU formulas otherwise).
Tolerances: fp32 MFMA path 2e-5 relative (summation order only); bf16 path is compared against
the same formula evaluated on bf16-rounded operands with fp32 accumulation."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from spann3r_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def bf(t):
    return t.to(torch.bfloat16).float()


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-3}


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(196, 1024, 1024), (392, 3072, 1024), (20, 768, 768), (100, 96, 1024),
                                   (196, 588, 1024), (49, 256, 96), (196, 1024, 1736), (7, 32, 8), (300, 160, 200)])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3])
@pytest.mark.parametrize("packed", [0, 1, 2])
def test_gemm_plain(wdt, M, N, K, tile, packed):
    ops = _ops()
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2)
    # asymmetric content so a transposed write cannot pass
    A[:, 0] += torch.arange(M) * 0.01
    W[:, 1] += torch.arange(N) * 0.02
    Ad, Wd = A.to(DEV), W.to(DEV).to(wdt)
    if packed:
        Wd = ops.PackedWeight(Wd)
    if packed == 2:        # A in fragment order too (as the producing kernels write it); bf16 A in bf16 mode
        Ad = ops.PackedAct.from_dense(Ad.to(wdt))
        assert torch.equal(Ad.to_dense(), A.to(DEV).to(wdt))
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(Ad, Wd, out, M=M, N=N, K=K, lda=K, ldc=N, tile=tile)
    ref = (bf(A) if wdt == torch.bfloat16 else A).double() @ (bf(W) if wdt == torch.bfloat16 else W).double().T
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu(), ref) < TOL[wdt], (M, N, K, tile)


LDS_TILES = [5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 20, 21, 22, 23]    # 13-15: bf16 only, at most 64 k-blocks per K slice; 20-23: pipelined


@pytest.mark.parametrize("wdt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(1960, 3072, 1024), (300, 256, 192), (129, 384, 64), (1100, 1024, 4096), (196, 768, 768),
                                   (196, 2304, 768), (196, 1024, 4096), (98, 96, 128)])
def test_gemm_lds_staged_tile(M, N, K, wdt):
    """tiles 5-12: both operands staged through LDS by global_load_lds (fragment-order A and W in the MFMA dtype): the
    128-row tiles of the whole-sequence encoder and the 112- / 208-row weight-streaming tiles (2-4 ring slots, pieces dealt
    unevenly over the waves, K halves over wave pairs); ragged M / N tiles, K down to one k-block; epilogues: bias + GELU
    -> packed, residual + row statistics + packed copy, folded LayerNorm, split-K partials."""
    ops = _ops()
    KB = 64 if wdt == torch.bfloat16 else 32
    if K % KB:
        pytest.skip("whole k-blocks only")
    if wdt == torch.float32 and M * N * K > 2e9:
        pytest.skip("fp32 case kept small")
    cast = bf if wdt == torch.bfloat16 else (lambda t: t)
    tol = TOL[wdt]
    A, W, b, r1 = rnd(M, K, seed=1), rnd(N, K, seed=2) * 0.05, rnd(N, seed=3), rnd(M, N, seed=4)
    A[:, 0] += torch.arange(M) * 0.01
    W[:, 1] += torch.arange(N) * 0.02
    Ap = ops.PackedAct.from_dense(A.to(DEV).to(wdt))
    Wp = ops.PackedWeight(W.to(DEV).to(wdt))
    ref = cast(A).double() @ cast(W).double().T
    out1 = torch.empty(M, N, device=DEV)
    ops.gemm(Ap, Wp, out1, M=M, N=N, K=K, lda=K, ldc=N, tile=1)          # a register-ring tile
    tiles = [t for t in LDS_TILES if t < 13 or t >= 20 or wdt == torch.bfloat16]
    for tile in tiles:
        out = torch.full((M, N), float("nan"), device=DEV)
        ops.gemm(Ap, Wp, out, M=M, N=N, K=K, lda=K, ldc=N, tile=tile)
        assert rel_err(out.cpu(), ref) < tol, tile
        # the same launch must agree with the register-ring tiles to fp32 summation order
        assert rel_err(out.cpu(), out1.cpu()) < 1e-5, tile
    nkb = K // KB
    for tile, S in ((7, 2), (8, 3), (11, 4), (13, 2), (14, 3), (20, 2), (21, 3), (22, 5), (23, 4)):
        if nkb >= S and tile in tiles:                                                     # split-K partials of the streaming tiles
            part = torch.full((S, M, N), float("nan"), device=DEV)
            ops.gemm(Ap, Wp, part, M=M, N=N, K=K, lda=K, ldc=N, splitk=S, tile=tile)
            assert rel_err(part.sum(0).cpu(), ref) < tol, (tile, S)
    if N % 32 == 0:
        for tile in (5, 7, 8, 11, 13, 14, 20, 21, 22, 23):
            if tile not in tiles:
                continue
            # producer epilogue: bias + residual, row statistics, packed copy
            st = torch.zeros(M, N // 32, 2, device=DEV)
            c2 = ops.PackedAct(M, N, wdt, DEV)
            x = torch.empty(M, N, device=DEV)
            ops.gemm(Ap, Wp, x, M=M, N=N, K=K, lda=K, ldc=N, bias=b.to(DEV), res1=r1.to(DEV), ldr1=N, stats_out=st, c2=c2, tile=tile)
            xr = ref + b.double() + r1.double()
            assert rel_err(x.cpu(), xr) < tol
            assert rel_err(st[..., 0].sum(1).cpu(), x.cpu().double().sum(1)) < 1e-5
            assert torch.equal(c2.to_dense(), x.to(wdt))
            # consumer epilogue: LayerNorm of x folded into the next GEMM + GELU -> packed
            g, beta = rnd(N, seed=5) + 1, rnd(N, seed=6)
            W2 = rnd(256, N, seed=7) * 0.05
            Wf = (W2 * g[None]).to(wdt)
            s_n = Wf.float().sum(1).to(DEV)
            b2 = (rnd(256, seed=8) + W2 @ beta).to(DEV)
            h = ops.PackedAct(M, 256, wdt, DEV)
            if N % KB == 0 and (tile < 13 or tile >= 20 or N <= 1024):          # the role tiles fold a LayerNorm of at most 1024 columns
                ops.gemm(c2, ops.PackedWeight(Wf.to(DEV)), h, M=M, N=256, K=N, lda=N, ldc=256, bias=b2, act=ops.ACT_GELU,
                         ln=ops.LnFold(st, N, s_n, 1e-6), tile=tile)
                xf = x.cpu().double()
                mu, var = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
                y = ((cast(x.cpu()).double() @ Wf.double().T) - mu * Wf.double().sum(1)[None]) / torch.sqrt(var + 1e-6) + b2.cpu().double()
                assert rel_err(h.to_dense().float().cpu(), F.gelu(y)) < (8e-3 if wdt == torch.bfloat16 else 1e-4), tile


@pytest.mark.parametrize("tile", [20, 21, 22, 23])
def test_gemm_pipelined_tiles_long_k_and_groups(tile):
    """the pipelined LDS tiles on what the model throws at them: a long contraction with an odd number of k-blocks (ring
    wrap-around, both unroll parities), a grouped launch (two problems, per-problem bias) and the automatic tile choice;
    results must equal the register-ring tile's to summation order and be identical from run to run"""
    ops = _ops()
    wdt = torch.bfloat16
    for (M, N, K, G) in ((1100, 320, 64 * 37, 1), (520, 768, 64 * 12, 2), (1960, 1024, 4096, 1)):
        As = [rnd(M, K, seed=11 + g) for g in range(G)]
        Ws = [rnd(N, K, seed=21 + g) * 0.05 for g in range(G)]
        bias = rnd(G, N, seed=5).to(DEV)
        if G == 1:
            A = ops.PackedAct.from_dense(As[0].to(DEV).to(wdt))
            W = ops.PackedWeight(Ws[0].to(DEV).to(wdt))
            kw = dict(M=M, N=N, K=K, lda=K, ldc=N, bias=bias[0])
        else:
            A = ops.PackedAct.group(G, M, K, wdt, DEV)
            for g in range(G):
                pd = ops.PackedAct.from_dense(As[g].to(DEV).to(wdt)).data.view(-1)
                assert pd.numel() == A.stride
                A.at(g).data[:pd.numel()].copy_(pd)
            W = ops.PackedWeightGroup([ops.PackedWeight(w.to(DEV).to(wdt)) for w in Ws])
            kw = dict(M=M, N=N, K=K, lda=K, ldc=N, bias=bias, batch=G, strideA=A.stride, strideW=W.stride, strideC=M * N, sb={"bias": N * 4})
        outs = {}
        for t in (1, tile, tile, -1):
            out = torch.full((G, M, N), float("nan"), device=DEV)
            ops.gemm(A, W, out, tile=t, **kw)
            outs.setdefault(t, []).append(out.cpu())
        for g in range(G):
            ref = bf(As[g]).double() @ bf(Ws[g]).double().T + bias[g].cpu().double()
            assert rel_err(outs[tile][0][g], ref) < 2e-3, (tile, M, N, K, g)
            assert rel_err(outs[tile][0][g], outs[1][0][g]) < 1e-5
            assert rel_err(outs[-1][0][g], outs[1][0][g]) < 1e-5
        assert torch.equal(outs[tile][0], outs[tile][1])


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
def test_gemm_epilogue_bias_gelu_residuals(wdt):
    ops = _ops()
    M, N, K = 196, 1024, 1024
    A, W, b, r1, r2 = rnd(M, K, seed=1), rnd(N, K, seed=2) * 0.05, rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5)
    Ad, Wd = A.to(DEV), W.to(DEV).to(wdt)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(Ad, Wd, out, M=M, N=N, K=K, lda=K, ldc=N, bias=b.to(DEV), res1=r1.to(DEV), ldr1=N, res2=r2.to(DEV), ldr2=N,
             act=ops.ACT_GELU, alpha=0.5)
    Ar, Wr = (bf(A), bf(W)) if wdt == torch.bfloat16 else (A, W)
    ref = F.gelu(0.5 * (Ar.double() @ Wr.double().T) + b.double()) + r1.double() + r2.double()
    assert rel_err(out.cpu(), ref) < TOL[wdt]
    # in-place residual (C aliases res1), ReLU, bf16 output
    x = r1.clone().to(DEV)
    ops.gemm(Ad, Wd, x, M=M, N=N, K=K, lda=K, ldc=N, bias=b.to(DEV), res1=x, ldr1=N)
    assert rel_err(x.cpu(), (Ar.double() @ Wr.double().T) + b.double() + r1.double()) < TOL[wdt]
    ob = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(Ad, Wd, ob, M=M, N=N, K=K, lda=K, ldc=N, act=ops.ACT_RELU)
    assert rel_err(ob.float().cpu(), F.relu(Ar.double() @ Wr.double().T)) < 5e-3


@pytest.mark.parametrize("M,N,K", [(196, 1024, 1024), (196, 768, 3072), (392, 1024, 4096), (20, 768, 776)])
@pytest.mark.parametrize("tile", [-1, 0, 3])
def test_gemm_bf16_activations(M, N, K, tile):
    """bf16 mode keeps GEMM inputs in bf16 (a_bf16): same numerics as converting fp32 on load."""
    ops = _ops()
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2) * 0.05, rnd(N, seed=3)
    Ab = A.to(torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(Ab.to(DEV), W.to(DEV).to(torch.bfloat16), out, M=M, N=N, K=K, lda=K, ldc=N, bias=b.to(DEV), act=ops.ACT_GELU, tile=tile)
    ref = F.gelu(Ab.double() @ bf(W).double().T + b.double())
    assert rel_err(out.float().cpu(), ref) < 6e-3
    out32 = torch.empty(M, N, device=DEV)
    ops.gemm(Ab.to(DEV), W.to(DEV).to(torch.bfloat16), out32, M=M, N=N, K=K, lda=K, ldc=N, tile=tile)
    assert rel_err(out32.cpu(), Ab.double() @ bf(W).double().T) < TOL[torch.bfloat16]


@pytest.mark.parametrize("M,N,K,S,tile", [(300, 384, 8 * 192, 8, -1), (520, 512, 16 * 256, 16, -1), (520, 512, 16 * 256, 16, 23), (130, 640, 16 * 128, 16, 20)])
def test_splitk_slices_on_xcds(M, N, K, S, tile):
    """Split-K over a multiple of 8 slices: the launch hands whole K slices to the XCDs (gemm_kernel's tile map).  Every partial is
    checked against ITS slice of the product, so a slice that lands in the wrong partial (or is computed twice) shows."""
    ops = _ops()
    wdt = torch.bfloat16
    A, W = rnd(M, K, seed=11), rnd(N, K, seed=12) * 0.05
    Ap = ops.PackedAct.from_dense(A.to(DEV).to(wdt))
    Wp = ops.PackedWeight(W.to(DEV).to(wdt))
    part = torch.full((S, M, N), float("nan"), device=DEV)
    ops.gemm(Ap, Wp, part, M=M, N=N, K=K, lda=K, ldc=N, splitk=S, tile=tile)
    Ar, Wr = bf(A).double(), bf(W).double()
    per = K // S
    for s_ in range(S):
        ref = Ar[:, s_ * per:(s_ + 1) * per] @ Wr[:, s_ * per:(s_ + 1) * per].T
        assert rel_err(part[s_].cpu(), ref) < 1e-5, s_


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,S", [(196, 1024, 4096, 4), (196, 768, 768, 1), (392, 1024, 1024, 2), (20, 768, 3080, 3)])
def test_splitk_reduce_ln(wdt, M, N, K, S):
    """Split-K GEMM (PARTIAL epilogue) + sp3_reduce_ln == Linear + residual + two LayerNorms."""
    ops = _ops()
    A, W, b, res = rnd(M, K, seed=1), rnd(N, K, seed=2) * 0.05, rnd(N, seed=3), rnd(M, N, seed=4)
    g1, b1, g2, b2 = rnd(N, seed=5) + 1, rnd(N, seed=6), rnd(N, seed=7) + 1, rnd(N, seed=8)
    part = torch.full((S, M, N), float("nan"), device=DEV)
    ops.gemm(A.to(DEV), W.to(DEV).to(wdt), part, M=M, N=N, K=K, lda=K, ldc=N, splitk=S)
    x = res.clone().to(DEV)
    o1 = torch.empty(M, N, device=DEV)
    o2 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.reduce_ln(part, S, M, N, bias=b.to(DEV), res=x, x_out=x, ln1=(g1.to(DEV), b1.to(DEV)), out1=o1,
                  ln2=(g2.to(DEV), b2.to(DEV)), out2=o2, eps=1e-6)
    Ar, Wr = (bf(A), bf(W)) if wdt == torch.bfloat16 else (A, W)
    xr = Ar.double() @ Wr.double().T + b.double() + res.double()
    assert rel_err(x.cpu(), xr) < TOL[wdt]
    assert rel_err(o1.cpu(), F.layer_norm(xr, (N,), g1.double(), b1.double(), 1e-6)) < (1e-4 if wdt == torch.float32 else 5e-3)
    assert rel_err(o2.float().cpu(), F.layer_norm(xr, (N,), g2.double(), b2.double(), 1e-6)) < 8e-3
    # no LayerNorm, no residual: plain finish
    y = torch.empty(M, N, device=DEV)
    ops.reduce_ln(part, S, M, N, bias=b.to(DEV), x_out=y)
    assert rel_err(y.cpu(), Ar.double() @ Wr.double().T + b.double()) < TOL[wdt]
    assert ops.conv_splitk(196, 256, 2304, torch.bfloat16) > 1 and ops.conv_splitk(3136, 256, 2304, torch.bfloat16) == 1


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C,N", [(196, 1024, 3072), (392, 768, 768), (20, 1024, 4096)])
def test_layernorm_folded_into_gemm(wdt, M, C, N):
    """Producer GEMM emits x (fp32), a fragment-order copy and per-32-column (sum, sumsq) partials; the consumer GEMM
    computes LayerNorm(x) @ W^T + b as rstd*(x @ (g*W)^T) - rstd*mean*s + (b + W @ beta) without an LN launch."""
    ops = _ops()
    K0 = 256
    A0, W0, b0, res = rnd(M, K0, seed=1), rnd(C, K0, seed=2) * 0.2, rnd(C, seed=3), rnd(M, C, seed=4) * 2 + 0.3
    g, beta = rnd(C, seed=5) * 0.3 + 1, rnd(C, seed=6) * 0.2
    W, b = rnd(N, C, seed=7) * 0.05, rnd(N, seed=8)
    x = res.clone().to(DEV)
    xp = ops.PackedAct(M, C, wdt, DEV)
    stats = torch.full((M, C // 32, 2), float("nan"), device=DEV)
    ops.gemm(A0.to(DEV), W0.to(DEV).to(wdt), x, M=M, N=C, K=K0, lda=K0, ldc=C, bias=b0.to(DEV), res1=x, ldr1=C,
             stats_out=stats, c2=xp)
    A0r, W0r = (bf(A0), bf(W0)) if wdt == torch.bfloat16 else (A0, W0)
    xr = A0r.double() @ W0r.double().T + b0.double() + res.double()
    assert rel_err(x.cpu(), xr) < TOL[wdt]
    assert torch.equal(xp.to_dense(), x.to(wdt))
    st = stats.cpu().double()
    assert rel_err(st[..., 0].sum(1), xr.sum(1)) < 1e-4 and rel_err(st[..., 1].sum(1), (xr * xr).sum(1)) < 1e-4
    # consumer
    Wf = (W * g[None, :]).to(wdt)
    s_n = Wf.float().sum(1)
    bfold = b + W @ beta
    y = torch.empty(M, N, device=DEV)
    ops.gemm(xp, ops.PackedWeight(Wf.to(DEV)), y, M=M, N=N, K=C, lda=C, ldc=N, bias=bfold.to(DEV),
             ln=ops.LnFold(stats, C, s_n.to(DEV), 1e-6))
    ref = F.layer_norm(xr, (C,), g.double(), beta.double(), 1e-6) @ W.double().T + b.double()
    assert rel_err(y.cpu(), ref) < (2e-4 if wdt == torch.float32 else 2e-2)


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
def test_gemm_split_a_ldw_batch(wdt):
    ops = _ops()
    # split-A (cat along K without the copy), as used by encode_feat_key
    M, N, K1, K2 = 40, 128, 1024, 768
    A1, A2, W = rnd(M, K1, seed=1), rnd(M, K2, seed=2), rnd(N, K1 + K2, seed=3)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(A1.to(DEV), W.to(DEV).to(wdt), out, M=M, N=N, K=K1 + K2, lda=K1, ldc=N, A2=A2.to(DEV), lda2=K2, K1=K1)
    cat = torch.cat((A1, A2), 1)
    Ar, Wr = (bf(cat), bf(W)) if wdt == torch.bfloat16 else (cat, W)
    assert rel_err(out.cpu(), Ar.double() @ Wr.double().T) < TOL[wdt]
    # batched with W row stride (the memory-read P.V^T GEMM): W [N, ldw] uses only the first K columns
    B, M, N, K, ldw, lda = 2, 20, 64, 40, 128, 48
    A, W = rnd(B, M, lda, seed=4), rnd(B, N, ldw, seed=5)
    res = rnd(B, M, N, seed=6)
    out = torch.empty(B, M, N, device=DEV)
    ops.gemm(A.to(DEV), W.to(DEV).to(wdt), out, M=M, N=N, K=K, lda=lda, ldc=N, ldw=ldw, batch=B, strideA=M * lda,
             strideW=N * ldw, strideC=M * N, res1=res.to(DEV), ldr1=N)
    Ar, Wr = (bf(A), bf(W)) if wdt == torch.bfloat16 else (A, W)
    ref = torch.einsum("bmk,bnk->bmn", Ar[:, :, :K].double(), Wr[:, :, :K].double()) + res.double()
    assert rel_err(out.cpu(), ref) < TOL[wdt]


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(1, 14, 14, 96, 256, 1), (2, 5, 4, 768, 768, 2), (1, 56, 56, 256, 256, 1),
                                                   (1, 9, 7, 128, 128, 1), (1, 4, 5, 384, 256, 1)])
def test_conv3x3(wdt, B, H, W, Cin, Cout, stride):
    ops = _ops()
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2) * 0.05, rnd(Cout, seed=3)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    r1, r2 = rnd(B, Cout, OH, OW, seed=4), rnd(B, Cout, OH, OW, seed=5)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    out = torch.empty(B, OH, OW, Cout, device=DEV)
    ops.conv3x3(nhwc(x).to(DEV), wp.to(DEV).to(wdt), out, B=B, H=H, W_=W, Cin=Cin, Cout=Cout, stride=stride,
                bias=b.to(DEV), res1=nhwc(r1).to(DEV), res2=nhwc(r2).to(DEV), relu_in=True, act=ops.ACT_RELU)
    xr, wr = (bf(F.relu(x)), bf(w)) if wdt == torch.bfloat16 else (F.relu(x), w)
    ref = F.relu(F.conv2d(xr.double(), wr.double(), b.double(), stride=stride, padding=1)) + r1.double() + r2.double()
    assert rel_err(out.cpu(), nhwc(ref)) < TOL[wdt]


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(1, 14, 14, 256, 256, 1), (1, 14, 14, 768, 768, 2), (1, 7, 7, 768, 256, 1), (2, 28, 28, 256, 256, 1)])
def test_conv3x3_splitk(wdt, B, H, W, Cin, Cout, stride):
    """small maps: K split over several workgroups per output tile (PARTIAL epilogue) + sp3_reduce_ln with the conv's
    bias / ReLU / two residuals"""
    ops = _ops()
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2) * 0.05, rnd(Cout, seed=3)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    assert ops.conv_splitk(B * OH * OW, Cout, 9 * Cin, wdt) > 1 or B > 1
    r1, r2 = rnd(B, Cout, OH, OW, seed=4), rnd(B, Cout, OH, OW, seed=5)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    wp = ops.PackedWeight(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV).to(wdt))
    out = torch.empty(B, OH, OW, Cout, device=DEV)
    ops.conv3x3(nhwc(x).to(DEV), wp, out, B=B, H=H, W_=W, Cin=Cin, Cout=Cout, stride=stride, bias=b.to(DEV), res1=nhwc(r1).to(DEV),
                res2=nhwc(r2).to(DEV), relu_in=True, act=ops.ACT_RELU, splitk_ws=torch.empty(1 << 21, device=DEV))
    xr, wr = (bf(F.relu(x)), bf(w)) if wdt == torch.bfloat16 else (F.relu(x), w)
    ref = F.relu(F.conv2d(xr.double(), wr.double(), b.double(), stride=stride, padding=1)) + r1.double() + r2.double()
    assert rel_err(out.cpu(), nhwc(ref)) < TOL[wdt]


@pytest.mark.parametrize("in_dt,out_dt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 56, 56, 256, 256), (2, 13, 9, 128, 128), (1, 8, 8, 64, 192), (1, 20, 28, 256, 128),
                                            (1, 3, 5, 384, 64)])
@pytest.mark.parametrize("variant", ["rcu1", "rcu2", "plain"])
@pytest.mark.parametrize("tile_px", ["8x8", "8x16", "8x16n32"])
def test_conv3x3_tile(in_dt, out_dt, B, H, W, Cin, Cout, variant, tile_px):
    """sp3_conv3x3_tile (LDS halo tile, packed bf16 weights; 8 x 8 and 8 x 16 pixel tiles) vs F.conv2d on the bf16-rounded operands;
    ragged tiles, input ReLU, bias / ReLU / two residuals, fp32 and bf16 maps."""
    ops = _ops()
    if tile_px != "8x8" and Cin % 128:
        with pytest.raises(RuntimeError, match="Cin"):
            ops.conv3x3(torch.zeros(B, H, W, Cin, device=DEV), ops.PackedWeight(torch.zeros(Cout, 9 * Cin, device=DEV, dtype=torch.bfloat16)),
                        torch.zeros(B, H, W, Cout, device=DEV), B=B, H=H, W_=W, Cin=Cin, Cout=Cout, force_tile_kernel=True, tile_px=tile_px)
        return
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2) * 0.05, rnd(Cout, seed=3)
    r1, r2 = rnd(B, Cout, H, W, seed=4), rnd(B, Cout, H, W, seed=5)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    wp = ops.PackedWeight(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV).to(torch.bfloat16))
    out = torch.full((B, H, W, Cout), float("nan"), device=DEV, dtype=out_dt)
    xin = nhwc(x).to(DEV).to(in_dt)
    relu_in = variant in ("rcu1", "rcu2")
    kw = dict(B=B, H=H, W_=W, Cin=Cin, Cout=Cout, relu_in=relu_in, force_tile_kernel=True, tile_px=tile_px)
    if variant == "rcu1":
        ops.conv3x3(xin, wp, out, bias=b.to(DEV), act=ops.ACT_RELU, **kw)
    elif variant == "rcu2":
        ops.conv3x3(xin, wp, out, bias=b.to(DEV), res1=nhwc(r1).to(DEV), res2=nhwc(r2).to(DEV), **kw)
    else:
        ops.conv3x3(xin, wp, out, **kw)
    xr = bf(x)
    xr = F.relu(xr) if relu_in else xr
    ref = F.conv2d(xr.double(), bf(w).double(), None if variant == "plain" else b.double(), padding=1)
    if variant == "rcu1":
        ref = F.relu(ref)
    elif variant == "rcu2":
        ref = ref + r1.double() + r2.double()
    tol = 2e-5 if out_dt == torch.float32 else 6e-3
    assert rel_err(out.float().cpu(), nhwc(ref)) < tol


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 112, 112, 256, 256), (4, 56, 56, 256, 256), (1, 100, 150, 128, 128)])
def test_conv3x3_tile_choice_by_size(B, H, W, Cin, Cout):
    """maps large enough to fill the chip with 8 x 16 tiles take the wide kernel on their own; both tiles give the same result up to
    the order of the fp32 partial sums, bf16 maps with bf16 residuals as in the DPT head"""
    ops = _ops()
    x, w, b = rnd(B, H, W, Cin, seed=1).to(DEV).to(torch.bfloat16), rnd(Cout, 9 * Cin, seed=2) * 0.05, rnd(Cout, seed=3).to(DEV)
    r1 = rnd(B, H, W, Cout, seed=4).to(DEV).to(torch.bfloat16)
    wp = ops.PackedWeight(w.to(DEV).to(torch.bfloat16))
    outs = {}
    for px in (None, "8x8", "8x16"):
        out = torch.full((B, H, W, Cout), float("nan"), device=DEV)
        ops.conv3x3(x, wp, out, B=B, H=H, W_=W, Cin=Cin, Cout=Cout, bias=b, relu_in=True, tile_px=px)
        outs[px] = out
    assert torch.isfinite(outs[None]).all()
    assert rel_err(outs["8x16"].cpu(), outs["8x8"].cpu()) < 1e-5
    assert torch.equal(outs[None], outs["8x16"])                 # >= 256 wide workgroups: the size rule picks the wide tile
    # a batch-1 56 x 56 map (112 wide workgroups of 64 channels): the 32-channel instance of the wide tile
    xs, small = x[:1, :56, :56].contiguous(), {}
    for px in (None, "8x8", "8x16n32"):
        out = torch.full((1, 56, 56, Cout), float("nan"), device=DEV)
        ops.conv3x3(xs, wp, out, B=1, H=56, W_=56, Cin=Cin, Cout=Cout, bias=b, relu_in=True, tile_px=px)
        small[px] = out
    assert rel_err(small["8x16n32"].cpu(), small["8x8"].cpu()) < 1e-5, rel_err(small["8x16n32"].cpu(), small["8x8"].cpu())
    if Cout == 256:
        assert torch.equal(small[None], small["8x16n32"])
    ob = torch.empty(B, H, W, Cout, device=DEV, dtype=torch.bfloat16)
    ops.conv3x3(x, wp, ob, B=B, H=H, W_=W, Cin=Cin, Cout=Cout, bias=b, res1=r1, relu_in=True)
    assert rel_err(ob.float().cpu(), (outs["8x8"] + r1.float()).cpu()) < 6e-3


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ks,C", [(4, 96), (2, 192)])
def test_conv_transpose(wdt, ks, C):
    ops = _ops()
    B, H, W = 2, 4, 5
    x, w, b = rnd(B, C, H, W, seed=1), rnd(C, C, ks, ks, seed=2) * 0.1, rnd(C, seed=3)
    wp = w.permute(2, 3, 1, 0).reshape(-1, C).contiguous()
    out = torch.empty(B, ks * H, ks * W, C, device=DEV)
    ops.conv_transpose_ks(x.permute(0, 2, 3, 1).contiguous().to(DEV), wp.to(DEV).to(wdt), out, B=B, H=H, W_=W, Cin=C, Cout=C,
                          ks=ks, bias=b.to(DEV))
    xr, wr = (bf(x), bf(w)) if wdt == torch.bfloat16 else (x, w)
    ref = F.conv_transpose2d(xr.double(), wr.double(), b.double(), stride=ks)
    assert rel_err(out.cpu(), ref.permute(0, 2, 3, 1)) < TOL[wdt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_packed_producers(dt):
    """Every producer's fragment-order output (out_packed) holds exactly what its row-major output holds."""
    ops = _ops()
    rows, C = 197, 1024
    x, g, b = rnd(rows, C, seed=1) * 2, rnd(C, seed=2) + 1, rnd(C, seed=3)
    xd, gd, bd = x.to(DEV), g.to(DEV), b.to(DEV)
    dense = torch.empty(rows, C, device=DEV, dtype=dt)
    pk = ops.PackedAct(rows, C, dt, DEV)
    ops.layernorm(xd, gd, bd, 1e-6, dense, rows=rows, C_=C)
    ops.layernorm(xd, gd, bd, 1e-6, pk, rows=rows, C_=C)
    assert torch.equal(pk.to_dense(), dense)
    # reduce_ln: one packed and one dense LayerNorm output
    part = rnd(2, rows, C, seed=4).to(DEV)
    pk2 = ops.PackedAct(rows, C, dt, DEV)
    d2 = torch.empty(rows, C, device=DEV, dtype=dt)
    ops.reduce_ln(part, 2, rows, C, bias=bd, ln1=(gd, bd), out1=pk2, ln2=(gd, bd), out2=d2)
    assert torch.equal(pk2.to_dense(), d2)
    # GEMM epilogue (bias + GELU) into a packed buffer
    M, N, K = 196, 768, 256
    A, W = rnd(M, K, seed=5).to(DEV), (rnd(N, K, seed=6) * 0.1).to(DEV).to(dt)
    od, op = torch.empty(M, N, device=DEV, dtype=dt), ops.PackedAct(M, N, dt, DEV)
    ops.gemm(A, W, od, M=M, N=N, K=K, lda=K, ldc=N, bias=rnd(N, seed=7).to(DEV), act=ops.ACT_GELU)
    ops.gemm(A, W, op, M=M, N=N, K=K, lda=K, ldc=N, bias=rnd(N, seed=7).to(DEV), act=ops.ACT_GELU)
    assert torch.equal(op.to_dense(), od)
    # attention output
    B, heads, Nq = 2, 3, 37
    Cc = heads * 64
    q, k = rnd(B, Nq, heads, 64, seed=8).to(DEV).to(dt), rnd(B, Nq, heads, 64, seed=9).to(DEV).to(dt)
    vt = torch.zeros(B * heads * 64, 64, device=DEV, dtype=dt)
    vt[:, :Nq] = rnd(B * heads * 64, Nq, seed=10).to(DEV).to(dt)
    ad, apk = torch.empty(B * Nq, Cc, device=DEV, dtype=dt), ops.PackedAct(B * Nq, Cc, dt, DEV)
    for o in (ad, apk):
        ops.attention(q, Nq * Cc, Cc, k, Nq * Cc, Cc, vt, 64, o, Cc, B=B, heads=heads, Nq=Nq, Nk=Nq, scale=0.125)
    assert torch.equal(apk.to_dense(), ad)
    # im2col
    img = rnd(2, 3, 32, 48, seed=11).to(DEV)
    cd, cp = torch.empty(12, 768, device=DEV, dtype=dt), ops.PackedAct(12, 768, dt, DEV)
    for o in (cd, cp):
        ops.im2col_patch(img, o, B=2, C_=3, H=32, W_=48, p=16, strides=img.stride())
    assert torch.equal(cp.to_dense(), cd)


# ----------------------------------------------------------------------------- RoPE / projection / attention
def _pos(B, nh, nw):
    from oracle import spann3r_oracle as O
    return O.positions(B, nh, nw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fwd", [1.0, -1.0])
def test_rope_2d_dropin(dtype, fwd):
    """sp3_rope_2d == curope.rope_2d semantics: in place on the [B,N,H,D] strided view of a [B,H,N,D] tensor."""
    from oracle import spann3r_oracle as O
    ops = _ops()
    B, H, nh, nw, D = 2, 12, 5, 7, 64
    N = nh * nw
    tok = rnd(B, H, N, D, seed=3)
    pos = _pos(B, nh, nw)
    t = tok.to(dtype).to(DEV)
    ops.rope_2d(t.transpose(1, 2), pos.to(DEV), 100.0, fwd)       # the call cuRoPE2D.forward makes (curope2d.py:39)
    ref = O.rope2d(tok.to(dtype).float(), pos, 100.0, fwd)
    assert rel_err(t.float().cpu(), ref) < (1e-5 if dtype == torch.float32 else 8e-3)
    with pytest.raises(RuntimeError):
        ops.rope_2d(t, pos[:1].to(DEV), 100.0, 1.0)               # batch mismatch -> RuntimeError like TORCH_CHECK


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
def test_proj_rope_vt(wdt):
    """Fused qkv projection: bias + RoPE(q,k) + per-head V^T store vs Linear -> reshape -> rope2d."""
    from oracle import spann3r_oracle as O
    from spann3r_amd.engine import _rope_tables
    ops = _ops()
    B, nh, nw, C, heads = 2, 4, 5, 768, 12
    P = nh * nw
    R = B * P
    x, W, b = rnd(R, C, seed=1), rnd(3 * C, C, seed=2) * 0.05, rnd(3 * C, seed=3)
    pos = _pos(B, nh, nw)
    cos, sin = _rope_tables(64, 100.0, DEV)
    npad = 64
    qk = torch.zeros(R, 2 * C, device=DEV, dtype=wdt)
    vt = torch.zeros(B * heads * 64, npad, device=DEV, dtype=wdt)
    ops.proj_rope_vt(x.to(DEV), W.to(DEV).to(wdt), b.to(DEV), qk, 2 * C, vt, npad, M=R, N=3 * C, K=C, lda=C, rope_cols=2 * C,
                     pos=pos.reshape(-1, 2).to(torch.int32).to(DEV), cos=cos, sin=sin, tokens=P, heads=heads)
    xr, Wr = (bf(x), bf(W)) if wdt == torch.bfloat16 else (x, W)
    y = (xr.double() @ Wr.double().T + b.double()).float().reshape(B, P, 3, heads, 64)
    q, k, v = (y[:, :, i].transpose(1, 2) for i in range(3))
    q, k = O.rope2d(q, pos), O.rope2d(k, pos)
    tol = 1e-5 if wdt == torch.float32 else 8e-3
    got = qk.float().cpu().reshape(B, P, 2, heads, 64)
    assert rel_err(got[:, :, 0].transpose(1, 2), q) < tol
    assert rel_err(got[:, :, 1].transpose(1, 2), k) < tol
    gv = vt.float().cpu().reshape(B, heads, 64, npad)
    assert rel_err(gv[..., :P], v.transpose(-1, -2)) < tol
    assert float(gv[..., P:].abs().max()) == 0.0


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,heads,Nq,Nk", [(1, 16, 196, 196), (2, 12, 20, 20), (1, 12, 50, 300), (1, 2, 1024, 1024), (2, 3, 17, 65)])
def test_attention(wdt, B, heads, Nq, Nk):
    ops = _ops()
    C = heads * 64
    q, k, v = rnd(B, Nq, heads, 64, seed=1), rnd(B, Nk, heads, 64, seed=2), rnd(B, Nk, heads, 64, seed=3)
    q = q * 2.0
    npad = (Nk + 63) // 64 * 64
    vt = torch.zeros(B, heads, 64, npad)
    vt[..., :Nk] = v.permute(0, 2, 3, 1)
    out = torch.empty(B * Nq, C, device=DEV)
    ops.attention(q.to(DEV).to(wdt), Nq * C, C, k.to(DEV).to(wdt), Nk * C, C, vt.reshape(-1, npad).to(DEV).to(wdt), npad, out, C,
                  B=B, heads=heads, Nq=Nq, Nk=Nk, scale=0.125)
    if wdt == torch.bfloat16:
        q, k, v = bf(q), bf(k), bf(v)
    a = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * 0.125, -1)
    ref = torch.einsum("bhqk,bkhd->bqhd", a, v.double()).reshape(B * Nq, C)
    assert rel_err(out.cpu(), ref) < (2e-5 if wdt == torch.float32 else 1e-2)
    if wdt == torch.bfloat16:
        ob = torch.empty(B * Nq, C, device=DEV, dtype=torch.bfloat16)
        ops.attention(q.to(DEV).to(wdt), Nq * C, C, k.to(DEV).to(wdt), Nk * C, C, vt.reshape(-1, npad).to(DEV).to(wdt), npad, ob, C,
                      B=B, heads=heads, Nq=Nq, Nk=Nk, scale=0.125)
        assert rel_err(ob.float().cpu(), ref) < 1.5e-2


def _pack_v_for_pv(v, npad):
    """host mirror of the PV-operand order written by the qkv_packed epilogue: v [B,N,H,64] -> [(b,h)][npad/32][4][64][8]"""
    B, N, H, _ = v.shape
    out = torch.zeros(B, H, npad // 32, 4, 64, 8, dtype=v.dtype)
    for n in range(N):
        u, kk = n // 32, n % 32
        w16 = kk % 16
        e, g = (w16 % 4) + 4 * (kk // 16), w16 // 4
        for db in range(4):
            out[:, :, u, db, g * 16:(g + 1) * 16, e] = v[:, n, :, db * 16:(db + 1) * 16]
    return out


@pytest.mark.parametrize("B,heads,Nq,Nk", [(1, 16, 196, 196), (2, 12, 20, 20), (1, 12, 50, 300), (2, 3, 17, 65),
                                           (1, 4, 1024, 1024), (2, 3, 100, 700), (1, 2, 70, 513),      # (long sequences: up to 16 key tiles, 4 per wave)
                                           (2, 4, 530, 530), (4, 2, 100, 600), (3, 8, 33, 520), (2, 12, 1024, 1024)])   # heads x B % 8 == 0 at >= 512 keys: whole (head, batch) pairs per XCD
def test_attention_packed(B, heads, Nq, Nk):
    """bf16 attention on fragment-order q/k + PV-order V == softmax(qk^T/8)v on the bf16-rounded operands."""
    ops = _ops()
    C = heads * 64
    q, k, v = rnd(B, Nq, heads, 64, seed=1) * 2, rnd(B, Nk, heads, 64, seed=2), rnd(B, Nk, heads, 64, seed=3)
    qb, kb_, vb = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    npq, npk = (Nq + 63) // 64 * 64, (Nk + 63) // 64 * 64
    qd = torch.zeros(B, npq, C, dtype=torch.bfloat16); qd[:, :Nq] = qb.reshape(B, Nq, C)
    kd = torch.zeros(B, npk, C, dtype=torch.bfloat16); kd[:, :Nk] = kb_.reshape(B, Nk, C)
    qp = ops.PackedAct.from_dense(qd.reshape(B * npq, C).to(DEV))
    kp = ops.PackedAct.from_dense(kd.reshape(B * npk, C).to(DEV))
    vp = _pack_v_for_pv(vb, npk).to(DEV)
    a = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", qb.double(), kb_.double()) * 0.125, -1)
    ref = torch.einsum("bhqk,bkhd->bqhd", a, vb.double()).reshape(B * Nq, C)
    out = torch.empty(B * Nq, C, device=DEV)
    ops.attention_packed(qp, C, 0, npq, kp, C, 0, npk, vp, out, C, B=B, heads=heads, Nq=Nq, Nk=Nk, scale=0.125)
    assert rel_err(out.cpu(), ref) < 1e-2
    outp = ops.PackedAct(B * Nq, C, torch.bfloat16, DEV)
    ops.attention_packed(qp, C, 0, npq, kp, C, 0, npk, vp, outp, C, B=B, heads=heads, Nq=Nq, Nk=Nk, scale=0.125)
    assert rel_err(outp.to_dense().float().cpu(), ref) < 1.5e-2


def test_proj_rope_vt_packed():
    """The qkv_packed epilogue writes the same numbers as the row-major one, in the layouts attention_packed reads."""
    from oracle import spann3r_oracle as O
    from spann3r_amd.engine import _rope_tables
    ops = _ops()
    B, nh, nw, C, heads = 2, 4, 5, 768, 12
    P, R, npad = nh * nw, 2 * nh * nw, 64
    x, W, b = rnd(R, C, seed=1), rnd(3 * C, C, seed=2) * 0.05, rnd(3 * C, seed=3)
    pos = _pos(B, nh, nw)
    cos, sin = _rope_tables(64, 100.0, DEV)
    args = dict(M=R, N=3 * C, K=C, lda=C, rope_cols=2 * C, pos=pos.reshape(-1, 2).to(torch.int32).to(DEV), cos=cos, sin=sin,
                tokens=P, heads=heads)
    Wd = W.to(DEV).to(torch.bfloat16)
    qk = torch.zeros(R, 2 * C, device=DEV, dtype=torch.bfloat16)
    vt = torch.zeros(B * heads * 64, npad, device=DEV, dtype=torch.bfloat16)
    ops.proj_rope_vt(x.to(DEV), Wd, b.to(DEV), qk, 2 * C, vt, npad, **args)
    qkp = ops.PackedAct(B * npad, 2 * C, torch.bfloat16, DEV)
    vtp = torch.zeros(B * heads * (npad // 32) * 4 * 64 * 8, device=DEV, dtype=torch.bfloat16)
    ops.proj_rope_vt(x.to(DEV), Wd, b.to(DEV), qkp.data, 0, vtp, npad, qkv_packed=True, **args)
    dense = qkp.to_dense().reshape(B, npad, 2 * C)
    assert torch.equal(dense[:, :P].reshape(R, 2 * C), qk)
    assert float(dense[:, P:].abs().max()) == 0.0
    v_row = vt.reshape(B, heads, 64, npad)[..., :P].permute(0, 3, 1, 2).contiguous()       # [B,P,H,64]
    assert torch.equal(vtp.reshape(B, heads, npad // 32, 4, 64, 8).cpu(), _pack_v_for_pv(v_row.cpu(), npad))


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("C", [768, 1024, 1792, 96])
def test_layernorm(C):
    ops = _ops()
    rows = 197
    x, g, b = rnd(rows, C, seed=1) * 3 + 0.5, rnd(C, seed=2) + 1, rnd(C, seed=3)
    ref = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-6)
    out = torch.empty(rows, C, device=DEV)
    ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-6, out, rows=rows, C_=C)
    assert rel_err(out.cpu(), ref) < 1e-5
    # strided output (writes into the right half of a wider buffer), bf16 output
    wide = torch.zeros(rows, 2 * C, device=DEV)
    ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-6, wide[:, C:], rows=rows, C_=C, ldo=2 * C)
    assert rel_err(wide[:, C:].cpu(), ref) < 1e-5 and float(wide[:, :C].abs().max()) == 0.0
    ob = torch.empty(rows, C, device=DEV, dtype=torch.bfloat16)
    ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-6, ob, rows=rows, C_=C)
    assert rel_err(ob.float().cpu(), ref) < 5e-3
    # transposed store into a [C, cap] bank at column offset 24
    cap = 256
    bank = torch.zeros(C, cap, device=DEV)
    ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, bank[:, 24:], rows=rows, C_=C, ldo=cap, transposed=True)
    ref5 = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-5)
    assert rel_err(bank[:, 24:24 + rows].cpu(), ref5.T) < 1e-5
    assert float(bank[:, :24].abs().max()) == 0.0 and float(bank[:, 24 + rows:].abs().max()) == 0.0


# ----------------------------------------------------------------------------- spatial-memory kernels
@pytest.mark.parametrize("pdt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("thresh", [0.0, 5e-4])
def test_softmax_thresh_and_colsum(thresh, pdt):
    ops = _ops()
    B, rows, M, ld = 2, 37, 1765, 1792
    S = rnd(B, rows, ld, seed=1) * 4
    Pm = torch.full((B, rows, ld), float("nan"), device=DEV)
    Mpad = (M + 7) // 8 * 8
    kb = 64 if pdt == torch.bfloat16 else 32
    Kp, rp = (M + kb - 1) // kb * kb, (rows + 15) // 16 * 16
    pk = torch.full((B, rp * Kp), float("nan"), device=DEV, dtype=pdt)
    ops.softmax_thresh(S.to(DEV), Pm, ld=ld, rows=rows, M=M, Mpad=Mpad, thresh=thresh, batch=B, strideS=rows * ld,
                       packed=pk, stride_packed=rp * Kp)
    # the fragment-order copy holds exactly the (rounded) probabilities, zero filled up to Kp
    dense = []
    for b_ in range(B):
        dense.append(ops.PackedAct(rows, Kp, pdt, DEV, data=pk[b_].view(ops.packed_shape(rows, Kp, pdt))).to_dense())
        assert torch.equal(dense[b_][:, :M], Pm[b_, :, :M].to(pdt)) and float(dense[b_][:, M:].float().abs().max()) == 0.0
    # packed-only call (what memory_read issues): same fragment-order bytes for the valid rows
    pk2 = torch.full((B, rp * Kp), float("nan"), device=DEV, dtype=pdt)
    ops.softmax_thresh(S.to(DEV), None, ld=ld, rows=rows, M=M, Mpad=M, thresh=thresh, batch=B, strideS=rows * ld,
                       packed=pk2, stride_packed=rp * Kp)
    d2 = ops.PackedAct(rows, Kp, pdt, DEV, data=pk2[0].view(ops.packed_shape(rows, Kp, pdt))).to_dense()
    assert torch.equal(d2, dense[0])
    # the streaming two-launch form the long-bank read uses (sp3_softmax_pack): same probabilities (v_exp_f32 instead of libm
    # expf: a last-bit difference before the bf16 rounding; entries at the threshold may flip), pad rows AND pads written as zeros
    pk3 = torch.full((B, rp * Kp), float("nan"), device=DEV, dtype=pdt)
    ops.softmax_pack(S.to(DEV), pk3, torch.empty(B * rows * 4, device=DEV), ld=ld, rows=rows, M=M, thresh=thresh, batch=B, strideS=rows * ld,
                     stride_packed=rp * Kp)
    assert not torch.isnan(pk3.float()).any()
    for b_ in range(B):
        d3 = ops.PackedAct(rows, Kp, pdt, DEV, data=pk3[b_].view(ops.packed_shape(rows, Kp, pdt))).to_dense()
        assert float((d3.double() - dense[b_].double()).abs().sum(-1).max()) < (5e-3 if pdt == torch.bfloat16 else 1e-4)
        assert float(d3[:, M:].float().abs().max()) == 0.0
    a = torch.softmax(S[..., :M].double(), -1)
    if thresh > 0:
        a32 = torch.softmax(S[..., :M], -1)
        a = torch.where(a32 < thresh, torch.zeros_like(a), a)
        a = a / a.sum(-1, keepdim=True)
    got = Pm.cpu()
    # entries within 1e-6 of the threshold may legitimately flip; compare row-wise with an L1 budget
    assert float((got[..., :M].double() - a).abs().sum(-1).max()) < 5e-3
    assert float(got[..., M:Mpad].abs().max()) == 0.0
    attn = rnd(M, seed=9).abs().to(DEV)
    before = attn.clone()
    ops.colsum_accum(Pm[0], ld, rows, M, attn)
    assert rel_err((attn - before).cpu(), got[0, :, :M].double().sum(0)) < 1e-5
    # column sums straight from the fragment-order copy: exactly the sums of the stored (rounded) probabilities
    attn2 = before.clone()
    pk[:, :] = torch.where(torch.isnan(pk), torch.zeros_like(pk), pk)      # pad rows of the test buffer were never written
    ops.colsum_packed(pk[0], rows, M, attn2)
    assert rel_err((attn2 - before).cpu(), dense[0][:, :M].double().sum(0).cpu()) < 1e-5


@pytest.mark.parametrize("wdt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M0", [0, 196, 588])
def test_bank_write_pack_stats_packed_gathers(wdt, M0):
    """sp3_bank_write (one launch per stored frame): raw copies, gamma_q (.) LN_k(k) and LN_v(v)^T in fragment order, the
    LN_q fold constants; then a read's S GEMM with LN_q folded against torch; then the prune gathers on the packed banks."""
    ops = _ops()
    P, C, cap = 196, 1024, 1024
    kb = 64 if wdt == torch.bfloat16 else 32
    cast = bf if wdt == torch.bfloat16 else (lambda t: t)
    names = ("gk", "bk", "gv", "bv", "gq", "bq")
    nrm = {n: (rnd(C, seed=20 + i) * 0.2 + (1.0 if n[0] == "g" else 0.0)) for i, n in enumerate(names)}
    bank = dict(k_raw=torch.zeros(cap, C, device=DEV), v_raw=torch.zeros(cap, C, device=DEV),
                k_hat=torch.zeros(ops.packed_shape(cap, C, wdt), dtype=wdt, device=DEV),
                v_hat_t=torch.zeros(ops.packed_shape(C, cap, wdt), dtype=wdt, device=DEV),
                s_bank=torch.zeros(cap, device=DEV), b_bank=torch.zeros(cap, device=DEV))
    norms = tuple(nrm[n].to(DEV) for n in names)
    alpha = 1.0 / 32.0
    ks, vs, M = [], [], M0
    if M0:                                               # earlier frames first: the new frame must not disturb them
        k0, v0 = rnd(M0, C, seed=1) * 2, rnd(M0, C, seed=2) * 2
        for f in range(M0 // P):
            ops.bank_write(k0[f * P:(f + 1) * P].to(DEV), v0[f * P:(f + 1) * P].to(DEV), bank, f * P, P, C, cap, norms, alpha)
        ks.append(k0); vs.append(v0)
    k1, v1 = rnd(P, C, seed=3) * 2 + 0.3, rnd(P, C, seed=4) * 2 - 0.1
    ops.bank_write(k1.to(DEV), v1.to(DEV), bank, M, P, C, cap, norms, alpha)
    ks.append(k1); vs.append(v1)
    K, V = torch.cat(ks), torch.cat(vs)
    M = M0 + P
    assert torch.equal(bank["k_raw"][:M].cpu(), K) and torch.equal(bank["v_raw"][:M].cpu(), V)
    khat = F.layer_norm(K.double(), (C,), nrm["gk"].double(), nrm["bk"].double(), 1e-5)
    vhat = F.layer_norm(V.double(), (C,), nrm["gv"].double(), nrm["bv"].double(), 1e-5)
    kp = ops.PackedAct(cap, C, wdt, DEV, data=bank["k_hat"]).to_dense()[:M].float().cpu()
    tol = 6e-3 if wdt == torch.bfloat16 else 2e-5
    assert rel_err(kp, khat * nrm["gq"].double()) < tol
    vt = ops.PackedAct(C, cap, wdt, DEV, data=bank["v_hat_t"]).to_dense().float().cpu()
    assert rel_err(vt[:, :M], vhat.T) < tol and float(vt[:, M:].abs().max()) == 0.0
    assert rel_err(bank["s_bank"][:M].cpu(), alpha * kp.double().sum(1)) < 1e-5
    assert rel_err(bank["b_bank"][:M].cpu(), alpha * (khat * nrm["bq"].double()).sum(1)) < 1e-5
    # the read's S GEMM: raw q in fragment order, LN_q folded
    q = rnd(P, C, seed=5) * 1.5 + 0.2
    qp = ops.PackedAct(P, C, wdt, DEV)
    qs = torch.zeros(P, C // 32, 2, device=DEV)
    ops.pack_stats(q.to(DEV), qp, qs, rows=P, C_=C)
    assert torch.equal(qp.to_dense().cpu(), q.to(wdt)) and rel_err(qs[..., 1].sum(1).cpu(), (q.double() ** 2).sum(1)) < 1e-5
    S = torch.full((P, cap), float("nan"), device=DEV)
    ops.gemm(qp, ops.PackedWeight.wrap(bank["k_hat"], M, C), S, M=P, N=M, K=C, lda=C, ldc=cap, alpha=alpha, bias=bank["b_bank"],
             ln=ops.LnFold(qs, C, bank["s_bank"], 1e-5))
    qn = F.layer_norm(q.double(), (C,), nrm["gq"].double(), nrm["bq"].double(), 1e-5)
    assert rel_err(S[:, :M].cpu(), qn @ khat.T * alpha) < (2e-2 if wdt == torch.bfloat16 else 5e-5)
    # P.V GEMM against the fragment-order V^T with the bank's k-extent (ldw = cap)
    Kp = (M + kb - 1) // kb * kb
    pr = torch.softmax(rnd(P, M, seed=6) * 3, -1)
    Pp = torch.zeros(P, Kp)
    Pp[:, :M] = pr
    A = ops.PackedAct.from_dense(Pp.to(DEV).to(wdt))
    out = torch.empty(P, C, device=DEV)
    ops.gemm(A, ops.PackedWeight.wrap(bank["v_hat_t"], C, cap), out, M=P, N=C, K=Kp, lda=Kp, ldc=C, ldw=cap, res1=q.to(DEV), ldr1=C)
    assert rel_err(out.cpu(), cast(pr).double() @ cast(vhat.float()).double() + q.double()) < (1e-2 if wdt == torch.bfloat16 else 5e-5)
    # prune gathers on the packed banks
    g = torch.Generator().manual_seed(7)
    n_sel = M - 40
    sel_h = torch.randperm(M, generator=g)[:n_sel]
    sel = sel_h.to(torch.int32).to(DEV)
    kd = torch.zeros_like(bank["k_hat"])
    vd = torch.full_like(bank["v_hat_t"], 7.0)
    ops.gather_packed_rows(bank["k_hat"], kd, sel, n_sel, C)
    ops.gather_packed_cols(bank["v_hat_t"], vd, sel, n_sel, cap, C, cap)
    assert torch.equal(ops.PackedAct(cap, C, wdt, DEV, data=kd).to_dense()[:n_sel].cpu(), ops.PackedAct(cap, C, wdt, DEV, data=bank["k_hat"]).to_dense().cpu()[sel_h])
    vg = ops.PackedAct(C, cap, wdt, DEV, data=vd).to_dense().cpu()
    assert torch.equal(vg[:, :n_sel], ops.PackedAct(C, cap, wdt, DEV, data=bank["v_hat_t"]).to_dense().cpu()[:, sel_h]) and float(vg[:, n_sel:].abs().max()) == 0.0


@pytest.mark.parametrize("wdt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,M,thresh", [(196, 1764, 5e-4), (196, 196, 5e-4), (37, 1765 // 4 * 4, 0.0), (50, 6000, 5e-4)])
def test_two_launch_memory_read(wdt, rows, M, thresh):
    """Score GEMM that leaves (max, sum exp) partials per 32 keys + P.V GEMM with the softmax loader + column sums, against
    torch: out = renorm(thresh(softmax(S))) @ V^T + q, mem_attn += column sums (spann3r/model.py:159-183)."""
    ops = _ops()
    C = 1024
    kb = 64 if wdt == torch.bfloat16 else 32
    cap = (M + 63) // 64 * 64 + 64
    cast = bf if wdt == torch.bfloat16 else (lambda t: t)
    q = rnd(rows, C, seed=1)
    Kh = rnd(M, C, seed=2) * 2.0
    V = rnd(M, C, seed=3)
    Khp = torch.zeros(cap, C); Khp[:M] = Kh
    Vt = torch.zeros(C, cap); Vt[:, :M] = V.T
    Vt[:, M:] = 3.0                                       # stale columns past the bank's end must not leak
    A = ops.PackedAct.from_dense(q.to(DEV).to(wdt))
    Wk = ops.PackedAct.from_dense(Khp.to(DEV).to(wdt))
    Wv = ops.PackedAct.from_dense(Vt.to(DEV).to(wdt))
    S = torch.full((rows, cap), float("nan"), device=DEV)
    nt = (M + 31) // 32
    st = torch.full((rows, nt, 2), float("nan"), device=DEV)       # compact [rows][ceil(M/32)][2]
    alpha = 0.25
    ops.gemm(A, ops.PackedWeight.wrap(Wk.data, M, C), S, M=rows, N=M, K=C, lda=C, ldc=cap, alpha=alpha, sm_stats_out=st)
    Sref = (cast(q).double() @ cast(Kh).double().T) * alpha
    assert rel_err(S[:, :M].cpu(), Sref) < (1e-5 if wdt == torch.float32 else 1e-5)
    # the partials: per 32-key group (max, sum exp(x - max)) of the stored scores
    Sg = S[:, :M].cpu().double()
    stv = st.cpu().double()
    for t in (0, nt // 2, nt - 1):
        blk = Sg[:, 32 * t:min(32 * t + 32, M)]
        assert rel_err(stv[:, t, 0], blk.max(1).values) < 1e-6
        assert rel_err(stv[:, t, 1], torch.exp(blk - blk.max(1, keepdim=True).values).sum(1)) < 1e-5
    out = torch.full((rows, C), float("nan"), device=DEV)
    zk = torch.full((rows, 4), float("nan"), device=DEV)
    S[:, M:] = float("nan")                               # scores past the bank's end are never used
    ops.gemm(S, ops.PackedWeight.wrap(Wv.data, C, cap), out, M=rows, N=C, K=M, lda=cap, ldc=C, ldw=cap, res1=q.to(DEV), ldr1=C,
             softmax=(st, thresh, zk))
    a = torch.softmax(Sg, -1)
    keep = torch.softmax(S[:, :M].cpu(), -1) >= thresh
    a = torch.where(keep, a, torch.zeros_like(a))
    zref = a.sum(-1)
    an = a / zref[:, None]
    ref = (cast(an.float()).double() if wdt == torch.bfloat16 else an) @ cast(V).double() + q.double()
    assert rel_err(zk[:, 0].cpu(), zref) < 2e-3            # entries at the threshold may flip
    assert rel_err(zk[:, 1].cpu(), Sg.max(1).values) < 1e-6 and rel_err(1.0 / zk[:, 2].cpu().double(), torch.exp(Sg - Sg.max(1, keepdim=True).values).sum(1)) < 1e-5
    assert rel_err(out.cpu(), ref) < (6e-3 if wdt == torch.bfloat16 else 2e-3)
    attn = rnd(cap, seed=9).abs().to(DEV)
    before = attn.clone()
    ops.colsum_softmax(S, cap, rows, M, zk, thresh, attn)
    assert rel_err((attn - before)[:M].cpu(), an.sum(0)) < 2e-3 and torch.equal(attn[M:], before[M:])
    # the same launch with the append bookkeeping of the frame that follows the read (sp3_mem_append semantics)
    attn2, count = before.clone(), torch.arange(cap, device=DEV).float()
    Pn = min(64, cap - M)
    ops.colsum_softmax(S, cap, rows, M, zk, thresh, attn2, count, Pn)
    assert torch.equal(attn2[:M], attn[:M]) and float(attn2[M:M + Pn].abs().max()) == 0 and torch.equal(attn2[M + Pn:], before[M + Pn:])
    c = count.cpu()
    assert torch.equal(c[:M], torch.arange(M).float() + 1) and float(c[M:M + Pn].abs().max()) == 0 and torch.equal(c[M + Pn:], torch.arange(M + Pn, cap).float())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_paired_launch_equals_two_launches(dt):
    """sp3_gemm2: two differently shaped groups of problems in one launch == the two launches, bit for bit"""
    ops = _ops()
    M, K = 196, 768
    A = ops.PackedAct.group(2, M, K, dt, DEV)
    A.data.copy_(rnd(*A.data.shape, seed=1).to(DEV).to(dt))
    outs, keep = {}, []
    for mode in ("two", "pair"):
        res = []
        ctx = ops.pair() if mode == "pair" else None
        if ctx:
            ctx.__enter__()
        for j, N in enumerate((2304, 1536)):
            Ws = ops.PackedWeightGroup([ops.PackedWeight((rnd(N, K, seed=10 * j + z) * 0.05).to(DEV).to(dt)) for z in range(2)])
            bias = rnd(2, N, seed=5 + j).to(DEV)
            out = torch.zeros(2, M, N, device=DEV)
            # (tile 0: the general kernel in both modes -- the lean instances' pairing is tests/test_gemm_lean_gpu.py)
            ops.gemm(A, Ws, out, M=M, N=N, K=K, lda=K, ldc=N, bias=bias, batch=2, strideA=A.stride, strideW=Ws.stride,
                     strideC=M * N, sb={"bias": N * 4}, act=ops.ACT_GELU if j else ops.ACT_NONE, tile=0)
            res.append(out)
            keep += [Ws, bias]                   # a paired launch is issued at __exit__: its operands must outlive the loop body
        if ctx:
            ctx.__exit__(None, None, None)
        torch.cuda.synchronize()
        outs[mode] = res
    for a, b in zip(outs["two"], outs["pair"]):
        assert torch.equal(a, b) and float(a.abs().max()) > 0
    with pytest.raises(RuntimeError):
        with ops.pair():
            pass


def test_attention_f32x3():
    """fp32 attention with bf16x3 products against float64, next to the exact-fp32 kernel"""
    ops = _ops()
    B, heads, Nq, Nk = 2, 3, 50, 196
    C = heads * 64
    q, k, v = rnd(B, Nq, heads, 64, seed=1) * 2, rnd(B, Nk, heads, 64, seed=2), rnd(B, Nk, heads, 64, seed=3)
    npk = (Nk + 63) // 64 * 64
    vt = torch.zeros(B * heads * 64, npk)
    vt.view(B, heads, 64, npk)[..., :Nk] = v.permute(0, 2, 3, 1)
    a = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * 0.125, -1)
    ref = torch.einsum("bhqk,bkhd->bqhd", a, v.double()).reshape(B * Nq, C)
    errs = {}
    for x3 in (False, True):
        ops.set_product_mode("f32x3" if x3 else "fp32")
        try:
            out = torch.empty(B * Nq, C, device=DEV)
            ops.attention(q.reshape(B, Nq, C).to(DEV), Nq * C, C, k.reshape(B, Nk, C).to(DEV), Nk * C, C, vt.to(DEV), npk, out, C,
                          B=B, heads=heads, Nq=Nq, Nk=Nk, scale=0.125)
            errs[x3] = rel_err(out.cpu(), ref)
        finally:
            ops.set_product_mode("fp32")
    assert errs[False] < 2e-6 and 1e-6 < errs[True] < 5e-5, errs


def test_attention_and_gemm_f16x3():
    """fp16 two-way split products (x = h + l * 2^-11, three fp16 MFMAs): within a few 1e-7 of float64, i.e. fp32-grade"""
    ops = _ops()
    B, heads, Nq, Nk = 2, 3, 50, 196
    C = heads * 64
    q, k, v = rnd(B, Nq, heads, 64, seed=1) * 2, rnd(B, Nk, heads, 64, seed=2), rnd(B, Nk, heads, 64, seed=3)
    npk = (Nk + 63) // 64 * 64
    vt = torch.zeros(B * heads * 64, npk)
    vt.view(B, heads, 64, npk)[..., :Nk] = v.permute(0, 2, 3, 1)
    a = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * 0.125, -1)
    ref = torch.einsum("bhqk,bkhd->bqhd", a, v.double()).reshape(B * Nq, C)
    M, N, K = 196, 768, 1024
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2) * 0.05, rnd(N, seed=3)
    gref = A.double() @ W.double().T + b.double()
    ops.set_product_mode("f16x3")
    try:
        out = torch.empty(B * Nq, C, device=DEV)
        ops.attention(q.reshape(B, Nq, C).to(DEV), Nq * C, C, k.reshape(B, Nk, C).to(DEV), Nk * C, C, vt.to(DEV), npk, out, C,
                      B=B, heads=heads, Nq=Nq, Nk=Nk, scale=0.125)
        ea = rel_err(out.cpu(), ref)
        eg = []
        for tile in (0, 1, 2):
            o = torch.empty(M, N, device=DEV)
            ops.gemm(A.to(DEV), W.to(DEV), o, M=M, N=N, K=K, lda=K, ldc=N, bias=b.to(DEV), tile=tile)
            eg.append(rel_err(o.cpu(), gref))
    finally:
        ops.set_product_mode("fp32")
    assert ea < 3e-6 and max(eg) < 3e-6, (ea, eg)


def test_gemm_f16x3_presplit_weights_same_bits():
    """ops.PackedWeight(halves=True) (the weights of an f16x3 engine: the (h, l) fp16 planes taken once at pack time, w_packed = 2) against
    the same fragment-order fp32 weight split inside the K loop (w_packed = 1): the same two conversions, so the SAME BITS -- on every
    register-ring tile, with a K that is not a whole k-block, and through the 3x3 convolution loader; any other product mode refuses
    the planes."""
    ops = _ops()
    ops.set_product_mode("f16x3")
    try:
        for (M, N, K) in ((196, 768, 1024), (100, 200, 168)):
            A, W, b = rnd(M, K, seed=1).to(DEV), (rnd(N, K, seed=2) * 0.05).to(DEV), rnd(N, seed=3).to(DEV)
            W[0, :8] = torch.tensor([0.0, 1e-9, -3e-6, 6.1e-5, 1.0, -1000.0, 3.0e4, 1e-3], device=DEV)     # zeros, fp16 subnormals, large values
            gref = A.double().cpu() @ W.double().cpu().T + b.double().cpu()
            w1, w2 = ops.PackedWeight(W), ops.PackedWeight(W, halves=True)
            assert not w1.halves and w2.halves and w1.data.shape == w2.data.shape and w1.data.dtype == w2.data.dtype
            for tile in (0, 1, 2):
                o1, o2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
                ops.gemm(A, w1, o1, M=M, N=N, K=K, lda=K, ldc=N, bias=b, tile=tile)
                ops.gemm(A, w2, o2, M=M, N=N, K=K, lda=K, ldc=N, bias=b, tile=tile)
                assert torch.equal(o1, o2), (M, N, K, tile)
                assert rel_err(o2.cpu(), gref) < 3e-6
        B, H, Wd, Cin, Cout = 1, 12, 10, 64, 32
        x, wc, bc = rnd(B, H, Wd, Cin, seed=4).to(DEV), (rnd(Cout, 3, 3, Cin, seed=5) * 0.05).to(DEV), rnd(Cout, seed=6).to(DEV)
        c1, c2 = torch.empty(B, H, Wd, Cout, device=DEV), torch.empty(B, H, Wd, Cout, device=DEV)
        ops.conv3x3(x, ops.PackedWeight(wc.reshape(Cout, -1).contiguous()), c1, B=B, H=H, W_=Wd, Cin=Cin, Cout=Cout, bias=bc, splitk_ws=torch.empty(1 << 21, device=DEV))
        ops.conv3x3(x, ops.PackedWeight(wc.reshape(Cout, -1).contiguous(), halves=True), c2, B=B, H=H, W_=Wd, Cin=Cin, Cout=Cout, bias=bc, splitk_ws=torch.empty(1 << 21, device=DEV))
        assert torch.equal(c1, c2)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wc.permute(0, 3, 1, 2).double().cpu(), bc.double().cpu(), padding=1).permute(0, 2, 3, 1)
        assert rel_err(c2.cpu(), ref) < 3e-6
    finally:
        ops.set_product_mode("fp32")
    with pytest.raises(ValueError):
        ops.gemm(A, w2, o2, M=M, N=N, K=K, lda=K, ldc=N, bias=b)            # fp32 products on fp16 planes: refused


def test_gemm_f32x3_products():
    """fp32 operands through three bf16 MFMAs per k-block (sp3_gemm_desc.f32x3): ~16 mantissa bits per product, fp32 accumulate"""
    ops = _ops()
    M, N, K = 196, 768, 1024
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2) * 0.05, rnd(N, seed=3)
    ref = A.double() @ W.double().T + b.double()
    outs = {}
    for x3 in (False, True):
        ops.set_product_mode("f32x3" if x3 else "fp32")
        try:
            for tile in (0, 1):
                out = torch.empty(M, N, device=DEV)
                ops.gemm(A.to(DEV), W.to(DEV), out, M=M, N=N, K=K, lda=K, ldc=N, bias=b.to(DEV), tile=tile)
                outs[(x3, tile)] = rel_err(out.cpu(), ref)
        finally:
            ops.set_product_mode("fp32")
    assert max(outs[(False, 0)], outs[(False, 1)]) < 2e-6                      # exact fp32 products
    assert 2e-6 < max(outs[(True, 0)], outs[(True, 1)]) < 3e-5, outs            # 2^-16-class products, far inside TF32's 5e-4
    # bf16 operands are untouched by the switch
    ops.set_product_mode("f32x3")
    try:
        o1, o2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        ops.gemm(A.to(DEV).to(torch.bfloat16), W.to(DEV).to(torch.bfloat16), o1, M=M, N=N, K=K, lda=K, ldc=N)
        ops.set_product_mode("fp32")
        ops.gemm(A.to(DEV).to(torch.bfloat16), W.to(DEV).to(torch.bfloat16), o2, M=M, N=N, K=K, lda=K, ldc=N)
    finally:
        ops.set_product_mode("fp32")
    assert torch.equal(o1, o2)


def test_cos_sim_append_prune_gather():
    ops = _ops()
    T, P, C = 3, 50, 1024
    k, wm = rnd(P, C, seed=1), rnd(T, P, C, seed=2)
    wm[1] = k * 1.7 + 0.01 * wm[1]
    score = torch.zeros(T, device=DEV)
    ops.cos_sim(k.to(DEV), wm.to(DEV), T, P, C, score, torch.empty(T * P, device=DEV))
    ref = torch.einsum("pc,tpc->tp", F.normalize(k.double(), dim=-1), F.normalize(wm.double(), dim=-1)).mean(-1)
    assert rel_err(score.cpu(), ref) < 1e-5 and float(score[1]) > 0.95
    # append bookkeeping
    M = 120
    count, attn = torch.arange(200.).to(DEV), torch.ones(200, device=DEV)
    ops.mem_append(count, attn, M, P)
    c = count.cpu()
    assert torch.equal(c[:M], torch.arange(M).float() + 1) and float(c[M:M + P].abs().max()) == 0
    assert float(attn[M:M + P].abs().max()) == 0 and float(attn[:M].min()) == 1
    # prune selection: weight = attn/count, protected (count < 10) = 1e8, ties by index
    M, top_k = 5096, 4000
    cnt = torch.arange(M - 1, -1, -1).float() // 196
    at = rnd(M, seed=5).abs() * (cnt + 1)
    sel = torch.zeros(top_k, dtype=torch.int32, device=DEV)
    ops.prune_select(at.to(DEV), cnt.to(DEV), M, 10.0, top_k, sel)
    wgt = torch.where(cnt < 10, torch.full_like(at, 1e8), at / cnt)
    order = sorted(range(M), key=lambda j: (-float(wgt[j]), j))[:top_k]
    assert sel.cpu().tolist() == order
    # the first eval prune of a 512x384 sequence holds 11 * 768 = 8448 tokens (> 8192)
    M2 = 8448
    cnt2 = torch.arange(M2 - 1, -1, -1).float() // 768
    at2 = rnd(M2, seed=15).abs() * (cnt2 + 1)
    sel2 = torch.zeros(top_k, dtype=torch.int32, device=DEV)
    ops.prune_select(at2.to(DEV), cnt2.to(DEV), M2, 10.0, top_k, sel2)
    w2 = torch.where(cnt2 < 10, torch.full_like(at2, 1e8), at2 / cnt2)
    assert sel2.cpu().tolist() == sorted(range(M2), key=lambda j: (-float(w2[j]), j))[:top_k]
    # gathers
    src = rnd(M, 64, seed=6)
    dst = torch.zeros(top_k, 64, device=DEV)
    ops.gather_rows(src.to(DEV), dst, sel, top_k, 64)
    assert torch.equal(dst.cpu(), src[order])
    srcb = src.to(torch.bfloat16)
    dstb = torch.zeros(top_k, 64, device=DEV, dtype=torch.bfloat16)
    ops.gather_rows(srcb.to(DEV), dstb, sel, top_k, 64)
    assert torch.equal(dstb.cpu(), srcb[order])
    cap = 5120
    st = torch.zeros(16, cap)
    st[:, :M] = rnd(16, M, seed=7)
    dt = torch.full((16, cap), 7.0, device=DEV)
    ops.gather_cols(st.to(DEV), cap, dt, cap, sel, top_k, cap, 16)
    assert torch.equal(dt[:, :top_k].cpu(), st[:, order]) and float(dt[:, top_k:].abs().max()) == 0
    d1 = torch.zeros(top_k, device=DEV)
    ops.gather_1d(at.to(DEV), d1, sel, top_k)
    assert torch.equal(d1.cpu(), at[order])


# ----------------------------------------------------------------------------- DPT helpers
def test_im2col_upsample_headfinal():
    from oracle import spann3r_oracle as O
    ops = _ops()
    B, H, W, p = 2, 32, 48, 16
    img = rnd(B, 3, H, W, seed=1)
    col = torch.empty(B * (H // p) * (W // p), 3 * p * p, device=DEV)
    ops.im2col_patch(img.to(DEV), col, B=B, C_=3, H=H, W_=W, p=p, strides=img.stride())
    ref = F.unfold(img, p, stride=p).transpose(1, 2).reshape(-1, 3 * p * p)
    assert torch.equal(col.cpu(), ref)
    colb = torch.empty(B * (H // p) * (W // p), 3 * p * p, device=DEV, dtype=torch.bfloat16)
    ops.im2col_patch(img.to(DEV), colb, B=B, C_=3, H=H, W_=W, p=p, strides=img.stride())
    assert torch.equal(colb.cpu(), ref.to(torch.bfloat16))
    nhwc = img.permute(0, 2, 3, 1).contiguous()          # pts3d layout
    sb, sy, sx, sc = nhwc.stride()
    ops.im2col_patch(nhwc.to(DEV), col, B=B, C_=3, H=H, W_=W, p=p, strides=(sb, sc, sy, sx))
    assert torch.equal(col.cpu(), ref)
    # bilinear x2, align_corners=True, with the dpt_head.py:57 crop
    x = rnd(B, 8, 4, 3, seed=2)
    up = torch.empty(B, 7, 5, 8, device=DEV)
    ops.upsample2x(x.permute(0, 2, 3, 1).contiguous().to(DEV), up, B=B, H=4, W_=3, C_=8, outH=7, outW=5)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)[:, :, :7, :5]
    assert rel_err(up.cpu(), ref.permute(0, 2, 3, 1)) < 1e-6
    # bf16 maps: the 16-byte kernel (C % 8 == 0) and the 8-byte one (C = 12) do the fp32 arithmetic of the fp32 kernel on the rounded input
    for C_ in (16, 12):
        xb = rnd(B, C_, 9, 6, seed=7).to(torch.bfloat16)
        nh = xb.permute(0, 2, 3, 1).contiguous().to(DEV)
        ub = torch.empty(B, 17, 11, C_, device=DEV, dtype=torch.bfloat16)
        uf = torch.empty(B, 17, 11, C_, device=DEV)
        ops.upsample2x(nh, ub, B=B, H=9, W_=6, C_=C_, outH=17, outW=11)
        ops.upsample2x(nh.float(), uf, B=B, H=9, W_=6, C_=C_, outH=17, outW=11)
        assert torch.equal(ub, uf.to(torch.bfloat16)), C_
        assert rel_err(uf.cpu(), F.interpolate(xb.float(), scale_factor=2, mode="bilinear", align_corners=True)[:, :, :17, :11].permute(0, 2, 3, 1)) < 1e-6
    # final 1x1 conv + postprocess
    pix, C = 1000, 128
    f, w, b = F.relu(rnd(pix, C, seed=3)), rnd(4, C, seed=4) * 0.1, rnd(4, seed=5)
    pts, conf, raw = torch.empty(pix, 3, device=DEV), torch.empty(pix, device=DEV), torch.empty(pix, 4, device=DEV)
    ops.head_final(f.to(DEV), w.to(DEV), b.to(DEV), pix, C, pts, conf, raw)
    r = (f.double() @ w.double().T + b.double()).float()
    assert rel_err(raw.cpu(), r) < 1e-5
    post = O.postprocess(r.T.reshape(1, 4, 1, pix))
    assert rel_err(pts.cpu(), post["pts3d"].reshape(pix, 3)) < 1e-5
    assert rel_err(conf.cpu(), post["conf"].reshape(pix)) < 1e-5
    # bf16 feature map (C = 128: the 16-byte kernel; 1003 pixels: ragged last workgroup) against float64 on the rounded features
    pix = 1003
    fb = F.relu(rnd(pix, C, seed=6)).to(torch.bfloat16)
    pts, conf, raw = torch.empty(pix, 3, device=DEV), torch.empty(pix, device=DEV), torch.empty(pix, 4, device=DEV)
    ops.head_final(fb.to(DEV), w.to(DEV), b.to(DEV), pix, C, pts, conf, raw)
    r = (fb.double() @ w.double().T + b.double()).float()
    assert rel_err(raw.cpu(), r) < 1e-5
    post = O.postprocess(r.T.reshape(1, 4, 1, pix))
    assert rel_err(pts.cpu(), post["pts3d"].reshape(pix, 3)) < 1e-5 and rel_err(conf.cpu(), post["conf"].reshape(pix)) < 1e-5
