"""The lean small-M GEMM instances (csrc/gemm_sm.hip, tiles 30..): every epilogue against float64 formulas on the rounded
operands AND against the general kernel (tile 0) on the same descriptor -- the per-frame step's Linears
(croco/models/blocks.py:73-79,94-112,149-169) must not change when the dispatcher moves them to the lean family."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _ops():
    from spann3r_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def bf(t):
    return t.to(BF).float()


def _dense(ops, pa, g, M, K):
    """problem g of a PackedAct.group as a dense [M, K] tensor"""
    shape = ops.packed_shape(M, K, BF)
    n = 1
    for v in shape:
        n *= v
    return ops.PackedAct(M, K, BF, DEV, data=pa.data.view(-1)[g * pa.stride: g * pa.stride + n].view(shape)).to_dense()


def _plan_of(ops, fn):
    """runs fn() (which issues exactly one GEMM) with a hook that records the tile the library planned"""
    seen = []
    orig = ops._gemm_launch

    def spy(d, what, name):
        from spann3r_amd import lib as L
        seen.append(L.load().sp3_gemm_plan(C.byref(d)) if d.tile < 0 else d.tile)
        return orig(d, what, name)
    ops._gemm_launch = spy
    try:
        fn()
    finally:
        ops._gemm_launch = orig
    return seen


@pytest.mark.parametrize("M,N,K,G,res,stats,ln", [
    (196, 1024, 1024, 1, True, True, False),      # val proj                        -> tile 34
    (196, 1024, 1024, 1, True, False, True),      # value_out: folded LayerNorm      -> tile 34
    (196, 1024, 4096, 1, True, True, False),      # val fc2                          -> tile 35
    (196, 768, 768, 2, True, True, False),        # decoder proj / cproj, both sides -> tile 36
    (196, 1024, 768, 1, False, True, False),      # pos patch embed                  -> tile 36
    (196, 768, 3072, 2, True, True, False),       # decoder fc2                      -> tile 37
    (196, 1024, 1792, 2, False, True, False),     # key MLP out                      -> tile 38
    (100, 768, 768, 1, True, True, False),        # ragged M
    (256, 1024, 1024, 1, True, True, False),
    # many rows (bm_kernel): whole-sequence encoder (10 frames x 196 tokens), a 512x512 frame (1024 tokens)
    (1960, 1024, 1024, 1, True, True, False),     # encoder proj            -> tile 56
    (1960, 1024, 4096, 1, True, True, False),     # encoder fc2             -> tile 57
    (1024, 768, 768, 2, True, True, False),       # decoder proj at 512x512 -> tile 58
    (1024, 768, 3072, 2, True, True, False),      #         fc2             -> tile 59
    (1024, 1024, 1792, 2, False, True, False),    # key MLP out             -> tile 60
    (1024, 1024, 1024, 1, True, False, True),     # value_out at 512x512: folded LayerNorm
    (300, 1024, 1024, 1, True, True, False),      # ragged: last row block half empty
])
def test_lean_stream(M, N, K, G, res, stats, ln):
    ops = _ops()
    A = ops.PackedAct.group(G, M, K, BF, DEV)
    Ad = [rnd(M, K, seed=1 + g) for g in range(G)]
    for g in range(G):
        A.at(g).data.view(-1)[:A.stride].copy_(ops.PackedAct.from_dense(Ad[g].to(DEV).to(BF)).data.view(-1))
    Wd = [rnd(N, K, seed=10 + g) * 0.05 for g in range(G)]
    gam, beta = rnd(K, seed=31) * 0.3 + 1, rnd(K, seed=32) * 0.2
    if ln:
        Wd = [w * gam[None, :] for w in Wd]
    Ws = ops.PackedWeightGroup([ops.PackedWeight(w.to(DEV).to(BF)) for w in Wd])
    bias = rnd(G, N, seed=5)
    R = rnd(G, M, N, seed=6) * 2 + 0.3
    lnf = None
    if ln:
        # statistics partials of the (fp32) rows whose bf16 copy is A
        xs = torch.stack(Ad)
        st = torch.stack((xs.reshape(G, M, K // 32, 32).sum(-1), (xs * xs).reshape(G, M, K // 32, 32).sum(-1)), -1).contiguous().to(DEV)
        s_n = torch.stack([bf(w).sum(1) for w in Wd]).contiguous().to(DEV)
        lnf = ops.LnFold(st, K, s_n, 1e-6, sb_stats=M * (K // 32) * 8, sb_s=N * 4)
    outs = {}
    for tile in (-1, 0):
        out = torch.full((G, M, N), float("nan"), device=DEV)
        so = torch.full((G, M, N // 32, 2), float("nan"), device=DEV) if stats else None
        c2 = ops.PackedAct.group(G, M, N, BF, DEV) if stats else None
        kw = dict(M=M, N=N, K=K, lda=K, ldc=N, bias=bias.to(DEV), res1=R.to(DEV) if res else None, ldr1=N if res else 0,
                  stats_out=so, c2=c2, ln=lnf, tile=tile)
        sb = {"bias": N * 4}
        if stats:
            sb["stats_out"], sb["c2"] = M * (N // 32) * 8, c2.stride * 2
        if G > 1:
            kw.update(batch=G, strideA=A.stride, strideW=Ws.stride, strideC=M * N, sb=sb)
        planned = _plan_of(ops, lambda: ops.gemm(A, Ws, out, **kw))
        assert (planned[0] >= 30) == (tile < 0), planned
        outs[tile] = (out, so, c2)
    out, so, c2 = outs[-1]
    for g in range(G):
        x = bf(Ad[g]).double()
        if ln:
            xf = Ad[g].double()
            mu, var = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
            # what the fold computes: rstd * (x_bf16 . W'^T) - rstd * mu * s + b
            ref = (x @ bf(Wd[g]).double().T) / torch.sqrt(var + 1e-6) - (mu / torch.sqrt(var + 1e-6)) * bf(Wd[g]).double().sum(1)[None] + bias[g].double()
        else:
            ref = x @ bf(Wd[g]).double().T + bias[g].double()
        if res:
            ref = ref + R[g].double()
        assert rel_err(out[g].cpu(), ref) < 2e-5, (g, rel_err(out[g].cpu(), ref))
        assert rel_err(out[g].cpu(), outs[0][0][g].cpu()) < 2e-5            # the general kernel: summation order only
        if stats:
            assert torch.equal(_dense(ops, c2, g, M, N), out[g].to(BF))
            st = so[g].cpu().double()
            xr = out[g].cpu().double()
            assert rel_err(st[..., 0], xr.reshape(M, N // 32, 32).sum(-1)) < 1e-5
            assert rel_err(st[..., 1], (xr * xr).reshape(M, N // 32, 32).sum(-1)) < 1e-5


@pytest.mark.parametrize("M,N,K,G,act", [(196, 4096, 1024, 1, "gelu"), (196, 3072, 768, 2, "gelu"), (196, 3072, 768, 1, "none"), (60, 4096, 1024, 1, "gelu"),
                                         (1960, 4096, 1024, 1, "gelu"), (1024, 4096, 1024, 1, "gelu"), (1024, 3072, 768, 2, "gelu"), (500, 3072, 768, 1, "none"),
                                         # >= 16 M-tiles of 256 rows (the 512 x 512 whole-sequence encoder): the blocked tile maps of bm_kernel
                                         (4096, 4096, 1024, 1, "gelu"),      # M <= N: 8 M-tiles x all z-slots per block (mblk)
                                         (8192, 2048, 1024, 1, "gelu")])     # M > N: 8 N-tiles x 4 M-tile slots per block (xm = 2)
def test_lean_packed_fc1(M, N, K, G, act):
    """norm (folded) + fc1 + GELU into fragment order (croco/models/blocks.py:74-75,129)"""
    ops = _ops()
    A = ops.PackedAct.group(G, M, K, BF, DEV)
    Ad = [rnd(M, K, seed=1 + g) * 2 + 0.1 for g in range(G)]
    for g in range(G):
        A.at(g).data.view(-1)[:A.stride].copy_(ops.PackedAct.from_dense(Ad[g].to(DEV).to(BF)).data.view(-1))
    gam = rnd(K, seed=31) * 0.3 + 1
    Wd = [rnd(N, K, seed=10 + g) * 0.05 * gam[None, :] for g in range(G)]
    Ws = ops.PackedWeightGroup([ops.PackedWeight(w.to(DEV).to(BF)) for w in Wd])
    bias = rnd(G, N, seed=5)
    xs = torch.stack(Ad)
    st = torch.stack((xs.reshape(G, M, K // 32, 32).sum(-1), (xs * xs).reshape(G, M, K // 32, 32).sum(-1)), -1).contiguous().to(DEV)
    s_n = torch.stack([bf(w).sum(1) for w in Wd]).contiguous().to(DEV)
    lnf = ops.LnFold(st, K, s_n, 1e-6, sb_stats=M * (K // 32) * 8, sb_s=N * 4)
    outs = {}
    for tile in (-1, 0):
        h = ops.PackedAct.group(G, M, N, BF, DEV)
        kw = dict(M=M, N=N, K=K, lda=K, ldc=N, bias=bias.to(DEV), act=ops.ACT_GELU if act == "gelu" else ops.ACT_NONE, ln=lnf, tile=tile)
        if G > 1:
            kw.update(batch=G, strideA=A.stride, strideW=Ws.stride, strideC=h.stride, sb={"bias": N * 4})
        planned = _plan_of(ops, lambda: ops.gemm(A, Ws, h, **kw))
        assert (planned[0] >= 30) == (tile < 0), planned
        outs[tile] = h
    for g in range(G):
        xf = Ad[g].double()
        mu, var = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
        rstd = 1.0 / torch.sqrt(var + 1e-6)
        ref = rstd * (bf(Ad[g]).double() @ bf(Wd[g]).double().T) - rstd * mu * bf(Wd[g]).double().sum(1)[None] + bias[g].double()
        if act == "gelu":
            ref = F.gelu(ref)
        got, old = _dense(ops, outs[-1], g, M, N).float().cpu(), _dense(ops, outs[0], g, M, N).float().cpu()
        assert rel_err(got, ref) < 4e-3                       # bf16 output rounding
        assert rel_err(got, old) < 8e-3 and float((got != old).float().mean()) < 0.02      # same values up to a rare last-bit flip


def _pos(B, nh, nw):
    ys, xs = torch.meshgrid(torch.arange(nh), torch.arange(nw), indexing="ij")
    return torch.stack((ys.reshape(-1), xs.reshape(-1)), -1)[None].expand(B, -1, -1).contiguous()


@pytest.mark.parametrize("B,nh,nw,C,heads,kind", [(1, 14, 14, 1024, 16, "qkv"), (1, 14, 14, 768, 12, "qkv"), (1, 14, 14, 768, 12, "q"),
                                                   (2, 10, 10, 768, 12, "qkv"), (2, 8, 12, 1024, 16, "kv"),
                                                   (10, 14, 14, 1024, 16, "qkv"), (1, 32, 32, 1024, 16, "qkv"), (1, 32, 32, 768, 12, "qkv"),
                                                   (1, 32, 32, 768, 12, "kv"), (1, 32, 32, 768, 12, "q"), (3, 14, 14, 768, 12, "qkv"),
                                                   (4, 32, 32, 1024, 16, "qkv")])          # 16 M-tiles x 3 z-slots: bm_kernel's blocked map (mblk)
def test_lean_rope_vt(B, nh, nw, C, heads, kind):
    """q/k/v projection with folded LayerNorm, bias, 2-D RoPE and the attention kernel's layouts: lean instance vs general kernel"""
    from spann3r_amd.engine import _rope_tables
    ops = _ops()
    P = nh * nw
    R, npad = B * P, (P + 63) // 64 * 64
    N = {"qkv": 3 * C, "kv": 2 * C, "q": C}[kind]
    rope_cols = {"qkv": 2 * C, "kv": C, "q": C}[kind]
    x = rnd(R, C, seed=1) * 2 + 0.2
    gam = rnd(C, seed=31) * 0.3 + 1
    W, b = rnd(N, C, seed=2) * 0.05 * gam[None, :], rnd(N, seed=3)
    A = ops.PackedAct.from_dense(x.to(DEV).to(BF))
    Wp = ops.PackedWeight(W.to(DEV).to(BF))
    st = torch.stack((x.reshape(R, C // 32, 32).sum(-1), (x * x).reshape(R, C // 32, 32).sum(-1)), -1).contiguous().to(DEV)
    lnf = ops.LnFold(st, C, bf(W).sum(1).to(DEV), 1e-6)
    pos = _pos(B, nh, nw).reshape(-1, 2).to(torch.int32).to(DEV)
    cos, sin = _rope_tables(64, 100.0, DEV)
    outs = {}
    for tile in (-1, 0):
        qkp = ops.PackedAct(B * npad, rope_cols, BF, DEV)
        vtp = torch.zeros(B * heads * npad * 64, device=DEV, dtype=BF) if kind != "q" else None
        planned = _plan_of(ops, lambda: ops.proj_rope_vt(A, Wp, b.to(DEV), qkp.data, 0, vtp, npad, M=R, N=N, K=C, lda=C, rope_cols=rope_cols,
                                                         pos=pos, cos=cos, sin=sin, tokens=P, heads=heads, qkv_packed=True, ln=lnf, tile=tile))
        assert (planned[0] >= 30) == (tile < 0), planned
        outs[tile] = (qkp.to_dense().float().cpu(), None if vtp is None else vtp.float().cpu())
    (q1, v1), (q0, v0) = outs[-1], outs[0]
    assert float(q1.abs().max()) > 0.1
    assert rel_err(q1, q0) < 8e-3 and float((q1 != q0).float().mean()) < 0.02
    if npad > P:
        pad = q1.reshape(B, npad, rope_cols)[:, P:]
        assert float(pad.abs().max()) == 0.0
    if v1 is not None:
        assert float(v1.abs().max()) > 0.1
        assert rel_err(v1, v0) < 8e-3 and float((v1 != v0).float().mean()) < 0.02


@pytest.mark.parametrize("M,npad,grid", [(196, 256, 14), (1024, 1024, 32)])
def test_lean_pair_equals_two_launches(M, npad, grid):
    """sp3_gemm2 on a lean instance (a decoder layer's q/k/v + cross k/v projections, both sides): bit-identical to the two launches"""
    from spann3r_amd.engine import _rope_tables
    ops = _ops()
    K, heads, P, B = 768, 12, M, 1
    A = ops.PackedAct.group(2, M, K, BF, DEV)
    A.data.copy_(rnd(*A.data.shape, seed=1).to(DEV).to(BF))
    pos = _pos(1, grid, grid).reshape(-1, 2).to(torch.int32).to(DEV)
    cos, sin = _rope_tables(64, 100.0, DEV)
    st = (rnd(2, M, K // 32, 2, seed=4).abs() * 30 + 40).to(DEV)
    st[..., 0] *= 0.05
    res, keep = {}, []
    for mode in ("two", "pair"):
        outs = []
        ctx = ops.pair() if mode == "pair" else None
        if ctx:
            ctx.__enter__()
        for j, (N, rc) in enumerate(((3 * K, 2 * K), (2 * K, K))):
            Ws = ops.PackedWeightGroup([ops.PackedWeight((rnd(N, K, seed=10 * j + z) * 0.05).to(DEV).to(BF)) for z in range(2)])
            bias, s_n = rnd(2, N, seed=5 + j).to(DEV), rnd(2, N, seed=7 + j).to(DEV)
            qk = torch.zeros(ops.packed_shape(2 * B * npad, rc, BF), device=DEV, dtype=BF)
            vt = torch.zeros(2 * B * heads * npad * 64, device=DEV, dtype=BF)
            ops.proj_rope_vt(A, Ws, bias, qk, 0, vt, npad, M=M, N=N, K=K, lda=K, rope_cols=rc, pos=pos, cos=cos, sin=sin, tokens=P,
                             heads=heads, qkv_packed=True, batch=2, strideA=A.stride, strideW=Ws.stride, strideC=B * npad * rc,
                             ln=ops.LnFold(st, K, s_n, 1e-6, sb_stats=M * (K // 32) * 8, sb_s=N * 4),
                             sb={"bias": N * 4, "vt": B * heads * npad * 64 * 2})
            outs += [qk, vt]
            keep += [Ws, bias, s_n]              # a paired launch is issued at __exit__: its operands must outlive the loop body
        if ctx:
            ctx.__exit__(None, None, None)
        torch.cuda.synchronize()
        res[mode] = outs
    for a, b in zip(res["two"], res["pair"]):
        assert torch.equal(a, b) and float(a.float().abs().max()) > 0


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,variant", [(1, 14, 14, 256, 256, 1, "rcu1"), (1, 14, 14, 256, 256, 1, "rcu2"), (1, 7, 7, 256, 256, 1, "rcu2"),
                                                           (1, 28, 28, 256, 256, 1, "rcu2"), (1, 14, 14, 768, 768, 2, "bias"), (1, 14, 14, 384, 256, 1, "plain"),
                                                           (1, 7, 7, 768, 256, 1, "plain"), (1, 28, 28, 192, 256, 1, "plain"), (2, 9, 11, 256, 256, 1, "rcu2")])
@pytest.mark.parametrize("maps", ["fp32", "bf16"])
def test_lean_conv3x3_small_maps(B, H, W, Cin, Cout, stride, variant, maps):
    """DPT small-map 3x3 convolutions (croco/models/dpt_block.py:33-75,95-113) on the lean conv instances (tiles 40 / 41): one
    launch, against F.conv2d on the rounded operands and against the general implicit-GEMM kernel"""
    ops = _ops()
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2) * 0.05, rnd(Cout, seed=3)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    r1, r2 = rnd(B, Cout, OH, OW, seed=4), rnd(B, Cout, OH, OW, seed=5)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    wp = ops.PackedWeight(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV).to(BF))
    kw = dict(B=B, H=H, W_=W, Cin=Cin, Cout=Cout, stride=stride)
    relu_in = variant in ("rcu1", "rcu2")
    mdt = torch.float32 if maps == "fp32" else BF             # bf16 mode of the DPT heads: input, residual and output maps in bf16
    if maps == "bf16":
        x, r1, r2 = bf(x), bf(r1), bf(r2)
    if variant == "rcu1":
        kw.update(bias=b.to(DEV), relu_in=True, act=ops.ACT_RELU)
    elif variant == "rcu2":
        kw.update(bias=b.to(DEV), relu_in=True, res1=nhwc(r1).to(DEV).to(mdt), res2=nhwc(r2).to(DEV).to(mdt))
    elif variant == "bias":
        kw.update(bias=b.to(DEV))
    outs = {}
    for tile in ((-1, 0) if maps == "fp32" else (-1,)):       # (the general kernel takes fp32 residual maps only)
        out = torch.full((B, OH, OW, Cout), float("nan"), device=DEV, dtype=mdt)
        planned = _plan_of(ops, lambda: ops.conv3x3(nhwc(x).to(DEV).to(mdt), wp, out, tile=tile, **kw))
        assert (planned[0] >= 40) == (tile < 0), planned
        outs[tile] = out.float().cpu()
    xr = bf(F.relu(x) if relu_in else x)
    ref = F.conv2d(xr.double(), bf(w).double(), b.double() if variant != "plain" else None, stride=stride, padding=1)
    if variant == "rcu1":
        ref = F.relu(ref)
    if variant == "rcu2":
        ref = ref + r1.double() + r2.double()
    assert rel_err(outs[-1], nhwc(ref)) < (2e-5 if maps == "fp32" else 4e-3)
    if maps == "fp32":
        assert rel_err(outs[-1], outs[0]) < 2e-5


@pytest.mark.parametrize("M,tile", [(196, 42), (784, 66), (1024, 67), (1000, 66)])
def test_lean_split_a_key_mlp(M, tile):
    """first layer of the key MLPs (spann3r/model.py:299-303: Linear(1792, 1792) + GELU on cat(feat, dec[-1])) with both halves of the
    concatenation as fragment-order bf16 matrices, both MLPs in one launch (tile 42; the many-row instances 66 / 67 above 256 rows:
    batch 4 and 512x512 frames), against the row-major split-A path"""
    ops = _ops()
    E, D, Kd, G = 1024, 768, 1792, 2
    f = [rnd(M, E, seed=1 + g) for g in range(G)]
    nrm = [rnd(M, D, seed=5 + g) for g in range(G)]
    Fp = ops.PackedAct.group(G, M, E, BF, DEV)
    Np = ops.PackedAct.group(G, M, D, BF, DEV)
    for g in range(G):
        Fp.at(g).data.view(-1)[:Fp.stride].copy_(ops.PackedAct.from_dense(f[g].to(DEV).to(BF)).data.view(-1))
        Np.at(g).data.view(-1)[:Np.stride].copy_(ops.PackedAct.from_dense(nrm[g].to(DEV).to(BF)).data.view(-1))
    Wd = [rnd(Kd, Kd, seed=10 + g) * 0.05 for g in range(G)]
    Ws = ops.PackedWeightGroup([ops.PackedWeight(w.to(DEV).to(BF)) for w in Wd])
    bias = rnd(G, Kd, seed=3)
    h = ops.PackedAct.group(G, M, Kd, BF, DEV)
    planned = _plan_of(ops, lambda: ops.gemm(Fp, Ws, h, M=M, N=Kd, K=Kd, lda=E, ldc=Kd, bias=bias.to(DEV), act=ops.ACT_GELU, A2=Np, lda2=D, K1=E,
                                             batch=G, strideA=Fp.stride, strideW=Ws.stride, strideC=h.stride,
                                             sb={"bias": Kd * 4, "A2": Np.stride * 2}))
    assert planned == [tile]
    # the general kernel on the fp32 row-major halves (what the model ran before)
    fd, nd = torch.stack(f).to(DEV).to(BF).float().contiguous(), torch.stack(nrm).to(DEV).to(BF).float().contiguous()
    h0 = ops.PackedAct.group(G, M, Kd, BF, DEV)
    ops.gemm(fd, Ws, h0, M=M, N=Kd, K=Kd, lda=E, ldc=Kd, bias=bias.to(DEV), act=ops.ACT_GELU, A2=nd, lda2=D, K1=E,
             batch=G, strideA=M * E, strideW=Ws.stride, strideC=h0.stride, sb={"bias": Kd * 4, "A2": M * D * 4})
    for g in range(G):
        ref = F.gelu(torch.cat((bf(f[g]), bf(nrm[g])), 1).double() @ bf(Wd[g]).double().T + bias[g].double())
        got, old = _dense(ops, h, g, M, Kd).float().cpu(), _dense(ops, h0, g, M, Kd).float().cpu()
        assert rel_err(got, ref) < 4e-3
        assert rel_err(got, old) < 8e-3 and float((got != old).float().mean()) < 0.02


def test_layernorm_dual_groups():
    """sp3_layernorm_dual: fp32 rows + a bf16 fragment-order copy whose row groups start on 16-row boundaries"""
    ops = _ops()
    G, R, C_ = 3, 196, 768
    x = rnd(G * R, C_, seed=1) * 3 + 0.5
    g, b = rnd(C_, seed=2) * 0.3 + 1, rnd(C_, seed=3) * 0.2
    out = torch.empty(G * R, C_, device=DEV)
    dual = ops.PackedAct.group(G, R, C_, BF, DEV)
    ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-6, out, rows=G * R, C_=C_, dual=dual, group_rows=R)
    ref = F.layer_norm(x.double(), (C_,), g.double(), b.double(), 1e-6)
    assert rel_err(out.cpu(), ref) < 1e-5
    for k in range(G):
        assert torch.equal(_dense(ops, dual, k, R, C_), out[k * R:(k + 1) * R].to(BF))


# ----------------------------------------------------------------------------- the bank's fill level as device state (round 6)
@pytest.mark.parametrize("M", [196, 1764, 2044, 4000, 8192])
def test_memory_read_with_the_extent_on_the_device(M):
    """The two launches of the short-bank memory read (tiles 43 / 44, spann3r/model.py:159-183) sized for a BUCKET of bank tokens,
    with the real count read from a device int32 (sp3_gemm_desc.dyn_n): scores, softmax statistics, fused output and the kept
    mass are bit-identical to the launches that carry the count as an argument -- for several counts through the SAME bucket
    descriptor, which is what lets one captured hipGraph serve a growing bank."""
    ops = _ops()
    P, C, thr = 196, 1024, 5e-4
    bucket = (M + 2047) // 2048 * 2048
    cap = bucket + 64
    q = rnd(P, C, seed=1) * 1.5 + 0.1
    qp, qs = ops.PackedAct(P, C, BF, DEV), torch.zeros(P, C // 32, 2, device=DEV)
    ops.pack_stats(q.to(DEV), qp, qs, rows=P, C_=C)
    Kh = ops.PackedAct.from_dense((rnd(cap, C, seed=2) * 2).to(DEV).to(BF))           # rows past M hold stale (non-zero) tokens
    Vt = ops.PackedAct.from_dense(rnd(C, cap, seed=3).to(DEV).to(BF))
    sb, bb = (rnd(cap, seed=4) * 0.1).to(DEV), (rnd(cap, seed=5) * 0.1).to(DEV)
    state = torch.zeros(4, dtype=torch.int32, device=DEV)
    nt_cap = (cap + 31) // 32

    def read(n_arg, dyn):
        S = torch.full((P, cap), float("nan"), device=DEV)
        st = torch.full((P * nt_cap * 2,), float("nan"), device=DEV)
        out, zk = torch.full((P, C), float("nan"), device=DEV), torch.full((P, 4), float("nan"), device=DEV)
        tiles = _plan_of(ops, lambda: ops.gemm(qp, ops.PackedWeight.wrap(Kh.data, n_arg, C), S, M=P, N=n_arg, K=C, lda=C, ldc=cap, alpha=1 / 32.,
                                               bias=bb, ln=ops.LnFold(qs, C, sb, 1e-5), sm_stats_out=st, dyn_n=dyn))
        tiles += _plan_of(ops, lambda: ops.gemm(S, ops.PackedWeight.wrap(Vt.data, C, cap), out, M=P, N=C, K=n_arg, lda=cap, ldc=C, ldw=cap,
                                                res1=q.to(DEV), ldr1=C, softmax=(st, thr, zk), dyn_n=dyn))
        assert tiles == [43, 44], tiles
        return S, st, out, zk
    ref = read(M, None)
    ops.bank_state_set(state, M, 3)
    got = read(bucket, state)
    ng = (M + 31) // 32
    assert torch.equal(ref[0][:, :M], got[0][:, :M]) and torch.isnan(got[0][:, M:]).all()       # nothing written past the bank's end
    assert torch.equal(ref[1][:P * ng * 2], got[1][:P * ng * 2])
    assert torch.equal(ref[2], got[2]) and torch.equal(ref[3], got[3])
    assert not torch.isnan(got[2]).any()
    # a smaller count through the same (bucket-sized) descriptor
    M2 = max(4, (M // 2) // 4 * 4)
    ref2 = read(M2, None)
    ops.bank_state_set(state, M2, 1)
    got2 = read(bucket, state)
    assert torch.equal(ref2[0][:, :M2], got2[0][:, :M2]) and torch.equal(ref2[2], got2[2]) and torch.equal(ref2[3], got2[3])
    # the general kernels do not take a device-side extent
    with pytest.raises(RuntimeError, match="dyn_n"):
        ops.gemm(qp, ops.PackedWeight.wrap(Kh.data, bucket, C), torch.empty(P, cap, device=DEV), M=P, N=bucket, K=C, lda=C, ldc=cap, dyn_n=state, tile=0)


@pytest.mark.parametrize("wdt", [BF, torch.float32])
def test_bank_write_and_similarity_window_from_device_state(wdt):
    """sp3_bank_write with the first row read from the device state and sp3_cos_sim_state (the working-memory window [M - wm P, M) of
    spann3r/model.py:97-118 from the same state): identical to the launches that carry M / wm as arguments, at fill levels that
    start inside and on a token-group boundary."""
    ops = _ops()
    P, C, cap, Tmax = 196, 1024, 1024, 5
    names = ("gk", "bk", "gv", "bv", "gq", "bq")
    norms = tuple((rnd(C, seed=20 + i) * 0.2 + (1.0 if n[0] == "g" else 0.0)).to(DEV) for i, n in enumerate(names))

    def bank():
        return dict(k_raw=torch.zeros(cap, C, device=DEV), v_raw=torch.zeros(cap, C, device=DEV),
                    k_hat=torch.zeros(ops.packed_shape(cap, C, wdt), dtype=wdt, device=DEV),
                    v_hat_t=torch.zeros(ops.packed_shape(C, cap, wdt), dtype=wdt, device=DEV),
                    s_bank=torch.zeros(cap, device=DEV), b_bank=torch.zeros(cap, device=DEV))
    a, b = bank(), bank()
    state = torch.zeros(4, dtype=torch.int32, device=DEV)
    for f in range(4):                                   # M = 0, 196, 392, 588: 196 % 8 = 4 -> aligned and unaligned starts
        k, v = (rnd(P, C, seed=10 + f) * 2).to(DEV), (rnd(P, C, seed=30 + f) * 2).to(DEV)
        ops.bank_write(k, v, a, f * P, P, C, cap, norms, 1 / 32.)
        ops.bank_state_set(state, f * P, min(f, Tmax))
        ops.bank_write(k, v, b, -12345, P, C, cap, norms, 1 / 32., state=state)
    for n in a:
        assert torch.equal(a[n], b[n]), n
    M = 4 * P
    probe = (rnd(P, C, seed=99)).to(DEV)
    for wm in (1, 3, 4):
        s1, s2 = torch.full((Tmax,), 7.0, device=DEV), torch.full((Tmax,), 7.0, device=DEV)
        ops.cos_sim(probe, a["k_raw"][M - wm * P:M], wm, P, C, s1, torch.empty(Tmax * P, device=DEV))
        ops.bank_state_set(state, M, wm)
        ops.cos_sim_state(probe, a["k_raw"], Tmax, P, C, state, s2, torch.empty(Tmax * P, device=DEV))
        assert torch.equal(s1, s2) and float(s2[wm:].min() if wm < Tmax else 7.0) == 7.0      # entries past wm untouched


@pytest.mark.parametrize("rows,M,S_k", [(320, 2240, 8), (1024, 4096, 8), (300, 2244, 16), (257, 640, 8),
                                        (1024, 12292, 8)])      # 4 x 97 score tiles (more than one round of workgroups), ragged last tile
def test_long_bank_read_without_a_score_matrix(rows, M, S_k):
    """The long-bank memory read of a > 256-row frame (spann3r/model.py:159-183 at attn_thresh = 0) as: score stage that writes
    bf16 p~ = exp(s - group max) + (max, sum) per 64-key group (tile 45), sp3_prob_merge, P.V stage with the groups' rescale in its loop
    (tile 46, split-K partials), reduce + q, column sums -- against float64 softmax on the fp32 scores of the general kernel.  Ragged
    ends (M not a multiple of 128 / 64, rows not a multiple of 256), stale tokens past the bank's end, and the same launches with the
    token count read from the device through a larger bucket."""
    ops = _ops()
    C = 1024
    cap = (M + 127) // 128 * 128 + 256
    rows_pad, ngc = (rows + 255) // 256 * 256, cap // 64
    q = rnd(rows, C, seed=1) * 1.5 + 0.1
    qp, qs = ops.PackedAct(rows, C, BF, DEV), torch.zeros(rows, C // 32, 2, device=DEV)
    ops.pack_stats(q.to(DEV), qp, qs, rows=rows, C_=C)
    Kd = rnd(cap, C, seed=2) * 2
    Kh = ops.PackedAct.from_dense(Kd.to(DEV).to(BF))                          # rows past M: stale tokens
    Vd = rnd(C, cap, seed=3)
    Vd[:, M:] = 3.0
    Vt = ops.PackedAct.from_dense(Vd.to(DEV).to(BF))
    sb, bb = (rnd(cap, seed=4) * 0.1).to(DEV), (rnd(cap, seed=5) * 0.1).to(DEV)
    alpha = 1 / 32.
    # reference scores: the general kernel's fp32 S (same folded LayerNorm), then float64 softmax / products on the bf16-rounded V_hat
    S = torch.empty(rows, cap, device=DEV)
    ops.gemm(qp, ops.PackedWeight.wrap(Kh.data, M, C), S, M=rows, N=M, K=C, lda=C, ldc=cap, alpha=alpha, bias=bb, ln=ops.LnFold(qs, C, sb, 1e-5), tile=0)
    p = torch.softmax(S[:, :M].double().cpu(), -1)
    ref = p @ bf(Vd[:, :M]).double().T + q.double()
    state = torch.zeros(4, dtype=torch.int32, device=DEV)

    def read(n_arg, dyn):
        pk = torch.zeros(ops.packed_shape(rows, cap, BF), dtype=BF, device=DEV)
        pt = ops.PackedAct(rows, cap, BF, DEV, data=pk)
        stats = torch.full((ngc * rows_pad * 2,), float("nan"), device=DEV)
        scale = torch.full((ngc * rows_pad,), float("nan"), device=DEV)
        part = torch.full((S_k * rows * C,), float("nan"), device=DEV)
        out = torch.empty(rows, C, device=DEV)
        attn = torch.ones(cap, device=DEV)
        t = _plan_of(ops, lambda: ops.gemm(qp, ops.PackedWeight.wrap(Kh.data, n_arg, C), pt, M=rows, N=n_arg, K=C, lda=C, ldc=cap, alpha=alpha, bias=bb,
                                           ln=ops.LnFold(qs, C, sb, 1e-5), sm_stats_out=stats, dyn_n=dyn))
        ops.prob_merge(stats, scale, rows, n_arg, cap, dyn_n=dyn)
        t += _plan_of(ops, lambda: ops.gemm(pt, ops.PackedWeight.wrap(Vt.data, C, cap), part, M=rows, N=C, K=n_arg, lda=cap, ldc=C, ldw=cap, splitk=S_k,
                                            softmax=(scale, 0.0, None), dyn_n=dyn))
        assert t == [45, 46], t
        ops.reduce_ln(part, S_k, rows, C, res=q.to(DEV), ldres=C, x_out=out, ldx=C)
        ops.colsum_prob(pk, scale, rows, n_arg, cap, attn, dyn_n=dyn)
        return out, attn, pt, scale
    out, attn, pt, scale = read(M, None)
    assert rel_err(out.cpu(), ref) < 6e-3, rel_err(out.cpu(), ref)
    assert rel_err((attn[:M] - 1).cpu(), p.sum(0)) < 6e-3 and torch.equal(attn[M:], torch.ones(cap - M, device=DEV))
    # p~ * scale is the softmax itself (bf16 rounding of p~)
    ng = (M + 63) // 64
    sc = scale.view(ngc, rows_pad)[:ng, :rows].T.repeat_interleave(64, 1)[:, :M].cpu().double()
    assert rel_err(pt.to_dense()[:, :M].double().cpu() * sc, p) < 6e-3
    assert float(pt.to_dense()[:, M:ng * 64].abs().max() if ng * 64 > M else 0.0) == 0.0           # keys past the end inside the last group
    # the token count on the device, launches sized for a larger bucket: bit-identical
    ops.bank_state_set(state, M, 0)
    out2, attn2, _, _ = read(min(cap, (M + 1023) // 1024 * 1024 + 128), state)
    assert torch.equal(out, out2) and torch.equal(attn, attn2)


@pytest.mark.parametrize("B,nh,nw", [(1, 14, 14), (2, 10, 10), (1, 3, 4)])
def test_cross_attention_with_the_q_projection_inside(B, nh, nw):
    """The decoder's cross-attention with its query projection computed per attention workgroup (sp3_attention_packed_qproj,
    croco/models/blocks.py:149-169) against the two launches it replaces: q/k-style projection GEMM (folded norm2, bias, 2-D RoPE,
    lean tile 31) + sp3_attention_packed -- two decoder sides as the two groups of a grouped launch, images per side = B."""
    from spann3r_amd.engine import _rope_tables
    ops = _ops()
    D, heads = 768, 12
    P = nh * nw
    R, npad = B * P, (P + 63) // 64 * 64
    xg = ops.PackedAct.group(2, R, D, BF, DEV)
    stg = torch.zeros(2, R, D // 32, 2, device=DEV)
    Ws, ss, bs = [], [], []
    for s_ in (0, 1):
        x = rnd(R, D, seed=1 + s_) * 2 + 0.2
        ops.pack_stats(x.to(DEV), xg.at(s_), stg[s_], rows=R, C_=D)
        gam = rnd(D, seed=31 + s_) * 0.3 + 1
        W = rnd(D, D, seed=2 + s_) * 0.05 * gam[None, :]
        Ws.append(ops.PackedWeight(W.to(DEV).to(BF)))
        ss.append(bf(W).sum(1))
        bs.append(rnd(D, seed=3 + s_))
    Wg = ops.PackedWeightGroup(Ws)
    sg, bg = torch.stack(ss).to(DEV).contiguous(), torch.stack(bs).to(DEV).contiguous()
    pos = _pos(B, nh, nw).reshape(-1, 2).to(torch.int32).to(DEV)
    cos, sin = _rope_tables(64, 100.0, DEV)
    kp = ops.PackedAct.from_dense((rnd(2 * B * npad, D, seed=9) * 1.5).to(DEV).to(BF))
    vtp = (rnd(2 * B * heads * npad * 64, seed=10)).to(DEV).to(BF)
    Rp = xg.rows_pad
    # the two launches
    cqp = torch.zeros(ops.packed_shape(2 * B * npad, D, BF), dtype=BF, device=DEV)
    ops.proj_rope_vt(xg, Wg, bg, cqp, 0, None, npad, M=R, N=D, K=D, lda=D, rope_cols=D, pos=pos, cos=cos, sin=sin, tokens=P, heads=heads, qkv_packed=True,
                     ln=ops.LnFold(stg, D, sg, 1e-6, sb_stats=R * (D // 32) * 8, sb_s=D * 4), batch=2, strideA=xg.stride, strideW=Wg.stride,
                     strideC=B * npad * D, sb={"bias": D * 4})
    ref = ops.PackedAct.group(2, R, D, BF, DEV)
    ops.attention_packed(cqp, D, 0, npad, kp, D, 0, npad, vtp, ref, D, B=2 * B, heads=heads, Nq=P, Nk=P, scale=0.125, o_group=B, o_group_rows=Rp)
    got = ops.PackedAct.group(2, R, D, BF, DEV)
    ops.attention_packed_qproj(xg, stg, Wg, sg, bg, pos, cos, sin, kp, D, 0, npad, vtp, got, D, B=2 * B, heads=heads, Nq=P, Nk=P, scale=0.125,
                               o_group=B, o_group_rows=Rp, eps=1e-6, stats_group_stride=R * (D // 32) * 2, vec_group_stride=D)
    for s_ in (0, 1):
        a, b = _dense(ops, got, s_, R, D).float().cpu(), _dense(ops, ref, s_, R, D).float().cpu()
        assert float(b.abs().max()) > 0.05
        frac = float((a != b).float().mean())
        print("side %d: rel diff %.2e, differing elements %.4f" % (s_, rel_err(a, b), frac))
        assert rel_err(a, b) < 8e-3 and frac < 0.05
