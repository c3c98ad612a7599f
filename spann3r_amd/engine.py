"""Stage-level host code of the hot path: packs weights for the kernels and sequences the launches.

One `Engine` per (model, device, precision).  Every method mirrors one reference stage
(file:line cited per method) and only launches HIP kernels through spann3r_amd.ops; activations
live in a persistent workspace (stable addresses -> the per-frame step is hipGraph-capturable).

Precision: 'fp32' -> fp32 weights, v_mfma_f32_16x16x4_f32 (parity mode, <=1e-3 vs the CPU oracle);
           'bf16' -> bf16 weights / q,k,v / memory bank, v_mfma_f32_16x16x32_bf16, fp32 accumulate,
                     fp32 residual stream, LayerNorm, softmax and DPT feature maps (bench mode).
"""
import math
import os

import torch

from . import ops
from .lib import ACT_NONE, ACT_GELU, ACT_RELU
from .config import Spann3RConfig


def _rope_tables(max_pos, base, device):
    """cos/sin [max_pos, 16] exactly as the reference's torch fallback builds them
    (croco/models/pos_embed.py:118-129 with D = head_dim/2 = 32)."""
    half = 32
    inv_freq = 1.0 / (base ** (torch.arange(0, half, 2).float() / half))
    t = torch.arange(max_pos, dtype=torch.float32)
    ang = torch.einsum("i,j->ij", t, inv_freq)
    return ang.cos().contiguous().to(device), ang.sin().contiguous().to(device)


def _rope_tables_narrow(max_pos, base, head_dim, device):
    """cos/sin [max_pos, 16] for RoPE2D on heads NARROWER than 64 (use_feat: 16 heads of 48, spann3r/model.py:225-235) in the
    kernels' 64-slot head layout: per axis D = head_dim / 2 with D / 2 frequencies base^(-2i/D) (croco/models/pos_embed.py:118-129);
    frequency i of an axis sits in table column i, the unused columns hold the identity rotation (cos 1, sin 0)."""
    D = head_dim // 2
    nf = D // 2
    assert nf <= 16 and head_dim % 4 == 0
    inv_freq = 1.0 / (base ** (torch.arange(0, D, 2).float() / D))
    t = torch.arange(max_pos, dtype=torch.float32)
    ang = torch.einsum("i,j->ij", t, inv_freq)
    cos, sin = torch.ones(max_pos, 16), torch.zeros(max_pos, 16)
    cos[:, :nf], sin[:, :nf] = ang.cos(), ang.sin()
    return cos.contiguous().to(device), sin.contiguous().to(device)


def narrow_head_slots(head_dim):
    """slot (0..63) of every dimension of a head of head_dim < 64 under RoPE2D: dimension d = axis * D + half * D/2 + i lands at
    axis * 32 + half * 16 + i, which is where the kernels' 64-wide rotary pairing (column c with c ^ 16 inside each 32-wide axis
    half) expects it; dot products and the attention output do not care about a permutation shared by q and k."""
    D = head_dim // 2
    nf = D // 2
    idx = []
    for d in range(head_dim):
        a, j = divmod(d, D)
        h, i = divmod(j, nf)
        idx.append(a * 32 + h * 16 + i)
    return torch.tensor(idx, dtype=torch.long)


class Engine:
    def __init__(self, cfg: Spann3RConfig, params: dict, device, precision="fp32"):
        assert precision in ("fp32", "f32x3", "f32x6", "f16x3", "bf16")
        self.cfg = cfg
        self.device = torch.device(device)
        self.precision = precision
        # "f32x3" / "f32x6" / "f16x3": fp32 weights and activations, GEMM products through split bf16 / fp16 MFMAs -- the product mode
        # belongs to THIS engine and is activated (per thread) at each of its entry points, see activate()
        self.f32_mode = ops.PRODUCT_MODES[precision]
        self.wdt = torch.bfloat16 if precision == "bf16" else torch.float32
        self.adt = self.wdt          # dtype of activations that only feed GEMMs
        import os
        self.packed_attn = precision == "bf16"      # fragment-order q/k/v + wave-split attention kernel
        # DPT feature maps: bf16 in bf16 mode (round 4: every map is a GEMM / convolution operand that the MFMA rounds to bf16 anyway;
        # stored as bf16 they cost half the bytes in the convolutions' loaders, the upsamplers and the residual adds), fp32 otherwise
        # bf16 DPT maps carry bf16 residuals, which only the lean small-map convolutions (and the LDS-tiled kernel) read: with
        # SP3_LEAN_GEMM=0 (the one documented A/B switch: the whole default path against the general kernels) the maps stay fp32
        self.mdt = torch.bfloat16 if (precision == "bf16" and ops.LEAN) else torch.float32
        self._ws = {}
        self.fuse_cross_q = True     # grouped decoder (bf16): the cross-attention's q projection runs inside its attention launch
        self._arena, self._arena_off, self._arenas = None, 0, []
        self._splitA_plan = {}       # rows -> does the library hold a lean instance for the packed split-A key MLP (encode_feat_keys_grouped)
        self._pos_cache = {}
        # RoPE tables for every grid the build supports, allocated ONCE: captured graphs hold their addresses
        self.max_pos = 256
        self.cos, self.sin = _rope_tables(self.max_pos, cfg.rope_base, self.device)
        self.rope_narrow = None
        if cfg.use_feat and cfg.mem_pos_enc:
            self.rope_narrow = _rope_tables_narrow(self.max_pos, cfg.rope_base, cfg.val_dim // cfg.enc_heads, self.device)
        self.w = {}
        self._pack(params)

    def activate(self):
        """Every entry point of the engine (and of the SpatialMemory built on it) calls this first: the fp32-operand GEMMs launched
        by this thread from here on use this engine's product mode (ops.set_product_mode is per thread -- two models of
        different precision in one process do not share it)."""
        ops.set_product_mode(self.f32_mode)
        return self

    # ------------------------------------------------------------------ weights
    def _pack(self, p):
        cfg, dev, wdt = self.cfg, self.device, self.wdt
        w = self.w
        # f16x3: every product is three fp16 MFMAs of x = h + l 2^-11; the weights' (h, l) planes are taken ONCE, here (ops.PackedWeight)
        hv = self.precision == "f16x3"

        def mat(t):      # MFMA operand: [N, K] in the compute dtype, re-ordered once into fragment order
            return ops.PackedWeight(t.detach().to(dev, torch.float32).reshape(t.shape[0], -1).to(wdt).contiguous(), halves=hv)

        def vec(t):      # biases / LN parameters stay fp32
            return t.detach().to(dev, torch.float32).contiguous()

        def conv3(t):    # [Cout,Cin,3,3] -> [Cout, (ky,kx,ci)]
            return ops.PackedWeight(t.detach().to(dev, torch.float32).permute(0, 2, 3, 1).reshape(t.shape[0], -1).to(wdt).contiguous(), halves=hv)

        def convt(t):    # ConvTranspose2d [Cin,Cout,k,k] -> [(ky,kx,co), ci]
            return ops.PackedWeight(t.detach().to(dev, torch.float32).permute(2, 3, 1, 0).reshape(-1, t.shape[0]).to(wdt).contiguous(), halves=hv)

        def fold(dst, weight, bias, norm):
            """LayerNorm folded into the consuming Linear (DESIGN.md "LN fold"): W' = gamma (.) W in the MFMA dtype,
            s_n = sum_k W'_nk taken from the ROUNDED W' (the identity (x-mu).W' = x.W' - mu.s must hold for the operand
            the MFMA really multiplies), b' = b + W . beta."""
            W = weight.detach().to(dev, torch.float32)
            g = p[norm + ".weight"].detach().to(dev, torch.float32)
            beta = p[norm + ".bias"].detach().to(dev, torch.float32)
            Wf = (W * g[None, :]).to(wdt)
            w[dst + ".w"] = ops.PackedWeight(Wf.contiguous(), halves=hv)
            w[dst + ".s"] = Wf.float().sum(1).contiguous()
            w[dst + ".b"] = (bias.detach().to(dev, torch.float32) + W @ beta).contiguous()

        def block(dst, src):
            fold(dst + "qkv", p[src + "attn.qkv.weight"], p[src + "attn.qkv.bias"], src + "norm1")
            w[dst + "proj.w"], w[dst + "proj.b"] = mat(p[src + "attn.proj.weight"]), vec(p[src + "attn.proj.bias"])
            w[dst + "fc2.w"], w[dst + "fc2.b"] = mat(p[src + "mlp.fc2.weight"]), vec(p[src + "mlp.fc2.bias"])

        w["patch.w"], w["patch.b"] = mat(p["dust3r.patch_embed.proj.weight"]), vec(p["dust3r.patch_embed.proj.bias"])
        if not cfg.use_feat:
            w["pospatch.w"], w["pospatch.b"] = mat(p["pos_patch_embed.proj.weight"]), vec(p["pos_patch_embed.proj.bias"])
        for i in range(cfg.enc_depth):
            block("enc%d." % i, "dust3r.enc_blocks.%d." % i)
            fold("enc%d.fc1" % i, p["dust3r.enc_blocks.%d.mlp.fc1.weight" % i], p["dust3r.enc_blocks.%d.mlp.fc1.bias" % i],
                 "dust3r.enc_blocks.%d.norm2" % i)
        for i in range(cfg.val_depth):
            src = "value_encoder.%d." % i
            if cfg.use_feat:
                # 16 heads of 48 (spann3r/model.py:225,228) -> zero-padded to 64 per head: qkv rows, proj columns (_attn_core)
                Hh, hd, Cv = cfg.enc_heads, cfg.val_dim // cfg.enc_heads, cfg.val_dim
                qw = p[src + "attn.qkv.weight"].detach().to(dev, torch.float32).reshape(3, Hh, hd, Cv)
                qb = p[src + "attn.qkv.bias"].detach().to(dev, torch.float32).reshape(3, Hh, hd)
                qwp, qbp = torch.zeros(3, Hh, 64, Cv, device=dev), torch.zeros(3, Hh, 64, device=dev)
                qwp[:, :, :hd], qbp[:, :, :hd] = qw, qb
                if cfg.mem_pos_enc:
                    # RoPE on the 48-wide heads (mem_pos_enc with use_feat): q and k rows go to the slots the 64-wide rotary pairing
                    # expects (narrow_head_slots); v keeps the plain zero padding (it meets the padded output projection)
                    slots = narrow_head_slots(hd).to(dev)
                    qwp[:2], qbp[:2] = 0.0, 0.0
                    qwp[:2, :, slots], qbp[:2, :, slots] = qw[:2], qb[:2]
                fold("val%d.qkv" % i, qwp.reshape(3 * Hh * 64, Cv), qbp.reshape(-1), src + "norm1")
                pw = p[src + "attn.proj.weight"].detach().to(dev, torch.float32).reshape(Cv, Hh, hd)
                pwp = torch.zeros(Cv, Hh, 64, device=dev)
                pwp[:, :, :hd] = pw
                w["val%d.proj.w" % i], w["val%d.proj.b" % i] = mat(pwp.reshape(Cv, Hh * 64)), vec(p[src + "attn.proj.bias"])
                w["val%d.fc2.w" % i], w["val%d.fc2.b" % i] = mat(p[src + "mlp.fc2.weight"]), vec(p[src + "mlp.fc2.bias"])
            else:
                block("val%d." % i, src)
            fold("val%d.fc1" % i, p[src + "mlp.fc1.weight"], p[src + "mlp.fc1.bias"], src + "norm2")
        for n, s in (("enc_norm", "dust3r.enc_norm"), ("dec_norm", "dust3r.dec_norm"),
                     ("norm_q", "norm_q"), ("norm_k", "norm_k"), ("norm_v", "norm_v")):
            w[n + ".w"], w[n + ".b"] = vec(p[s + ".weight"]), vec(p[s + ".bias"])
        w["dec_embed.w"], w["dec_embed.b"] = mat(p["dust3r.decoder_embed.weight"]), vec(p["dust3r.decoder_embed.bias"])
        fold("value_out", p["value_out.weight"], p["value_out.bias"], "value_norm")
        for side, name in ((1, "dec_blocks"), (2, "dec_blocks2")):
            for i in range(cfg.dec_depth):
                src, dst = "dust3r.%s.%d." % (name, i), "dec%d_%d." % (side, i)
                block(dst, src)
                fold(dst + "fc1", p[src + "mlp.fc1.weight"], p[src + "mlp.fc1.bias"], src + "norm3")
                ca = src + "cross_attn."
                fold(dst + "cq", p[ca + "projq.weight"], p[ca + "projq.bias"], src + "norm2")
                # projk and projv read the same input (norm_y of the other side): one GEMM, N = 2*768
                fold(dst + "ckv", torch.cat((p[ca + "projk.weight"].detach(), p[ca + "projv.weight"].detach()), 0),
                     torch.cat((p[ca + "projk.bias"].detach(), p[ca + "projv.bias"].detach()), 0), src + "norm_y")
                w[dst + "cproj.w"], w[dst + "cproj.b"] = mat(p[ca + "proj.weight"]), vec(p[ca + "proj.bias"])
        for h in (1, 2):
            s, d_ = "attn_head_%d." % h, "key%d." % h
            w[d_ + "0.w"], w[d_ + "0.b"] = mat(p[s + "0.weight"]), vec(p[s + "0.bias"])
            w[d_ + "2.w"], w[d_ + "2.b"] = mat(p[s + "2.weight"]), vec(p[s + "2.bias"])
            s, d_ = "dust3r.downstream_head%d.dpt." % h, "dpt%d." % h
            a = s + "act_postprocess."
            for i in range(4):
                w[d_ + "pp%d.w" % i], w[d_ + "pp%d.b" % i] = mat(p[a + "%d.0.weight" % i]), vec(p[a + "%d.0.bias" % i])
                w[d_ + "rn%d.w" % i] = conv3(p[s + "scratch.layer%d_rn.weight" % (i + 1)])
            w[d_ + "pp0t.w"], w[d_ + "pp0t.b"] = convt(p[a + "0.1.weight"]), vec(p[a + "0.1.bias"])
            if wdt == torch.bfloat16:
                # bf16 mode: the 96-channel map of act_postprocess[0] is carried with 128 channels (32 zero channels: zero rows of
                # the transposed convolution, zero bias, zero input columns of layer1_rn -- the same sums), so that layer1_rn's
                # K = 9 x 128 is whole 64-wide k-blocks and runs on the LDS-tiled convolution instead of the general loader
                wt = p[a + "0.1.weight"].detach().to(dev, torch.float32)                        # [Cin, Cout, 4, 4]
                wt = torch.cat((wt, torch.zeros(wt.shape[0], 32, 4, 4, device=dev)), 1)
                w[d_ + "pp0t.w"] = convt(wt)
                w[d_ + "pp0t.b"] = torch.cat((vec(p[a + "0.1.bias"]), torch.zeros(32, device=dev))).contiguous()
                wr = p[s + "scratch.layer1_rn.weight"].detach().to(dev, torch.float32)          # [256, 96, 3, 3]
                w[d_ + "rn0.w"] = conv3(torch.cat((wr, torch.zeros(wr.shape[0], 32, 3, 3, device=dev)), 1))
            w[d_ + "pp1t.w"], w[d_ + "pp1t.b"] = convt(p[a + "1.1.weight"]), vec(p[a + "1.1.bias"])
            w[d_ + "pp3c.w"], w[d_ + "pp3c.b"] = conv3(p[a + "3.1.weight"]), vec(p[a + "3.1.bias"])
            for r in (1, 2, 3, 4):
                rs, rd = s + "scratch.refinenet%d." % r, d_ + "ref%d." % r
                w[rd + "out.w"], w[rd + "out.b"] = mat(p[rs + "out_conv.weight"]), vec(p[rs + "out_conv.bias"])
                for u in (1, 2):
                    for c in (1, 2):
                        w[rd + "u%dc%d.w" % (u, c)] = conv3(p[rs + "resConfUnit%d.conv%d.weight" % (u, c)])
                        w[rd + "u%dc%d.b" % (u, c)] = vec(p[rs + "resConfUnit%d.conv%d.bias" % (u, c)])
            w[d_ + "h0.w"], w[d_ + "h0.b"] = conv3(p[s + "head.0.weight"]), vec(p[s + "head.0.bias"])
            w[d_ + "h2.w"], w[d_ + "h2.b"] = conv3(p[s + "head.2.weight"]), vec(p[s + "head.2.bias"])
            w[d_ + "h4.w"] = p[s + "head.4.weight"].detach().to(dev, torch.float32).reshape(4, -1).contiguous()
            w[d_ + "h4.b"] = vec(p[s + "head.4.bias"])

        if self.packed_attn:
            # grouped launches: the two decoder sides (and the two key MLPs) run as problems 0 / 1 of ONE launch per op
            def stack(dst, a, b, names):
                for nm in names:
                    x, y = w[a + nm], w[b + nm]
                    w[dst + nm] = ops.PackedWeightGroup([x, y]) if isinstance(x, ops.PackedWeight) else torch.stack((x, y)).contiguous()
            for i in range(cfg.dec_depth):
                stack("decg_%d." % i, "dec1_%d." % i, "dec2_%d." % i,
                      ["qkv.w", "qkv.b", "qkv.s", "proj.w", "proj.b", "cq.w", "cq.b", "cq.s", "ckv.w", "ckv.b", "ckv.s",
                       "cproj.w", "cproj.b", "fc1.w", "fc1.b", "fc1.s", "fc2.w", "fc2.b"])
            stack("keyg.", "key1.", "key2.", ["0.w", "0.b", "2.w", "2.b"])

    def weight_bytes(self):
        return sum((t.data if isinstance(t, ops.PackedWeight) else t).numel() * t.element_size() for t in self.w.values())

    def wsp(self, name, rows, K):
        """fragment-order activation buffer [rows, K] in the activation dtype (GEMM a_packed operand)"""
        key = ("packed", name, rows, K, self.adt)
        t = self._ws.get(key)
        if t is None:
            t = ops.PackedAct(rows, K, self.adt, self.device, data=self._alloc(ops.packed_shape(rows, K, self.adt), self.adt, zero=True))
            self._ws[key] = t
        return t

    # ------------------------------------------------------------------ workspace
    ARENA_CHUNK = 128 << 20        # bytes per arena chunk (one device allocation)
    ARENA_DIRECT = 32 << 20        # requests from this size on get an allocation of their own

    def _alloc(self, shape, dtype, zero=False):
        """Workspace memory comes out of a few large chunks (bump allocation, 256-byte aligned, never freed: the buffers are static --
        captured hipGraphs hold their addresses -- and live as long as the engine).  A geometry's first forward creates ~150
        workspaces; as ~150 device allocations that was 25 ms of the 69 ms a first 10-frame call took (tools/cold_start.py)."""
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * dtype.itemsize
        if nbytes >= self.ARENA_DIRECT or nbytes == 0:
            return (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
        off = (self._arena_off + 255) // 256 * 256
        if self._arena is None or off + nbytes > self._arena.numel():
            self._arena = torch.empty(self.ARENA_CHUNK, dtype=torch.uint8, device=self.device)
            self._arenas.append(self._arena)
            off = 0
        t = self._arena[off:off + nbytes].view(dtype).view(tuple(shape))
        self._arena_off = off + nbytes
        if zero:
            t.zero_()
        return t

    def ws(self, name, shape, dtype=torch.float32, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._ws.get(key)
        if t is None:
            t = self._alloc(shape, dtype, zero)
            self._ws[key] = t
        return t

    def positions(self, B, nh, nw):
        """PositionGetter (croco/models/blocks.py:195-207): int32 [B*nh*nw, 2] (y, x) for the kernels,
        plus the int64 [B, P, 2] tensor the reference API hands around."""
        key = (B, nh, nw)
        if key not in self._pos_cache:
            ys, xs = torch.meshgrid(torch.arange(nh), torch.arange(nw), indexing="ij")
            pos = torch.stack((ys.reshape(-1), xs.reshape(-1)), -1)[None].expand(B, -1, -1).contiguous()
            if max(nh, nw) > self.max_pos:
                raise ValueError("token grid %dx%d exceeds the %d positions of the RoPE tables" % (nh, nw, self.max_pos))
            self._pos_cache[key] = (pos.to(self.device), pos.reshape(-1, 2).to(torch.int32).to(self.device).contiguous(),
                                    torch.zeros(B * nh * nw, 2, dtype=torch.int32, device=self.device))
        return self._pos_cache[key]

    # ------------------------------------------------------------------ transformer pieces
    # Activations that only feed a GEMM (LayerNorm outputs, attention outputs, GELU outputs) are stored in `adt`
    # (bf16 in bf16 mode: exactly the rounding the MFMA operand conversion would apply on load, at half the traffic);
    # the residual stream, LayerNorm statistics and everything the API returns stay fp32.
    def _attn_core(self, xp, st, R, B, P, C, heads, pre, pos32, ao, tag="", head_dim=64, rope_tab=None):
        """norm1 (folded) + qkv projection with fused bias + 2-D RoPE + per-head V^T store, then softmax(qk^T/sqrt(d))v
        (croco/models/blocks.py:94-109, 128).  xp/st: fragment-order copy and row-statistics partials of the stream x.
        The kernels keep 64-wide heads: narrower ones (head_dim < 64: the 16 x 48 heads of the use_feat value encoder) run on
        weights zero-padded to 64 per head at pack time -- q.k is unchanged by zero columns, the padded v columns come out 0
        and meet zero columns of the padded output projection; only the softmax scale carries the true head_dim.
        A = heads * 64 is the width of q / k / v and of `ao`, C the width of the stream."""
        w = self.w
        A = heads * 64
        npad = (P + 63) // 64 * 64
        scale = head_dim ** -0.5
        cos, sin = rope_tab or (self.cos, self.sin)
        ln = ops.LnFold(st, C, w[pre + "qkv.s"], 1e-6)
        if self.packed_attn:
            # fragment-order q/k (+ PV-order V) straight from the projection epilogue; zero pad rows are never written
            qkp = self.ws("qkp" + tag, ops.packed_shape(B * npad, 2 * A, self.wdt), self.wdt, zero=True)
            vtp = self.ws("vtp" + tag, (B * heads * npad * 64,), self.wdt, zero=True)
            ops.proj_rope_vt(xp, w[pre + "qkv.w"], w[pre + "qkv.b"], qkp, 0, vtp, npad, M=R, N=3 * A, K=C, lda=C,
                             rope_cols=2 * A, pos=pos32, cos=cos, sin=sin, tokens=P, heads=heads, qkv_packed=True,
                             ln=ln)
            ops.attention_packed(qkp, 2 * A, 0, npad, qkp, 2 * A, A, npad, vtp, ao, A, B=B, heads=heads, Nq=P, Nk=P,
                                 scale=scale)
            return
        qk = self.ws("qk" + tag, (R, 2 * A), self.wdt)
        vt = self.ws("vt" + tag, (B * heads * 64, npad), self.wdt, zero=True)
        ops.proj_rope_vt(xp, w[pre + "qkv.w"], w[pre + "qkv.b"], qk, 2 * A, vt, npad, M=R, N=3 * A, K=C, lda=C,
                         rope_cols=2 * A, pos=pos32, cos=cos, sin=sin, tokens=P, heads=heads, ln=ln)
        ops.attention(qk, P * 2 * A, 2 * A, qk[:, A:], P * 2 * A, 2 * A, vt, npad, ao, A, B=B, heads=heads, Nq=P, Nk=P,
                      scale=scale)

    def _norm(self, name):
        return (self.w[name + ".w"], self.w[name + ".b"])

    def stats(self, name, R, C):
        """row-statistics partials [R, C/32, 2] of a stream tensor (written by its producer GEMM)"""
        return self.ws(name, (R, C // 32, 2))

    def _update(self, A, pre, R, C, K, x_in, x_out, xp, st):
        """x_out = x_in + A . W^T + b  (attention / MLP output projection, croco/models/blocks.py:110,128-129) with the
        producer-side halves of the next folded LayerNorm: fragment-order copy xp and statistics partials st."""
        w = self.w
        ops.gemm(A, w[pre + ".w"], x_out, M=R, N=C, K=K, lda=K, ldc=C, bias=w[pre + ".b"], res1=x_in, ldr1=C,
                 stats_out=st, c2=xp)

    def _mlp_fc1(self, xp, st, R, C, pre, h):
        """norm (folded) + fc1 + exact-erf GELU (croco/models/blocks.py:74-75), output in fragment order for fc2"""
        w = self.w
        Hd = C * self.cfg.mlp_ratio
        ops.gemm(xp, w[pre + "fc1.w"], h, M=R, N=Hd, K=C, lda=C, ldc=Hd, bias=w[pre + "fc1.b"], act=ACT_GELU,
                 ln=ops.LnFold(st, C, w[pre + "fc1.s"], 1e-6))

    def _block(self, x, xpA, stA, xpB, stB, R, B, P, C, heads, pre, pos32, tag="", last=False, head_dim=64, rope_tab=None):
        """Pre-LN ViT block on the fp32 stream x (croco/models/blocks.py:127-130): 5 launches, no LayerNorm kernel.
        (xpA, stA) describe x on entry and on exit (unless `last`); (xpB, stB) are scratch for the mid-block state."""
        A = heads * 64                                  # attention width (= C unless the heads are zero-padded, _attn_core)
        ao = self.wsp("attn_out" + tag, R, A)
        self._attn_core(xpA, stA, R, B, P, C, heads, pre, pos32, ao, tag=tag, head_dim=head_dim, rope_tab=rope_tab)
        self._update(ao, pre + "proj", R, C, A, x, x, xpB, stB)
        Hd = C * self.cfg.mlp_ratio
        h = self.wsp("mlp_hidden" + tag, R, Hd)
        self._mlp_fc1(xpB, stB, R, C, pre, h)
        self._update(h, pre + "fc2", R, C, Hd, x, x, xpA, stA)

    # ------------------------------------------------------------------ stages
    def _vit(self, col, R, B, P, patch_w, prefix, depth, pos32, tag=""):
        """patch-embed GEMM + `depth` blocks.  Returns (x fp32, fragment-order copy, statistics) of the final stream."""
        cfg, w = self.cfg, self.w
        E = cfg.enc_dim
        K0 = col.K
        x = self.ws("vit_x" + tag, (R, E))
        xpA, xpB = self.wsp("vit_xpA" + tag, R, E), self.wsp("vit_xpB" + tag, R, E)
        stA, stB = self.stats("vit_stA" + tag, R, E), self.stats("vit_stB" + tag, R, E)
        ops.gemm(col, w[patch_w + ".w"], x, M=R, N=E, K=K0, lda=K0, ldc=E, bias=w[patch_w + ".b"], stats_out=stA, c2=xpA)
        for i in range(depth):
            self._block(x, xpA, stA, xpB, stB, R, B, P, E, cfg.enc_heads, prefix + "%d." % i, pos32, tag=tag)
        return x, xpA, stA

    def encode_image(self, img, out=None, tag="", out_packed=None, group_rows=0):
        """dust3r._encode_image (dust3r/model.py:131-154): patch embed -> enc_depth blocks -> enc_norm.
        img fp32 [B,3,H,W] on device -> feat [B,P,1024] (written to `out` if given), pos int64 [B,P,2].
        out_packed (bf16 PackedAct.group): fragment-order copy of the features written by the enc_norm launch, one group per
        group_rows rows (the rows of one frame) -- the A operand of decoder_embed / the key MLPs."""
        self.activate()
        cfg, w = self.cfg, self.w
        B, Cin, H, W_ = img.shape
        p = cfg.patch
        assert Cin == 3 and H % p == 0 and W_ % p == 0, "Input image size is not a multiple of patch size"
        nh, nw = H // p, W_ // p
        P, E = nh * nw, cfg.enc_dim
        R = B * P
        pos64, pos32, _ = self.positions(B, nh, nw)
        col = self.wsp("im2col" + tag, R, 3 * p * p)
        ops.im2col_patch(img, col, B=B, C_=3, H=H, W_=W_, p=p, strides=img.stride())
        if out is None:
            out = torch.empty(B, P, E, device=self.device)
        x, _, _ = self._vit(col, R, B, P, "patch", "enc", cfg.enc_depth, pos32, tag=tag)
        ops.layernorm(x, w["enc_norm.w"], w["enc_norm.b"], 1e-6, out, rows=R, C_=E,     # API-visible: stays a kernel
                      dual=out_packed, group_rows=group_rows)
        return out, pos64

    def side_streams(self):
        """Two side streams: the two decoder branches of a layer (dust3r/model.py:196-198 reads only the PREVIOUS
        layer pair) and the two DPT heads are independent, so they run concurrently; under hipGraph capture the
        fork/join becomes graph edges."""
        if getattr(self, "_streams", None) is None:
            self._streams = {k: torch.cuda.Stream(device=self.device) for k in (1, 2, 3)}
        return self._streams

    def fork(self):
        main = torch.cuda.current_stream()
        st = self.side_streams()
        for k in (1, 2):
            st[k].wait_stream(main)
        return main, st

    def join(self, main, st):
        for k in (1, 2):
            main.wait_stream(st[k])

    def decoder(self, f1, f2, B, nh1, nw1, nh2, nw2, streams=None):
        """dust3r._decoder (dust3r/model.py:186-205).  f1, f2 fp32 [B,P,1024].
        Returns two lists of dec_depth+1 tensors ([B,P,1024] then [B,P,768] ...), last one dec_norm'ed.
        `streams` = {1: stream, 2: stream}: run the two sides concurrently (caller forks / joins)."""
        self.activate()
        cfg, w = self.cfg, self.w
        E, D, Hh = cfg.enc_dim, cfg.dec_dim, cfg.dec_heads
        P1, P2 = nh1 * nw1, nh2 * nw2
        Rs, Ps = {1: B * P1, 2: B * P2}, {1: P1, 2: P2}
        Pmax = max(P1, P2)
        pos = {1: self.positions(B, nh1, nw1)[1], 2: self.positions(B, nh2, nw2)[1]}
        f = {1: f1, 2: f2}
        outs = {1: [f1], 2: [f2]}
        depth = cfg.dec_depth
        cur_stream = torch.cuda.current_stream() if self.device.type == "cuda" else None
        st = streams or {1: cur_stream, 2: cur_stream}
        # per side, double-buffered by layer parity: fragment-order copy + statistics of the layer output
        xp = {s: [self.wsp("dec_xp_%d_%d" % (s, j), Rs[s], D) for j in (0, 1)] for s in (1, 2)}
        st_ = {s: [self.stats("dec_st_%d_%d" % (s, j), Rs[s], D) for j in (0, 1)] for s in (1, 2)}
        prev = {}
        # Cross-stream event waits inside a capture crash hipStreamEndCapture on ROCm 7.2 (tools/
        # probe_multistream_capture.py), plain fork/join does not: so both sides fork from and join to the main stream
        # once per layer (layer i+1 of either side needs layer i of BOTH sides anyway).
        main = cur_stream

        def fork_layer():
            if streams:
                st[1].wait_stream(main)
                st[2].wait_stream(main)

        def join_layer():
            if streams:
                main.wait_stream(st[1])
                main.wait_stream(st[2])

        fork_layer()
        for s in (1, 2):
            with torch.cuda.stream(st[s]):
                prev[s] = self.ws("dec%d_l0" % s, (Rs[s], D))
                # decoder_embed (dust3r/model.py:190-191)
                ops.gemm(f[s], w["dec_embed.w"], prev[s], M=Rs[s], N=D, K=E, lda=E, ldc=D, bias=w["dec_embed.b"],
                         stats_out=st_[s][0], c2=xp[s][0])
        join_layer()
        for i in range(depth):
            cur, nx = i % 2, (i + 1) % 2
            last = i == depth - 1
            new = {}
            fork_layer()
            for s in (1, 2):
                o = 3 - s
                tag = "_s%d" % s
                pre = "dec%d_%d." % (s, i)
                R, P, Ro, Po = Rs[s], Ps[s], Rs[o], Ps[o]
                with torch.cuda.stream(st[s]):
                    x = self.ws("dec%d_l%d" % (s, i + 1), (R, D))
                    xq, stq = self.wsp("dec_xq" + tag, R, D), self.stats("dec_stq" + tag, R, D)
                    # self attention (croco/models/blocks.py:187), norm1 folded into the qkv GEMM
                    ao = self.wsp("attn_out_dec" + tag, R, D)
                    self._attn_core(xp[s][cur], st_[s][cur], R, B, P, D, Hh, pre, pos[s], ao, tag=tag)
                    self._update(ao, pre + "proj", R, D, D, prev[s], x, xq, stq)
                    # cross attention to the other side's previous-layer tokens (:188-189): norm_y folded into the k/v
                    # projection, norm2 into the q projection
                    lnk = ops.LnFold(st_[o][cur], D, w[pre + "ckv.s"], 1e-6)
                    lnq = ops.LnFold(stq, D, w[pre + "cq.s"], 1e-6)
                    if self.packed_attn:
                        npq, npk = (P + 63) // 64 * 64, (Po + 63) // 64 * 64
                        kbuf = self.ws("ckp" + tag, ops.packed_shape(B * npk, D, self.wdt), self.wdt, zero=True)
                        vt = self.ws("cvtp" + tag, (B * Hh * npk * 64,), self.wdt, zero=True)
                        ops.proj_rope_vt(xp[o][cur], w[pre + "ckv.w"], w[pre + "ckv.b"], kbuf, 0, vt, npk, M=Ro, N=2 * D, K=D,
                                         lda=D, rope_cols=D, pos=pos[o], cos=self.cos, sin=self.sin, tokens=Po, heads=Hh,
                                         qkv_packed=True, ln=lnk)
                        if self.fuse_cross_q and P < 512 and D == 768 and ops.LEAN:
                            # (the q projection inside the attention launch, as in decoder_grouped: one group of B images)
                            ops.attention_packed_qproj(xq, stq, w[pre + "cq.w"], w[pre + "cq.s"], w[pre + "cq.b"], pos[s], self.cos, self.sin,
                                                       kbuf, D, 0, npk, vt, ao, D, B=B, heads=Hh, Nq=P, Nk=Po, scale=64 ** -0.5,
                                                       o_group=B, o_group_rows=R, eps=1e-6)
                        else:
                            qbuf = self.ws("cqp" + tag, ops.packed_shape(B * npq, D, self.wdt), self.wdt, zero=True)
                            ops.proj_rope_vt(xq, w[pre + "cq.w"], w[pre + "cq.b"], qbuf, 0, None, npq, M=R, N=D, K=D, lda=D,
                                             rope_cols=D, pos=pos[s], cos=self.cos, sin=self.sin, tokens=P, heads=Hh,
                                             qkv_packed=True, ln=lnq)
                            ops.attention_packed(qbuf, D, 0, npq, kbuf, D, 0, npk, vt, ao, D, B=B, heads=Hh, Nq=P, Nk=Po,
                                                 scale=64 ** -0.5)
                    else:
                        kbuf = self.ws("ck" + tag, (Ro, D), self.wdt)
                        vt = self.ws("cvt" + tag, (B * Hh * 64, (Pmax + 63) // 64 * 64), self.wdt, zero=True)
                        vt_ld = vt.shape[1]
                        ops.proj_rope_vt(xp[o][cur], w[pre + "ckv.w"], w[pre + "ckv.b"], kbuf, D, vt, vt_ld, M=Ro, N=2 * D, K=D,
                                         lda=D, rope_cols=D, pos=pos[o], cos=self.cos, sin=self.sin, tokens=Po, heads=Hh, ln=lnk)
                        qbuf = self.ws("cq" + tag, (R, D), self.wdt)
                        ops.proj_rope_vt(xq, w[pre + "cq.w"], w[pre + "cq.b"], qbuf, D, None, 0, M=R, N=D, K=D, lda=D,
                                         rope_cols=D, pos=pos[s], cos=self.cos, sin=self.sin, tokens=P, heads=Hh, ln=lnq)
                        ops.attention(qbuf, P * D, D, kbuf, Po * D, D, vt, vt_ld, ao, D, B=B, heads=Hh, Nq=P, Nk=Po,
                                      scale=64 ** -0.5)
                    self._update(ao, pre + "cproj", R, D, D, x, x, xq, stq)
                    # MLP (:190), norm3 folded into fc1; fc2 emits the copy + statistics the next layer's GEMMs consume
                    Hd = D * cfg.mlp_ratio
                    h = self.wsp("mlp_hidden_dec" + tag, R, Hd)
                    self._mlp_fc1(xq, stq, R, D, pre, h)
                    if last:
                        self._update(h, pre + "fc2", R, D, Hd, x, x, None, None)
                        normed = self.ws("dec%d_normed" % s, (R, D))
                        ops.layernorm(x, w["dec_norm.w"], w["dec_norm.b"], 1e-6, normed, rows=R, C_=D)   # API-visible
                        new[s] = normed
                    else:
                        self._update(h, pre + "fc2", R, D, Hd, x, x, xp[s][nx], st_[s][nx])
                        new[s] = x
            join_layer()
            for s in (1, 2):
                prev[s] = new[s]
                outs[s].append(new[s].view(B, Ps[s], D))
        return outs[1], outs[2]

    # ------------------------------------------------------------------ grouped decoder (bf16 mode)
    def wspg(self, name, R, K):
        key = ("packed2", name, R, K, self.adt)
        t = self._ws.get(key)
        if t is None:
            t = self._ws[key] = ops.PackedAct.group(2, R, K, self.adt, self.device, alloc=self._alloc)
        return t

    def decoder_grouped(self, f1, f2, B, nh, nw, f1p=None, f2p=None):
        """dust3r._decoder (dust3r/model.py:186-205) with both sides as problems 0 / 1 of one grouped launch per op:
        8 launches per layer on ONE stream instead of 2 x 10 on two streams with a fork/join per layer (the cross-stream
        dependencies cost ~8 us of idle GPU each).  Same kernels, same arithmetic per side as `decoder`."""
        self.activate()
        cfg, w = self.cfg, self.w
        E, D, Hh = cfg.enc_dim, cfg.dec_dim, cfg.dec_heads
        P = nh * nw
        R = B * P
        npad = (P + 63) // 64 * 64
        nt = D // 32
        pos = self.positions(B, nh, nw)[1]
        depth = cfg.dec_depth
        es = 2 if self.adt == torch.bfloat16 else 4
        x = [self.ws("decg_x%d" % l, (2, R, D)) for l in range(depth + 1)]
        xp = [self.wspg("decg_xp%d" % j, R, D) for j in (0, 1)]
        st = [self.ws("decg_st%d" % j, (2, R, nt, 2)) for j in (0, 1)]
        xq, stq = self.wspg("decg_xq", R, D), self.ws("decg_stq", (2, R, nt, 2))
        ao, h = self.wspg("decg_ao", R, D), self.wspg("decg_h", R, D * cfg.mlp_ratio)
        qkp = self.ws("decg_qkp", ops.packed_shape(2 * B * npad, 2 * D, self.wdt), self.wdt, zero=True)
        vtp = self.ws("decg_vtp", (2 * B * Hh * npad * 64,), self.wdt, zero=True)
        cqp = self.ws("decg_cqp", ops.packed_shape(2 * B * npad, D, self.wdt), self.wdt, zero=True)
        ckp = self.ws("decg_ckp", ops.packed_shape(2 * B * npad, D, self.wdt), self.wdt, zero=True)
        cvtp = self.ws("decg_cvtp", (2 * B * Hh * npad * 64,), self.wdt, zero=True)
        sb_st, sb_vt = R * nt * 8, B * Hh * npad * 64 * 2
        Rp = xp[0].rows_pad
        # decoder_embed (dust3r/model.py:190-191): one weight, two inputs that live in different buffers
        if f1p is not None and f2p is not None:
            # fragment-order bf16 copies of both inputs exist (enc_norm / the memory read wrote them): lean instance
            dA = f2p.data_ptr() - f1p.data_ptr()
            assert dA % 16 == 0
            ops.gemm(f1p, w["dec_embed.w"], x[0], M=R, N=D, K=E, lda=E, ldc=D, bias=w["dec_embed.b"], stats_out=st[0], c2=xp[0],
                     batch=2, strideA=dA // 2, strideC=R * D, sb={"stats_out": sb_st, "c2": xp[0].stride * es})
        else:
            dA = f2.data_ptr() - f1.data_ptr()
            assert dA % 16 == 0 and f1.dtype == f2.dtype == torch.float32
            ops.gemm(f1, w["dec_embed.w"], x[0], M=R, N=D, K=E, lda=E, ldc=D, bias=w["dec_embed.b"], stats_out=st[0], c2=xp[0],
                     batch=2, strideA=dA // 4, strideC=R * D, sb={"stats_out": sb_st, "c2": xp[0].stride * es})
        rope = dict(pos=pos, cos=self.cos, sin=self.sin, tokens=P, heads=Hh, qkv_packed=True, batch=2)
        for i in range(depth):
            cur, nx = i % 2, (i + 1) % 2
            g = "decg_%d." % i
            last = i == depth - 1
            xi, xo = x[i], x[i + 1]

            def upd(A, K, pre, res, stats, c2):
                sb = {"bias": D * 4}
                if stats is not None:
                    sb["stats_out"], sb["c2"] = sb_st, c2.stride * es
                ops.gemm(A, w[g + pre + ".w"], xo, M=R, N=D, K=K, lda=K, ldc=D, bias=w[g + pre + ".b"], res1=res, ldr1=D,
                         stats_out=stats, c2=c2, batch=2, strideA=A.stride, strideW=w[g + pre + ".w"].stride, strideC=R * D, sb=sb)

            # self attention (croco/models/blocks.py:187), norm1 folded into the qkv GEMM -- and, in the same launch, the
            # cross attention's k/v projection (:188-189; norm_y folded): both read only the previous layer's tokens.
            # Problem z of the k/v group reads side 1-z -> start at side 1 and step backwards.
            with ops.pair():
                ops.proj_rope_vt(xp[cur], w[g + "qkv.w"], w[g + "qkv.b"], qkp, 0, vtp, npad, M=R, N=3 * D, K=D, lda=D, rope_cols=2 * D,
                                 ln=ops.LnFold(st[cur], D, w[g + "qkv.s"], 1e-6, sb_stats=sb_st, sb_s=3 * D * 4),
                                 strideA=xp[cur].stride, strideW=w[g + "qkv.w"].stride, strideC=B * npad * 2 * D,
                                 sb={"bias": 3 * D * 4, "vt": sb_vt}, **rope)
                ops.proj_rope_vt(xp[cur].at(1), w[g + "ckv.w"], w[g + "ckv.b"], ckp, 0, cvtp, npad, M=R, N=2 * D, K=D, lda=D, rope_cols=D,
                                 ln=ops.LnFold(st[cur][1], D, w[g + "ckv.s"], 1e-6, sb_stats=-sb_st, sb_s=2 * D * 4),
                                 strideA=-xp[cur].stride, strideW=w[g + "ckv.w"].stride, strideC=B * npad * D,
                                 sb={"bias": 2 * D * 4, "vt": sb_vt}, **rope)
            ops.attention_packed(qkp, 2 * D, 0, npad, qkp, 2 * D, D, npad, vtp, ao, D, B=2 * B, heads=Hh, Nq=P, Nk=P,
                                 scale=64 ** -0.5, o_group=B, o_group_rows=Rp)
            upd(ao, D, "proj", xi, stq, xq)
            # cross attention (:188-189): q from this side (norm2); k/v of the OTHER side's previous layer were projected above
            if self.fuse_cross_q and P < 512 and D == 768 and ops.LEAN:
                # the q projection inside the attention launch: one launch less per layer (csrc/attention.hip attn_qproj)
                ops.attention_packed_qproj(xq, stq, w[g + "cq.w"], w[g + "cq.s"], w[g + "cq.b"], pos, self.cos, self.sin, ckp, D, 0, npad, cvtp,
                                           ao, D, B=2 * B, heads=Hh, Nq=P, Nk=P, scale=64 ** -0.5, o_group=B, o_group_rows=Rp, eps=1e-6,
                                           stats_group_stride=sb_st // 4, vec_group_stride=D)
            else:
                ops.proj_rope_vt(xq, w[g + "cq.w"], w[g + "cq.b"], cqp, 0, None, npad, M=R, N=D, K=D, lda=D, rope_cols=D,
                                 ln=ops.LnFold(stq, D, w[g + "cq.s"], 1e-6, sb_stats=sb_st, sb_s=D * 4),
                                 strideA=xq.stride, strideW=w[g + "cq.w"].stride, strideC=B * npad * D, sb={"bias": D * 4}, **rope)
                ops.attention_packed(cqp, D, 0, npad, ckp, D, 0, npad, cvtp, ao, D, B=2 * B, heads=Hh, Nq=P, Nk=P,
                                     scale=64 ** -0.5, o_group=B, o_group_rows=Rp)
            upd(ao, D, "cproj", xo, stq, xq)
            # MLP (:190), norm3 folded into fc1
            Hd = D * cfg.mlp_ratio
            ops.gemm(xq, w[g + "fc1.w"], h, M=R, N=Hd, K=D, lda=D, ldc=Hd, bias=w[g + "fc1.b"], act=ACT_GELU,
                     ln=ops.LnFold(stq, D, w[g + "fc1.s"], 1e-6, sb_stats=sb_st, sb_s=Hd * 4),
                     batch=2, strideA=xq.stride, strideW=w[g + "fc1.w"].stride, strideC=h.stride, sb={"bias": Hd * 4})
            if last:
                upd(h, Hd, "fc2", xo, None, None)
            else:
                upd(h, Hd, "fc2", xo, st[nx], xp[nx])
        normed = self.ws("decg_normed", (2, R, D))
        # API-visible (fp32) + the fragment-order copy the key MLPs read
        ops.layernorm(x[depth], w["dec_norm.w"], w["dec_norm.b"], 1e-6, normed, rows=2 * R, C_=D,
                      dual=self.wspg("decg_normed_packed", R, D) if self.adt == torch.bfloat16 else None, group_rows=R)
        outs = {}
        for s_ in (0, 1):
            outs[s_] = [(f1, f2)[s_]] + [x[l][s_].view(B, P, D) for l in range(1, depth)] + [normed[s_].view(B, P, D)]
        return outs[0], outs[1]

    def key_aux(self, R):
        """fragment-order copy + row-statistics partials of the keys, written by the key MLP's last GEMM: the next
        memory read consumes feat_k2 as a GEMM operand with LN_q folded (problem 1 of the pair)"""
        E = self.cfg.enc_dim
        return self.wspg("keyg_packed", R, E), self.ws("keyg_stats", (2, R, E // 32, 2))

    def encode_feat_keys_grouped(self, feat1, feat2, normed1, normed2, R, out1, out2, feat1p=None, feat2p=None):
        """both key MLPs (spann3r/model.py:299-303) as one grouped launch per layer; normed1/2 = the decoders' last outputs.
        feat1p / feat2p: fragment-order bf16 copies of the features (written by enc_norm): with the packed copy of the decoder
        outputs (decoder_grouped) the first layer runs on a lean split-A instance instead of converting fp32 rows on load."""
        self.activate()
        cfg, w = self.cfg, self.w
        E, D, Kd = cfg.enc_dim, cfg.dec_dim, cfg.key_dim
        dF, dN, dO = feat2.data_ptr() - feat1.data_ptr(), normed2.data_ptr() - normed1.data_ptr(), out2.data_ptr() - out1.data_ptr()
        assert dF % 16 == 0 and dN % 16 == 0 and dO % 16 == 0
        es = 2 if self.adt == torch.bfloat16 else 4
        h = self.wspg("keyg_hidden", R, Kd)
        packed = None
        if feat1p is not None and feat2p is not None and self.adt == torch.bfloat16 and ops.LEAN:
            # a packed split A runs on lean instances only (tiles 42 / 66 / 67; no general tile): ask the library whether it has one
            # for THIS descriptor (SP3_LEAN_BIG=0 with R > 256, or a batch the families do not take, answer no) before taking the layout
            np_ = self.wspg("decg_normed_packed", R, D)
            dFp = feat2p.data_ptr() - feat1p.data_ptr()
            assert dFp % 16 == 0
            packed = dict(A2=np_, lda2=D, K1=E, batch=2, strideA=dFp // 2, strideW=w["keyg.0.w"].stride, strideC=h.stride,
                          sb={"bias": Kd * 4, "A2": np_.stride * 2})
            ok = self._splitA_plan.get(R)
            if ok is None:
                ok = self._splitA_plan[R] = ops.gemm(feat1p, w["keyg.0.w"], h, M=R, N=Kd, K=Kd, lda=E, ldc=Kd, bias=w["keyg.0.b"],
                                                     act=ACT_GELU, plan_only=True, **packed) >= 30
            if not ok:
                packed = None
        if packed is not None:
            ops.gemm(feat1p, w["keyg.0.w"], h, M=R, N=Kd, K=Kd, lda=E, ldc=Kd, bias=w["keyg.0.b"], act=ACT_GELU, **packed)
        else:
            ops.gemm(feat1, w["keyg.0.w"], h, M=R, N=Kd, K=Kd, lda=E, ldc=Kd, bias=w["keyg.0.b"], act=ACT_GELU, A2=normed1, lda2=D, K1=E,
                     batch=2, strideA=dF // 4, strideW=w["keyg.0.w"].stride, strideC=h.stride, sb={"bias": Kd * 4, "A2": dN})
        kp, kst = self.key_aux(R)
        ops.gemm(h, w["keyg.2.w"], out1, M=R, N=E, K=Kd, lda=Kd, ldc=E, bias=w["keyg.2.b"], stats_out=kst, c2=kp,
                 batch=2, strideA=h.stride, strideW=w["keyg.2.w"].stride, strideC=dO // 4,
                 sb={"bias": E * 4, "stats_out": R * (E // 32) * 8, "c2": kp.stride * es})
        return kp.at(1), kst[1]

    def encode_feat_key(self, feat, dec_last, R, num, out, aux=False):
        """spann3r/model.py:299-303: Linear(1792,1792) -> GELU -> Linear(1792,1024) on cat(feat, dec[-1]);
        the concatenation is never materialised (split-A GEMM).  aux: also write the fragment-order copy and the
        row-statistics partials of the key (returned; see key_aux)."""
        self.activate()
        cfg, w = self.cfg, self.w
        E, D, Kd = cfg.enc_dim, cfg.dec_dim, cfg.key_dim
        h = self.wsp("key_hidden%d" % num, R, Kd)
        pre = "key%d." % num
        ops.gemm(feat, w[pre + "0.w"], h, M=R, N=Kd, K=Kd, lda=E, ldc=Kd, bias=w[pre + "0.b"], act=ACT_GELU,
                 A2=dec_last, lda2=D, K1=E)
        if aux:
            kp, kst = self.key_aux(R)
            ops.gemm(h, w[pre + "2.w"], out, M=R, N=E, K=Kd, lda=Kd, ldc=E, bias=w[pre + "2.b"], stats_out=kst[num - 1],
                     c2=kp.at(num - 1))
            return kp.at(num - 1), kst[num - 1]
        ops.gemm(h, w[pre + "2.w"], out, M=R, N=E, K=Kd, lda=Kd, ldc=E, bias=w[pre + "2.b"])
        return out

    def _conv_ws(self, head):
        """split-K scratch of one DPT head's small-map convolutions (one per head: the heads may run concurrently);
        sized for the largest map that is ever split: 8 partial copies of <= 256 x 768 or 2 of 1024 x 256 outputs per image"""
        return self.ws("conv_splitk_ws" + str(head), (1 << 23,))

    def _rcu(self, x, pre, B, H, W_, out, extra_res=None, tag=""):
        """ResidualConvUnit_custom (croco/models/dpt_block.py:120-142): conv2(relu(conv1(relu(x)))) + x [+ extra]."""
        w, F = self.w, self.cfg.dpt_feat
        t = self.ws("rcu_tmp" + tag, (B * H * W_, F), self.mdt)
        sk = self._conv_ws(tag[:1])
        ops.conv3x3(x, w[pre + "c1.w"], t, B=B, H=H, W_=W_, Cin=F, Cout=F, bias=w[pre + "c1.b"], relu_in=True, act=ACT_RELU,
                    splitk_ws=sk)
        ops.conv3x3(t, w[pre + "c2.w"], out, B=B, H=H, W_=W_, Cin=F, Cout=F, bias=w[pre + "c2.b"], res1=x, res2=extra_res,
                    splitk_ws=sk)
        return out

    def _fusion(self, pre, B, H, W_, x0, x1, tag, crop=None):
        """FeatureFusionBlock_custom (croco/models/dpt_block.py:190-218).  The 1x1 out_conv is applied BEFORE the
        bilinear x2 (both are linear and the interpolation weights sum to 1, so the two commute exactly in real
        arithmetic): 4x fewer FLOPs in out_conv.  x0/x1 NHWC [B,H,W,256]; returns ([B,2H,2W,256] or cropped, OH, OW)."""
        w, F = self.w, self.cfg.dpt_feat
        M = B * H * W_
        cur = x0
        if x1 is not None:
            s = self.ws("fus_sum_" + tag, (M, F), self.mdt)
            self._rcu(x1, pre + "u1", B, H, W_, s, extra_res=x0, tag=tag)
            cur = s
        r = self.ws("fus_rcu2_" + tag, (M, F), self.mdt)
        self._rcu(cur, pre + "u2", B, H, W_, r, tag=tag)
        oc = self.ws("fus_oc_" + tag, (M, F), self.mdt)
        ops.gemm(r, w[pre + "out.w"], oc, M=M, N=F, K=F, lda=F, ldc=F, bias=w[pre + "out.b"])
        OH, OW = (2 * H, 2 * W_) if crop is None else crop
        up = self.ws("fus_up_" + tag, (B * OH * OW, F), self.mdt)
        ops.upsample2x(oc, up, B=B, H=H, W_=W_, C_=F, outH=OH, outW=OW)
        return up, OH, OW

    def dpt_head(self, dec, B, nh, nw, num, want_raw=False):
        """DPTOutputAdapter_fix.forward + postprocess (dust3r/heads/dpt_head.py:34-65, postprocess.py:10-58).
        dec: list of dec_depth+1 token tensors; tokens ARE the NHWC map [B,nh,nw,C]."""
        self.activate()
        cfg, w = self.cfg, self.w
        F, Lc = cfg.dpt_feat, cfg.dpt_last
        pre = "dpt%d." % num
        R = B * nh * nw
        hk = cfg.hooks
        dims = (cfg.enc_dim, cfg.dec_dim, cfg.dec_dim, cfg.dec_dim)
        ld = (96, 192, 384, 768)
        mdt = self.mdt
        t = [self.ws("dpt%d_pp%d" % (num, i), (R, ld[i]), mdt) for i in range(4)]
        pairable = len({dec[hk[i]].dtype for i in range(4)}) == 1          # one kernel instance per pair (same operand dtypes)
        # the four hooks are independent: two launches of two differently shaped problems each (sp3_gemm2)
        import contextlib
        pair = ops.pair if pairable else contextlib.nullcontext
        for i0 in (0, 2):
            with pair():
                for i in (i0, i0 + 1):
                    ops.gemm(dec[hk[i]], w[pre + "pp%d.w" % i], t[i], M=R, N=ld[i], K=dims[i], lda=dims[i], ldc=ld[i], bias=w[pre + "pp%d.b" % i])
        # act_postprocess tails (croco/models/dpt_block.py:356-410)
        c0 = 128 if self.wdt == torch.bfloat16 else ld[0]         # (bf16 mode: 32 zero channels ride along, see _pack)
        l0 = self.ws("dpt%d_l0" % num, (B * 16 * nh * nw, c0), mdt)
        l1 = self.ws("dpt%d_l1" % num, (B * 4 * nh * nw, ld[1]), mdt)
        with pair():
            ops.conv_transpose_ks(t[0], w[pre + "pp0t.w"], l0, B=B, H=nh, W_=nw, Cin=ld[0], Cout=c0, ks=4, bias=w[pre + "pp0t.b"])
            ops.conv_transpose_ks(t[1], w[pre + "pp1t.w"], l1, B=B, H=nh, W_=nw, Cin=ld[1], Cout=ld[1], ks=2, bias=w[pre + "pp1t.b"])
        l2 = t[2]
        h3, w3 = (nh - 1) // 2 + 1, (nw - 1) // 2 + 1
        l3 = self.ws("dpt%d_l3" % num, (B * h3 * w3, ld[3]), mdt)
        sk = self._conv_ws(num)
        ops.conv3x3(t[3], w[pre + "pp3c.w"], l3, B=B, H=nh, W_=nw, Cin=ld[3], Cout=ld[3], stride=2, bias=w[pre + "pp3c.b"],
                    splitk_ws=sk)
        # scratch.layer_rn (3x3, no bias) -> 256 channels
        geo = ((4 * nh, 4 * nw), (2 * nh, 2 * nw), (nh, nw), (h3, w3))
        rn = []
        for i, src in enumerate((l0, l1, l2, l3)):
            Hh, Ww = geo[i]
            o = self.ws("dpt%d_rn%d" % (num, i), (B * Hh * Ww, F), mdt)
            ops.conv3x3(src, w[pre + "rn%d.w" % i], o, B=B, H=Hh, W_=Ww, Cin=(c0 if i == 0 else ld[i]), Cout=F, splitk_ws=sk)
            rn.append(o)
        # refinenets; path_4 is cropped to layer 3's size (dust3r/heads/dpt_head.py:57)
        p4, H4, W4 = self._fusion(pre + "ref4.", B, h3, w3, rn[3], None, "%d_4" % num, crop=(nh, nw))
        p3, H3, W3 = self._fusion(pre + "ref3.", B, H4, W4, p4, rn[2], "%d_3" % num)
        p2, H2, W2 = self._fusion(pre + "ref2.", B, H3, W3, p3, rn[1], "%d_2" % num)
        p1, H1, W1 = self._fusion(pre + "ref1.", B, H2, W2, p2, rn[0], "%d_1" % num)
        # head: conv3x3(256->128) -> x2 -> conv3x3(128->128) + ReLU -> 1x1(128->4) + postprocess
        a = self.ws("dpt%d_h0" % num, (B * H1 * W1, Lc), mdt)
        ops.conv3x3(p1, w[pre + "h0.w"], a, B=B, H=H1, W_=W1, Cin=F, Cout=Lc, bias=w[pre + "h0.b"])
        OH, OW = 2 * H1, 2 * W1
        u = self.ws("dpt%d_h0up" % num, (B * OH * OW, Lc), mdt)
        ops.upsample2x(a, u, B=B, H=H1, W_=W1, C_=Lc)
        c = self.ws("dpt%d_h2" % num, (B * OH * OW, Lc), mdt)
        ops.conv3x3(u, w[pre + "h2.w"], c, B=B, H=OH, W_=OW, Cin=Lc, Cout=Lc, bias=w[pre + "h2.b"], act=ACT_RELU)
        pts = self.ws("dpt_pts%d" % num, (B, OH, OW, 3))
        conf = self.ws("dpt_conf%d" % num, (B, OH, OW))
        raw = self.ws("dpt_raw%d" % num, (B, OH, OW, 4)) if want_raw else None
        ops.head_final(c, w[pre + "h4.w"], w[pre + "h4.b"], B * OH * OW, Lc, pts, conf, raw)
        return pts, conf, raw

    def encode_cur_value_feat(self, tok, out, res, pos32=None):
        """spann3r/model.py:312-314 (use_feat=True): value_out(value_norm(value_encoder(dec1[-1]))) -- 6 blocks of width 768 with
        16 heads of 48 (zero-padded to 64, _attn_core); no RoPE, or with mem_pos_enc RoPE2D on the 48-wide heads (pos32 = the
        tokens' int32 (y, x) positions: narrow_head_slots / _rope_tables_narrow); `res` (feat_k1) is added by the finishing GEMM if
        given.  tok fp32 [B,P,768]."""
        self.activate()
        cfg, w = self.cfg, self.w
        B, P, Cv = tok.shape
        assert Cv == cfg.val_dim
        R, E = B * P, cfg.enc_dim
        zero_pos = self.ws("valf_zero_pos", (R, 2), torch.int32, zero=True)     # rope=None (:232-234): all-zero positions = identity
        rope_tab = None
        if cfg.mem_pos_enc:
            assert pos32 is not None and self.rope_narrow is not None
            zero_pos, rope_tab = pos32, self.rope_narrow
        x = self.ws("valf_x", (R, Cv))
        xpA, xpB = self.wsp("valf_xpA", R, Cv), self.wsp("valf_xpB", R, Cv)
        stA, stB = self.stats("valf_stA", R, Cv), self.stats("valf_stB", R, Cv)
        ops.copy2d(tok.reshape(R, Cv), Cv, x, Cv, R, Cv)                       # the stream is updated in place: keep dec1[-1] intact
        ops.pack_stats(x, xpA, stA, rows=R, C_=Cv)
        for i in range(cfg.val_depth):
            self._block(x, xpA, stA, xpB, stB, R, B, P, Cv, cfg.enc_heads, "val%d." % i, zero_pos, tag="_valf",
                        head_dim=Cv // cfg.enc_heads, rope_tab=rope_tab)
        ops.gemm(xpA, w["value_out.w"], out, M=R, N=E, K=Cv, lda=Cv, ldc=E, bias=w["value_out.b"], res1=res, ldr1=E,
                 ln=ops.LnFold(stA, Cv, w["value_out.s"], 1e-6))
        return out

    def encode_cur_value(self, pts3d, out, res):
        """spann3r/model.py:305-320 (use_feat=False): pos_patch_embed(pts3d as a 3-channel image) -> 6 blocks without
        RoPE -> value_norm -> value_out; `res` (feat_k1) is added by the finishing kernel (:519/:521 `cur_v+feat_k1`)
        only if given.  pts3d fp32 [B,H,W,3] (any strides)."""
        self.activate()
        cfg, w = self.cfg, self.w
        B, H, W_, _ = pts3d.shape
        p, E = cfg.patch, cfg.enc_dim
        nh, nw = H // p, W_ // p
        P = nh * nw
        R = B * P
        _, pos32, zero_pos = self.positions(B, nh, nw)
        col = self.wsp("im2col_val", R, 3 * p * p)
        sb, sy, sx, sc = pts3d.stride()
        ops.im2col_patch(pts3d, col, B=B, C_=3, H=H, W_=W_, p=p, strides=(sb, sc, sy, sx))
        # rope=None in the reference unless mem_pos_enc (spann3r/model.py:232-234): all-zero positions make the fused RoPE the identity
        x, xp, st = self._vit(col, R, B, P, "pospatch", "val", cfg.val_depth, pos32 if cfg.mem_pos_enc else zero_pos, tag="_val")
        # value_norm folded into value_out; + feat_k1 in the epilogue
        ops.gemm(xp, w["value_out.w"], out, M=R, N=E, K=E, lda=E, ldc=E, bias=w["value_out.b"], res1=res, ldr1=E,
                 ln=ops.LnFold(st, E, w["value_out.s"], 1e-6))
        return out
