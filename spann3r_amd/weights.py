"""State-dict contract and a platform-independent synthetic
weight generator.

No checkpoint ships with the reference and there is no network, so every parity
fixture uses seeded synthetic weights.  The generator below is a pure integer hash
(splitmix64 over the element index) mapped to a uniform range: no libm call, no
torch/numpy RNG stream, so the very same tensors come out in this container and on
the GPU box -- which the golden fixtures under tests/golden/ rely on.
"""
from collections import OrderedDict
import zlib

import numpy as np
import torch

from .config import Spann3RConfig, FULL


# --------------------------------------------------------------------------- spec
def _block_spec(prefix, dim, hidden, out):
    out[prefix + "norm1.weight"] = (dim,)
    out[prefix + "norm1.bias"] = (dim,)
    out[prefix + "attn.qkv.weight"] = (3 * dim, dim)
    out[prefix + "attn.qkv.bias"] = (3 * dim,)
    out[prefix + "attn.proj.weight"] = (dim, dim)
    out[prefix + "attn.proj.bias"] = (dim,)
    out[prefix + "norm2.weight"] = (dim,)
    out[prefix + "norm2.bias"] = (dim,)
    out[prefix + "mlp.fc1.weight"] = (hidden, dim)
    out[prefix + "mlp.fc1.bias"] = (hidden,)
    out[prefix + "mlp.fc2.weight"] = (dim, hidden)
    out[prefix + "mlp.fc2.bias"] = (dim,)


def _dec_block_spec(prefix, dim, hidden, out):
    # key order follows nn.Module registration order of the reference DecoderBlock
    # (croco/models/blocks.py:173-184): norm1, attn, cross_attn, norm2, norm3, mlp, norm_y
    out[prefix + "norm1.weight"] = (dim,)
    out[prefix + "norm1.bias"] = (dim,)
    out[prefix + "attn.qkv.weight"] = (3 * dim, dim)
    out[prefix + "attn.qkv.bias"] = (3 * dim,)
    out[prefix + "attn.proj.weight"] = (dim, dim)
    out[prefix + "attn.proj.bias"] = (dim,)
    for p in ("projq", "projk", "projv", "proj"):
        out[prefix + "cross_attn.%s.weight" % p] = (dim, dim)
        out[prefix + "cross_attn.%s.bias" % p] = (dim,)
    out[prefix + "norm2.weight"] = (dim,)
    out[prefix + "norm2.bias"] = (dim,)
    out[prefix + "norm3.weight"] = (dim,)
    out[prefix + "norm3.bias"] = (dim,)
    out[prefix + "mlp.fc1.weight"] = (hidden, dim)
    out[prefix + "mlp.fc1.bias"] = (hidden,)
    out[prefix + "mlp.fc2.weight"] = (dim, hidden)
    out[prefix + "mlp.fc2.bias"] = (dim,)
    out[prefix + "norm_y.weight"] = (dim,)
    out[prefix + "norm_y.bias"] = (dim,)


def _dpt_spec(prefix, cfg, out):
    ed, dd, F, L = cfg.enc_dim, cfg.dec_dim, cfg.dpt_feat, cfg.dpt_last
    ld = (96, 192, 384, 768)
    # scratch.layer{1..4}_rn then the ModuleList aliases scratch.layer_rn.{0..3}
    for i, c in enumerate(ld):
        out[prefix + "scratch.layer%d_rn.weight" % (i + 1)] = (F, c, 3, 3)
    for i, c in enumerate(ld):
        out[prefix + "scratch.layer_rn.%d.weight" % i] = (F, c, 3, 3)   # alias of the above
    for r in (1, 2, 3, 4):
        p = prefix + "scratch.refinenet%d." % r
        out[p + "out_conv.weight"] = (F, F, 1, 1)
        out[p + "out_conv.bias"] = (F,)
        for u in ("resConfUnit1", "resConfUnit2"):
            for c in ("conv1", "conv2"):
                out[p + "%s.%s.weight" % (u, c)] = (F, F, 3, 3)
                out[p + "%s.%s.bias" % (u, c)] = (F,)
    out[prefix + "head.0.weight"] = (L, F, 3, 3)
    out[prefix + "head.0.bias"] = (L,)
    out[prefix + "head.2.weight"] = (L, L, 3, 3)
    out[prefix + "head.2.bias"] = (L,)
    out[prefix + "head.4.weight"] = (4, L, 1, 1)
    out[prefix + "head.4.bias"] = (4,)
    dims = (ed, dd, dd, dd)
    out[prefix + "act_postprocess.0.0.weight"] = (ld[0], dims[0], 1, 1)
    out[prefix + "act_postprocess.0.0.bias"] = (ld[0],)
    out[prefix + "act_postprocess.0.1.weight"] = (ld[0], ld[0], 4, 4)      # ConvTranspose2d (in, out, kh, kw)
    out[prefix + "act_postprocess.0.1.bias"] = (ld[0],)
    out[prefix + "act_postprocess.1.0.weight"] = (ld[1], dims[1], 1, 1)
    out[prefix + "act_postprocess.1.0.bias"] = (ld[1],)
    out[prefix + "act_postprocess.1.1.weight"] = (ld[1], ld[1], 2, 2)      # ConvTranspose2d
    out[prefix + "act_postprocess.1.1.bias"] = (ld[1],)
    out[prefix + "act_postprocess.2.0.weight"] = (ld[2], dims[2], 1, 1)
    out[prefix + "act_postprocess.2.0.bias"] = (ld[2],)
    out[prefix + "act_postprocess.3.0.weight"] = (ld[3], dims[3], 1, 1)
    out[prefix + "act_postprocess.3.0.bias"] = (ld[3],)
    out[prefix + "act_postprocess.3.1.weight"] = (ld[3], ld[3], 3, 3)      # Conv2d stride 2
    out[prefix + "act_postprocess.3.1.bias"] = (ld[3],)


def param_spec(cfg: Spann3RConfig = FULL) -> "OrderedDict[str, tuple]":
    """name -> shape for every state-dict key of spann3r.model.Spann3R."""
    o = OrderedDict()
    E, D = cfg.enc_dim, cfg.dec_dim
    o["dust3r.mask_token"] = (1, 1, D)
    o["dust3r.patch_embed.proj.weight"] = (E, 3, cfg.patch, cfg.patch)
    o["dust3r.patch_embed.proj.bias"] = (E,)
    for i in range(cfg.enc_depth):
        _block_spec("dust3r.enc_blocks.%d." % i, E, E * cfg.mlp_ratio, o)
    o["dust3r.enc_norm.weight"] = (E,)
    o["dust3r.enc_norm.bias"] = (E,)
    o["dust3r.decoder_embed.weight"] = (D, E)
    o["dust3r.decoder_embed.bias"] = (D,)
    for i in range(cfg.dec_depth):
        _dec_block_spec("dust3r.dec_blocks.%d." % i, D, D * cfg.mlp_ratio, o)
    o["dust3r.dec_norm.weight"] = (D,)
    o["dust3r.dec_norm.bias"] = (D,)
    for i in range(cfg.dec_depth):
        _dec_block_spec("dust3r.dec_blocks2.%d." % i, D, D * cfg.mlp_ratio, o)
    _dpt_spec("dust3r.downstream_head1.dpt.", cfg, o)
    _dpt_spec("dust3r.downstream_head2.dpt.", cfg, o)
    for i in range(cfg.val_depth):
        _block_spec("value_encoder.%d." % i, cfg.val_dim, cfg.val_dim * cfg.mlp_ratio, o)
    o["value_norm.weight"] = (cfg.val_dim,)
    o["value_norm.bias"] = (cfg.val_dim,)
    o["value_out.weight"] = (E, cfg.val_dim)
    o["value_out.bias"] = (E,)
    if not cfg.use_feat:                     # spann3r/model.py:239-241: no pos_patch_embed with use_feat
        o["pos_patch_embed.proj.weight"] = (E, 3, cfg.patch, cfg.patch)
        o["pos_patch_embed.proj.bias"] = (E,)
    for n in ("norm_q", "norm_k", "norm_v"):
        o[n + ".weight"] = (E,)
        o[n + ".bias"] = (E,)
    for h in (1, 2):
        o["attn_head_%d.0.weight" % h] = (cfg.key_dim, cfg.key_dim)
        o["attn_head_%d.0.bias" % h] = (cfg.key_dim,)
        o["attn_head_%d.2.weight" % h] = (E, cfg.key_dim)
        o["attn_head_%d.2.bias" % h] = (E,)
    return o


def alias_of(key: str):
    """scratch.layer_rn.{i}.weight is the same Parameter as scratch.layer{i+1}_rn.weight
    (croco/models/dpt_block.py:69-74)."""
    marker = ".scratch.layer_rn."
    if marker in key:
        head, tail = key.split(marker)
        idx = int(tail.split(".")[0])
        return "%s.scratch.layer%d_rn.weight" % (head, idx + 1)
    return None


# --------------------------------------------------------------------------- hash RNG
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def hash_uniform(n: int, stream: int) -> np.ndarray:
    """n float32 values in [-1, 1): splitmix64(stream * 2^32 + index), top 24 bits.
    Integer-only until the final exact int->float scaling."""
    out = np.empty(n, dtype=np.float32)
    base = np.uint64((stream & 0xFFFFFFFF) << 32)
    chunk = 1 << 22
    with np.errstate(over="ignore"):
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            z = (np.arange(s, e, dtype=np.uint64) + base + np.uint64(1)) * _GOLD
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
            u = (z >> np.uint64(40)).astype(np.float32)          # 24 bits, exact in fp32
            out[s:e] = u * np.float32(2.0 ** -23) - np.float32(1.0)
    return out


def _stream_id(seed: int, key: str) -> int:
    return (zlib.crc32(key.encode()) ^ (seed * 0x85EBCA6B)) & 0xFFFFFFFF


def _init_scale(key: str, shape) -> tuple:
    """(centre, half_width) of the uniform distribution for one tensor.
    Chosen so activations stay O(1) through 24+12+6 blocks and attention is peaked
    (not the near-uniform regime plain xavier gives, SURVEY.md §7 'Parity budget')."""
    leaf = key.rsplit(".", 1)[-1]
    is_norm = any(t in key for t in (".norm1.", ".norm2.", ".norm3.", ".norm_y.", "enc_norm.", "dec_norm.",
                                     "value_norm.", "norm_q.", "norm_k.", "norm_v."))
    if is_norm:
        return (1.0, 0.25) if leaf == "weight" else (0.0, 0.1)
    if key.endswith("mask_token"):
        return (0.0, 0.02)
    if leaf == "bias":
        return (0.0, 0.05)
    fan_in = int(np.prod(shape[1:]))
    if "act_postprocess.0.1." in key or "act_postprocess.1.1." in key:
        fan_in = shape[0]            # ConvTranspose2d with k == stride: one tap per output pixel
    gain = 1.0
    # residual-branch output projections are damped so the residual stream does not blow up
    if any(t in key for t in ("attn.proj.", "cross_attn.proj.", "mlp.fc2.", "conv2.")):
        gain = 0.5
    if ".head.4." in key:
        gain = 0.3          # keeps |xyz| ~ 1 so expm1/exp do not amplify round-off in the fixtures
    return (0.0, gain * float(np.sqrt(3.0 / fan_in)))


def synth_state_dict(seed: int = 0, cfg: Spann3RConfig = FULL, dtype=torch.float32):
    """Seeded synthetic state dict with exactly the reference's keys/shapes."""
    sd = OrderedDict()
    for key, shape in param_spec(cfg).items():
        src = alias_of(key)
        if src is not None:
            sd[key] = sd[src]
            continue
        n = int(np.prod(shape))
        centre, half = _init_scale(key, shape)
        u = hash_uniform(n, _stream_id(seed, key))
        t = torch.from_numpy(u).mul_(half).add_(centre).reshape(shape)
        sd[key] = t.to(dtype)
    return sd


def stress_state_dict(seed: int = 7, cfg: Spann3RConfig = FULL):
    """`synth_state_dict` reshaped towards the statistics of a TRAINED checkpoint, the regime the fast parity mode (f32x3) has
    to survive (random-init weights have a tiny dynamic range): per-output-channel scales of every Linear spanning
    two decades (1/11 .. 11, ~log-uniform; attention / MLP output projections 1/8 .. 3 so the residual stream stays finite),
    LayerNorm gains spanning 0.25 .. 4 with non-zero shifts, a few massive-activation channels in the residual streams
    (patch-embed / decoder-embed biases of +-40 on 4 channels, as trained ViTs have), and 3x sharper memory logits (norm_q /
    norm_k gains x 1.75) so that the spatial-memory softmax is close to one-hot.  Integer-hash streams only: every platform
    regenerates the same bits."""
    sd = synth_state_dict(seed, cfg)
    out = OrderedDict()
    for key, t in sd.items():
        src = alias_of(key)
        if src is not None:
            out[key] = out[src]
            continue
        t = t.clone()
        u = lambda n, tag: torch.from_numpy(hash_uniform(n, _stream_id(seed, key + "#" + tag)))      # in [-1, 1)

        def logscale(n, tag, lo, hi):
            """~log-uniform factors in [2^lo, 2^hi) from IEEE mul / add / ldexp only (libm's pow differs in the last bit between
            CPUs, and the fixtures must regenerate bit for bit): (1 + frac) * 2^floor of a uniform exponent"""
            x = (u(n, tag) + 1.0) * (0.5 * (hi - lo)) + lo
            e = torch.floor(x)
            return torch.ldexp(1.0 + (x - e), e.to(torch.int32))
        is_lin = key.endswith(".weight") and t.dim() == 2
        if ".dpt.head.4." in key:
            # the raw pointmap norm r goes through expm1: keep r ~ 1..3 as trained checkpoints do (depths of metres, not of 10^4:
            # at r ~ 10 a 1e-4 error of r alone is a 1e-3 error of the pointmap, whatever the kernels do)
            t *= 0.25
        elif is_lin and any(k in key for k in ("attn.proj", "mlp.fc2", "cross_attn.proj")):
            t *= logscale(t.shape[0], "rows", -3.0, 1.5)[:, None]                       # 1/8 .. ~3
        elif is_lin:
            t *= logscale(t.shape[0], "rows", -3.5, 3.5)[:, None]                       # 1/11 .. 11: two decades
        elif key.endswith(".weight") and t.dim() == 1 and ("norm" in key):
            t = t * logscale(t.shape[0], "gain", -2.0, 2.0)                              # 0.25 .. 4
            if key.startswith("norm_q.") or key.startswith("norm_k."):
                t = t * 1.75                                                             # ~3x sharper memory logits
        elif key.endswith(".bias") and "norm" in key:
            t = t + 0.5 * u(t.shape[0], "shift")
        elif key in ("dust3r.patch_embed.proj.bias", "dust3r.decoder_embed.bias", "pos_patch_embed.proj.bias"):
            idx = (torch.arange(4) * 37 + 11) % t.shape[0]
            t[idx] += torch.tensor([40.0, -40.0, 25.0, -25.0])
        out[key] = t
    return out


def state_dict_fingerprint(sd) -> float:
    """Cheap checksum the fixtures carry so a platform mismatch in the generator is
    diagnosed as such (and not as a kernel bug)."""
    acc = 0.0
    for k in ("dust3r.enc_blocks.0.attn.qkv.weight", "dust3r.dec_blocks2.1.mlp.fc1.weight",
              "value_out.weight", "dust3r.downstream_head2.dpt.head.2.weight"):
        if k in sd:
            acc += float(sd[k].double().abs().sum())
    return acc


def synth_frames(n_frames: int, h: int, w: int, batch: int = 1, seed: int = 1000):
    """frames as demo.py hands them over: list of {'img': [B,3,H,W] in [-1,1]}."""
    frames = []
    for i in range(n_frames):
        u = hash_uniform(batch * 3 * h * w, _stream_id(seed, "frame%d" % i))
        # smooth-ish content: mix of a low-frequency pattern and noise, clipped to ImgNorm range
        img = torch.from_numpy(u).reshape(batch, 3, h, w)
        # low-frequency content from IEEE add/mul only (no libm: bit-identical everywhere)
        yy = (torch.arange(h, dtype=torch.float32) * (2.0 / max(h - 1, 1)) - 1.0).view(1, 1, h, 1)
        xx = (torch.arange(w, dtype=torch.float32) * (2.0 / max(w - 1, 1)) - 1.0).view(1, 1, 1, w)
        sx = xx + 0.125 * (i % 5) - 0.25
        base = 0.9 * sx * (1.0 - sx * sx) * (1.0 - 0.75 * yy * yy) + 0.1 * yy
        frames.append({"img": (0.6 * img + base).clamp(-1, 1).contiguous()})
    return frames


def synth_loss_case(seed, n=4, B=2, H=24, W=32):
    """Seeded inputs of the training criterion: gts (pts3d, valid_mask, camera_pose) and preds_all shaped like
    Spann3R.forward's (res1 has 'pts3d' for the first pair, 'pts3d_in_other_view' afterwards; res2 always the latter)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    gts, preds_all = [], []
    for i in range(n):
        A = rn(B, 3, 3)
        Q, _ = torch.linalg.qr(A)
        pose = torch.eye(4).repeat(B, 1, 1)
        pose[:, :3, :3] = Q
        pose[:, :3, 3] = rn(B, 3) * 0.5
        pts = rn(B, H, W, 3) * 1.5 + torch.tensor([0.0, 0.0, 3.0])
        valid = torch.rand(B, H, W, generator=g) < 0.8
        if i == 1:
            valid[0, :5] = False
        gts.append(dict(pts3d=pts, valid_mask=valid, camera_pose=pose))
    for i in range(n - 1):
        r1 = {("pts3d" if i == 0 else "pts3d_in_other_view"): rn(B, H, W, 3) * 1.2 + torch.tensor([0.0, 0.0, 2.5]),
              "conf": 1 + torch.exp(rn(B, H, W) * 0.7)}
        r2 = {"pts3d_in_other_view": rn(B, H, W, 3) * 1.2 + torch.tensor([0.0, 0.0, 2.5]), "conf": 1 + torch.exp(rn(B, H, W) * 0.7)}
        preds_all.append((r1, r2))
    return gts, preds_all


def synth_pointmaps(seed, B=2, H=48, W=64, focal=(55.0, 83.0)):
    """pointmaps of pinhole cameras with known focals + noise, a few outliers, zeros and negative depths"""
    g = torch.Generator().manual_seed(seed)
    z = 1.0 + 3.0 * torch.rand(B, H, W, generator=g)
    u, v = torch.meshgrid(torch.arange(W).float() - W / 2, torch.arange(H).float() - H / 2, indexing="xy")
    f = torch.tensor(focal[:B]).view(B, 1, 1)
    pts = torch.stack((u * z / f, v * z / f, z), -1) + 0.01 * torch.randn(B, H, W, 3, generator=g)
    pts[0, 3, 5] = 0.0                       # 0/0 -> nan_to_num
    pts[1, 7, 9, 2] = 0.0                    # x/0 -> inf -> 0
    pts[:, ::7, ::5] *= torch.tensor([3.0, -2.0, 1.0])       # outliers
    return pts
