"""CPU oracle for the Spann3R per-frame forward hot path.  TEST INFRASTRUCTURE ONLY.

This is a plain torch-CPU fp32, functional (state-dict in, tensors out) restatement of the
algorithm the reference implements with nn.Modules.  It exists so that the `-m gpu` parity
tests, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg have something to check /
time against on the GPU box, where /root/reference does not exist.  The product path
(spann3r_amd/) never imports it.

Pinning: the reference ships no tests or golden vectors, so the oracle is pinned
against outputs of the *unmodified reference run in the build container*:
tests/golden/make_golden.py imports /root/reference, runs it on seeded synthetic weights and
frames, and commits the dumps under tests/golden/*.npz; tests/test_oracle_vs_golden.py checks
every function below against those dumps.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- primitives
def rope_tables(max_pos: int, half_dim: int, base: float):
    """cos/sin tables of the torch fallback RoPE2D (croco/models/pos_embed.py:118-129):
    inv_freq_i = base^(-2i/half_dim), angle = pos * inv_freq, tables [max_pos, half_dim/2]."""
    inv_freq = 1.0 / (base ** (torch.arange(0, half_dim, 2).float() / half_dim))
    t = torch.arange(max_pos, dtype=torch.float32)
    ang = torch.einsum("i,j->ij", t, inv_freq)
    return ang.cos(), ang.sin()


def rope2d(tokens, pos, base=100.0, fwd=1.0):
    """2-D rotary embedding on tokens [B,H,N,D] with pos [B,N,2] (y,x).
    croco/models/pos_embed.py:142-159 == curope/curope.cpp:11-47:
    D split in 4 quarters [uY|vY|uX|vX]; u' = u cos - v sin, v' = v cos + u sin with
    angle = pos_axis * fwd / base^(i/Q), Q = D/4."""
    B, Hh, N, D = tokens.shape
    Q = D // 4
    cos, sin = rope_tables(int(pos.max()) + 1, D // 2, base)      # [P, Q]
    out = torch.empty_like(tokens)
    for ax in range(2):
        c = cos[pos[:, :, ax]][:, None]                            # [B,1,N,Q]
        s = sin[pos[:, :, ax]][:, None] * fwd
        u = tokens[..., ax * 2 * Q: ax * 2 * Q + Q]
        v = tokens[..., ax * 2 * Q + Q: ax * 2 * Q + 2 * Q]
        out[..., ax * 2 * Q: ax * 2 * Q + Q] = u * c - v * s
        out[..., ax * 2 * Q + Q: ax * 2 * Q + 2 * Q] = v * c + u * s
    return out


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def positions(B, h, w):
    """PositionGetter (croco/models/blocks.py:195-207): (y,x) token coordinates, raster order."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack((ys.reshape(-1), xs.reshape(-1)), -1)[None].expand(B, -1, -1).contiguous()


def self_attention(x, pos, sd, p, heads, base, use_rope=True):
    """croco/models/blocks.py:94-112."""
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B, N, 3, heads, hd)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))       # [B,H,N,hd]
    if use_rope:
        q, k = rope2d(q, pos, base), rope2d(k, pos, base)
    a = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def cross_attention(xq, y, qpos, kpos, sd, p, heads, base):
    """croco/models/blocks.py:149-169 (key == value == norm_y(other side))."""
    B, Nq, C = xq.shape
    Nk = y.shape[1]
    hd = C // heads
    q = F.linear(xq, sd[p + "projq.weight"], sd[p + "projq.bias"]).reshape(B, Nq, heads, hd).transpose(1, 2)
    k = F.linear(y, sd[p + "projk.weight"], sd[p + "projk.bias"]).reshape(B, Nk, heads, hd).transpose(1, 2)
    v = F.linear(y, sd[p + "projv.weight"], sd[p + "projv.bias"]).reshape(B, Nk, heads, hd).transpose(1, 2)
    q, k = rope2d(q, qpos, base), rope2d(k, kpos, base)
    a = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, Nq, C)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def mlp(x, sd, p):
    """croco/models/blocks.py:73-79 (exact-erf GELU)."""
    h = F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    return F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])


def block(x, pos, sd, p, heads, base, use_rope=True, eps=1e-6):
    """Pre-LN ViT block, croco/models/blocks.py:127-130."""
    x = x + self_attention(layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps), pos, sd,
                           p + "attn.", heads, base, use_rope)
    return x + mlp(layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps), sd, p + "mlp.")


def decoder_block(x, y, xpos, ypos, sd, p, heads, base, eps=1e-6):
    """croco/models/blocks.py:186-191."""
    x = x + self_attention(layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps), xpos, sd,
                           p + "attn.", heads, base)
    yn = layer_norm(y, sd[p + "norm_y.weight"], sd[p + "norm_y.bias"], eps)
    x = x + cross_attention(layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps), yn, xpos, ypos,
                            sd, p + "cross_attn.", heads, base)
    return x + mlp(layer_norm(x, sd[p + "norm3.weight"], sd[p + "norm3.bias"], eps), sd, p + "mlp.")


def patch_embed(img, sd, p, patch):
    """dust3r/patch_embed.py:19-29: Conv2d(k=s=patch) -> [B,P,C]; plus positions."""
    x = F.conv2d(img, sd[p + "proj.weight"], sd[p + "proj.bias"], stride=patch)
    B, C, h, w = x.shape
    return x.flatten(2).transpose(1, 2), positions(B, h, w)


# ----------------------------------------------------------------------------- stages
def encode_image(img, sd, cfg):
    """dust3r/model.py:131-154."""
    x, pos = patch_embed(img, sd, "dust3r.patch_embed.", cfg.patch)
    for i in range(cfg.enc_depth):
        x = block(x, pos, sd, "dust3r.enc_blocks.%d." % i, cfg.enc_heads, cfg.rope_base)
    return layer_norm(x, sd["dust3r.enc_norm.weight"], sd["dust3r.enc_norm.bias"], 1e-6), pos


def decoder(f1, pos1, f2, pos2, sd, cfg):
    """dust3r/model.py:186-205.  Returns two lists of dec_depth+1 tensors: [enc-width input, layers...]."""
    outs1, outs2 = [f1], [f2]
    a = F.linear(f1, sd["dust3r.decoder_embed.weight"], sd["dust3r.decoder_embed.bias"])
    b = F.linear(f2, sd["dust3r.decoder_embed.weight"], sd["dust3r.decoder_embed.bias"])
    for i in range(cfg.dec_depth):
        na = decoder_block(a, b, pos1, pos2, sd, "dust3r.dec_blocks.%d." % i, cfg.dec_heads, cfg.rope_base)
        nb = decoder_block(b, a, pos2, pos1, sd, "dust3r.dec_blocks2.%d." % i, cfg.dec_heads, cfg.rope_base)
        a, b = na, nb
        outs1.append(a)
        outs2.append(b)
    outs1[-1] = layer_norm(outs1[-1], sd["dust3r.dec_norm.weight"], sd["dust3r.dec_norm.bias"], 1e-6)
    outs2[-1] = layer_norm(outs2[-1], sd["dust3r.dec_norm.weight"], sd["dust3r.dec_norm.bias"], 1e-6)
    return outs1, outs2


def encode_feat_key(feat, dec_last, sd, num):
    """spann3r/model.py:299-303: Linear(1792,1792) -> GELU -> Linear(1792,1024) on cat(feat, dec[-1])."""
    p = "attn_head_%d." % num
    x = torch.cat((feat, dec_last), dim=-1)
    return F.linear(F.gelu(F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])), sd[p + "2.weight"], sd[p + "2.bias"])


def _rcu(x, sd, p):
    """ResidualConvUnit_custom, croco/models/dpt_block.py:120-142 (bn=False, ReLU)."""
    o = F.conv2d(F.relu(x), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    o = F.conv2d(F.relu(o), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    return o + x


def _fusion(sd, p, x0, x1=None):
    """FeatureFusionBlock_custom, croco/models/dpt_block.py:190-218 (width_ratio=1, align_corners=True)."""
    out = x0
    if x1 is not None:
        out = out + _rcu(x1, sd, p + "resConfUnit1.")
    out = _rcu(out, sd, p + "resConfUnit2.")
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(out, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])


def dpt_raw(dec, hw, sd, cfg, num):
    """DPTOutputAdapter_fix.forward, dust3r/heads/dpt_head.py:34-65 -> raw [B,4,H,W]."""
    p = "dust3r.downstream_head%d.dpt." % num
    H, W = hw
    nh, nw = H // cfg.patch, W // cfg.patch
    L = [dec[h].transpose(1, 2).reshape(dec[h].shape[0], -1, nh, nw) for h in cfg.hooks]
    a = p + "act_postprocess."
    L[0] = F.conv_transpose2d(F.conv2d(L[0], sd[a + "0.0.weight"], sd[a + "0.0.bias"]),
                              sd[a + "0.1.weight"], sd[a + "0.1.bias"], stride=4)
    L[1] = F.conv_transpose2d(F.conv2d(L[1], sd[a + "1.0.weight"], sd[a + "1.0.bias"]),
                              sd[a + "1.1.weight"], sd[a + "1.1.bias"], stride=2)
    L[2] = F.conv2d(L[2], sd[a + "2.0.weight"], sd[a + "2.0.bias"])
    L[3] = F.conv2d(F.conv2d(L[3], sd[a + "3.0.weight"], sd[a + "3.0.bias"]),
                    sd[a + "3.1.weight"], sd[a + "3.1.bias"], stride=2, padding=1)
    L = [F.conv2d(l, sd[p + "scratch.layer%d_rn.weight" % (i + 1)], None, padding=1) for i, l in enumerate(L)]
    r = p + "scratch.refinenet"
    path4 = _fusion(sd, r + "4.", L[3])[:, :, :L[2].shape[2], :L[2].shape[3]]
    path3 = _fusion(sd, r + "3.", path4, L[2])
    path2 = _fusion(sd, r + "2.", path3, L[1])
    path1 = _fusion(sd, r + "1.", path2, L[0])
    h = F.conv2d(path1, sd[p + "head.0.weight"], sd[p + "head.0.bias"], padding=1)
    h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
    h = F.relu(F.conv2d(h, sd[p + "head.2.weight"], sd[p + "head.2.bias"], padding=1))
    return F.conv2d(h, sd[p + "head.4.weight"], sd[p + "head.4.bias"])


def postprocess(raw):
    """dust3r/heads/postprocess.py:10-58 with depth_mode=('exp',-inf,inf), conf_mode=('exp',1,inf)."""
    f = raw.permute(0, 2, 3, 1)
    xyz = f[..., 0:3]
    d = xyz.norm(dim=-1, keepdim=True)
    pts = xyz / d.clip(min=1e-8) * torch.expm1(d)
    conf = 1.0 + f[..., 3].exp()
    return {"pts3d": pts, "conf": conf}


def downstream_head(dec, true_shape, sd, cfg, num):
    """spann3r/model.py:327-331 -> dust3r/model.py:207-211 -> transpose_to_landscape wrapper
    (dust3r/utils/misc.py:54-96, landscape_only=True): all-landscape batches run as is,
    all-portrait batches run with (H,W) as given and the result is axis-swapped; a batch that mixes both runs the head
    once per orientation on the selected samples and scatters the results back (:80-94)."""
    hs, ws = true_shape[:, 0], true_shape[:, 1]
    Hm, Wm = int(true_shape.min()), int(true_shape.max())
    land = ws >= hs
    if bool(land.all()):
        return postprocess(dpt_raw(dec, (Hm, Wm), sd, cfg, num))
    if bool((~land).all()):
        res = postprocess(dpt_raw(dec, (Wm, Hm), sd, cfg, num))
        return {k: v.swapaxes(1, 2) for k, v in res.items()}
    sel = lambda mask: [None if d is None else d[mask] for d in dec]
    res_l = postprocess(dpt_raw(sel(land), (Hm, Wm), sd, cfg, num))
    res_p = {k: v.swapaxes(1, 2) for k, v in postprocess(dpt_raw(sel(~land), (Wm, Hm), sd, cfg, num)).items()}
    out = {}
    for k in res_l:
        x = res_l[k].new_empty((true_shape.shape[0],) + tuple(res_l[k].shape[1:]))
        x[land], x[~land] = res_l[k], res_p[k]
        out[k] = x
    return out


def encode_cur_value(pts3d, sd, cfg, dec_last=None, pos1=None):
    """spann3r/model.py:305-320: the value comes from the predicted pointmap through pos_patch_embed (use_feat=False) or from
    the last decoder output dec1[-1] with ITS positions (use_feat=True, :312-314: 768-wide blocks, 16 heads of 48); RoPE in the
    value encoder only with mem_pos_enc (:232-234)."""
    if getattr(cfg, "use_feat", False):
        x, pos = dec_last, pos1
    else:
        x, pos = patch_embed(pts3d.permute(0, 3, 1, 2), sd, "pos_patch_embed.", cfg.patch)
    for i in range(cfg.val_depth):
        x = block(x, pos, sd, "value_encoder.%d." % i, cfg.enc_heads, cfg.rope_base, use_rope=getattr(cfg, "mem_pos_enc", False))
    x = layer_norm(x, sd["value_norm.weight"], sd["value_norm.bias"], 1e-6)
    return F.linear(x, sd["value_out.weight"], sd["value_out.bias"])


# ----------------------------------------------------------------------------- spatial memory
class SpatialMemoryOracle:
    """spann3r/model.py:11-210 restated.  norm_q/k/v are LayerNorm(1024) with eps=1e-5 (:245-247)."""

    def __init__(self, sd, attn_thresh=5e-4, long_mem_size=4000, work_mem_size=5, sim_thresh=0.95):
        self.sd = sd
        self.attn_thresh = attn_thresh
        self.long_mem_size = long_mem_size
        self.work_mem_size = work_mem_size
        self.top_k = long_mem_size
        self.sim_thresh = sim_thresh
        self.num_patches = None
        self.mem_k = self.mem_v = self.mem_count = self.mem_attn = None
        self.lm = 0
        self.wm = 0
        self.log = []

    def _ln(self, x, name):
        return layer_norm(x, self.sd[name + ".weight"], self.sd[name + ".bias"], 1e-5)

    def add_mem(self, k, v):                                         # :80-95
        if self.num_patches is None:
            self.num_patches = k.shape[1]
        z = torch.zeros_like(k[:, :, :1])
        if self.mem_count is None:
            self.mem_count, self.mem_attn, self.mem_k, self.mem_v = z.clone(), z.clone(), k, v
        else:
            self.mem_count = torch.cat((self.mem_count + 1, z), 1)
            self.mem_attn = torch.cat((self.mem_attn, z), 1)
            self.mem_k = torch.cat((self.mem_k, k), 1)
            self.mem_v = torch.cat((self.mem_v, v), 1)

    def sim_scores(self, k):                                         # :97-112
        """mean-over-patches cosine similarity against each of the last `wm` stored frames -> [B, wm]."""
        n = self.wm * self.num_patches
        wmem = self.mem_k[:, -n:].reshape(self.mem_k.shape[0], -1, self.num_patches, self.mem_k.shape[-1])
        corr = torch.einsum("bpc,btpc->btp", F.normalize(k, dim=-1), F.normalize(wmem, dim=-1))
        return corr.mean(-1)

    def check_sim(self, k):                                          # :97-118
        if self.mem_k is None or self.sim_thresh == 1.0:
            return False
        return bool(self.sim_scores(k).max() > self.sim_thresh)

    def add_mem_check(self, k, v):                                   # :120-143
        if self.num_patches is None:
            self.num_patches = k.shape[1]
        if self.check_sim(k):
            self.log.append("skip")
            return
        self.add_mem(k, v)
        self.wm += 1
        if self.wm > self.work_mem_size:
            self.wm -= 1
            if self.long_mem_size == 0:
                P = self.num_patches
                self.mem_k, self.mem_v = self.mem_k[:, P:], self.mem_v[:, P:]
                self.mem_count, self.mem_attn = self.mem_count[:, P:], self.mem_attn[:, P:]
            else:
                self.lm += self.num_patches
        if self.lm > self.long_mem_size:
            self.memory_prune()
            self.lm = self.top_k - self.wm * self.num_patches

    def memory_read(self, feat):                                     # :145-183 (res=True, no dropout)
        q = self._ln(feat, "norm_q")
        aff = torch.einsum("bpc,bxc->bpx", q, self._ln(self.mem_k, "norm_k")) / math.sqrt(feat.shape[-1])
        attn = torch.softmax(aff, dim=-1)
        if self.attn_thresh > 0:
            attn = torch.where(attn < self.attn_thresh, torch.zeros_like(attn), attn)
            attn = attn / attn.sum(-1, keepdim=True)
        out = torch.einsum("bpx,bxc->bpc", attn, self._ln(self.mem_v, "norm_v")) + feat
        self.mem_attn = self.mem_attn + attn.sum(-2)[..., None]
        return out

    def prune_weights(self):                                         # :187-188
        w = self.mem_attn / self.mem_count
        return torch.where(self.mem_count < self.work_mem_size + 5, torch.full_like(w, 1e8), w)

    def memory_prune(self):                                          # :185-210
        w = self.prune_weights()
        _, idx = torch.topk(w, self.top_k, dim=1)
        self.log.append("prune %d->%d" % (self.mem_k.shape[1], self.top_k))
        g = idx.expand(-1, -1, self.mem_k.shape[-1])
        self.mem_k, self.mem_v = torch.gather(self.mem_k, 1, g), torch.gather(self.mem_v, 1, g)
        self.mem_attn, self.mem_count = torch.gather(self.mem_attn, 1, idx), torch.gather(self.mem_count, 1, idx)


# ----------------------------------------------------------------------------- forward
def default_shape(img):
    return torch.tensor(img.shape[-2:])[None].repeat(img.shape[0], 1)


@torch.no_grad()
def forward(frames, sd, cfg, training_policy=False, return_memory=False, taps=None):
    """Spann3R.forward, spann3r/model.py:473-539.  `training_policy` selects the train-mode memory
    policy (attn_thresh=0, unconditional add_mem, :474-475,518-519) with dropout disabled.
    `taps`, if a dict, receives per-step intermediate tensors for stage-level parity tests."""
    mem = SpatialMemoryOracle(sd, attn_thresh=0.0 if training_policy else 5e-4)
    feat2 = pos2 = shape2 = feat_k2 = None
    preds, preds_all = None, []
    for i in range(len(frames) - 1):
        v1, v2 = frames[i], frames[i + 1]
        if feat2 is None:                                            # :272-287, first pair in one batch
            s1 = v1.get("true_shape", default_shape(v1["img"]))
            s2 = v2.get("true_shape", default_shape(v2["img"]))
            f, p = encode_image(torch.cat((v1["img"], v2["img"]), 0), sd, cfg)
            (feat1, feat2), (pos1, pos2) = f.chunk(2, 0), p.chunk(2, 0)
            shape1, shape2 = s1, s2
        else:                                                        # :293-295
            feat1, pos1, shape1 = feat2, pos2, shape2
            feat2, pos2 = encode_image(v2["img"], sd, cfg)
            shape2 = v2.get("true_shape", default_shape(v2["img"]))
        feat_fuse = mem.memory_read(feat_k2) if feat_k2 is not None else feat1   # :496-500
        dec1, dec2 = decoder(feat_fuse, pos1, feat2, pos2, sd, cfg)              # :504
        feat_k1 = encode_feat_key(feat1, dec1[-1], sd, 1)                         # :507
        feat_k2 = encode_feat_key(feat2, dec2[-1], sd, 2)                         # :508
        res1 = downstream_head(dec1, shape1, sd, cfg, 1)                          # :512
        res2 = downstream_head(dec2, shape2, sd, cfg, 2)                          # :513
        cur_v = encode_cur_value(res1["pts3d"], sd, cfg, dec1[-1], pos1)          # :516
        if taps is not None:
            taps.setdefault("steps", []).append(dict(
                feat1=feat1, feat2=feat2, feat_fuse=feat_fuse, dec1=list(dec1), dec2=list(dec2),
                feat_k1=feat_k1, feat_k2=feat_k2, cur_v=cur_v,
                pts1=res1["pts3d"], conf1=res1["conf"], pts2=res2["pts3d"], conf2=res2["conf"]))
        if training_policy:
            mem.add_mem(feat_k1, cur_v + feat_k1)                                 # :519
        else:
            mem.add_mem_check(feat_k1, cur_v + feat_k1)                           # :521
        res2["pts3d_in_other_view"] = res2.pop("pts3d")                           # :523
        if preds is None:
            preds = [res1]
        else:
            res1["pts3d_in_other_view"] = res1.pop("pts3d")
            preds.append(res1)
        preds_all.append((res1, res2))
    preds.append(res2)
    if return_memory:
        return preds, preds_all, mem
    return preds, preds_all


# ----------------------------------------------------------------------------- offline reconstruction (spann3r/model.py:333-471)
@torch.no_grad()
def dust3r_forward(view1, view2, sd, cfg):
    """AsymmetricCroCo3DStereo.forward (dust3r/model.py:213-227).  The reference encodes only half of a symmetrized batch
    and interleaves (:176-180); encoding every image gives the same features, which is what this restatement does."""
    f1, p1 = encode_image(view1["img"], sd, cfg)
    f2, p2 = encode_image(view2["img"], sd, cfg)
    s1 = view1.get("true_shape", default_shape(view1["img"]))
    s2 = view2.get("true_shape", default_shape(view2["img"]))
    dec1, dec2 = decoder(f1, p1, f2, p2, sd, cfg)
    res1 = downstream_head(dec1, s1, sd, cfg, 1)
    res2 = downstream_head(dec2, s2, sd, cfg, 2)
    res2["pts3d_in_other_view"] = res2.pop("pts3d")
    return res1, res2


def find_initial_pair(graph, n_frames):                               # :333-358
    conf = torch.zeros(n_frames, n_frames)
    for i in range(len(graph["view1"]["idx"])):
        c1, c2 = graph["pred1"]["conf"][i], graph["pred2"]["conf"][i]
        conf[int(graph["view1"]["idx"][i]), int(graph["view2"]["idx"][i])] = ((c1 - 1) / c1).mean() + ((c2 - 1) / c2).mean()
    flat = int(conf.argmax())
    return flat // n_frames, flat % n_frames


@torch.no_grad()
def offline_reconstruction(frames, graph, sd, cfg):
    """Spann3R.offline_reconstruction (:394-471) with find_next_best_view (:360-392), one candidate at a time as the
    reference does it."""
    n = len(frames)
    todo, used = list(range(n)), []
    mem = SpatialMemoryOracle(sd)
    i0, i1 = find_initial_pair(graph, n)
    for i in (i0, i1):
        used.append(i)
        todo.remove(i)
    f, p = encode_image(torch.cat((frames[i0]["img"], frames[i1]["img"]), 0), sd, cfg)
    (feat1, feat2), (pos1, pos2) = f.chunk(2, 0), p.chunk(2, 0)
    shape1 = frames[i0].get("true_shape", default_shape(frames[i0]["img"]))
    shape2 = frames[i1].get("true_shape", default_shape(frames[i1]["img"]))
    dec1, dec2 = decoder(feat1, pos1, feat2, pos2, sd, cfg)
    res1 = downstream_head(dec1, shape1, sd, cfg, 1)
    res2 = downstream_head(dec2, shape2, sd, cfg, 2)
    feat_k2, preds, preds_all = None, None, []
    while True:
        if feat_k2 is not None:
            feat1, pos1, shape1 = feat2, pos2, shape2
            feat_fuse = mem.memory_read(feat_k2)
            best = None
            for i in todo:                                            # :360-392
                fi, pi = encode_image(frames[i]["img"], sd, cfg)
                si = frames[i].get("true_shape", default_shape(frames[i]["img"]))
                d1, d2 = decoder(feat_fuse, pos1, fi, pi, sd, cfg)
                r1 = downstream_head(d1, shape1, sd, cfg, 1)
                r2 = downstream_head(d2, si, sd, cfg, 2)
                sc = float(((r1["conf"] - 1) / r1["conf"]).mean() + ((r2["conf"] - 1) / r2["conf"]).mean())
                if sc > (0.0 if best is None else best[0]):
                    best = (sc, i, d1, d2, r1, r2, fi, pi, si)
            _, id_n, dec1, dec2, res1, res2, feat2, pos2, shape2 = best
            todo.remove(id_n)
            used.append(id_n)
        feat_k1 = encode_feat_key(feat1, dec1[-1], sd, 1)
        feat_k2 = encode_feat_key(feat2, dec2[-1], sd, 2)
        cur_v = encode_cur_value(res1["pts3d"], sd, cfg, dec1[-1], pos1)
        mem.add_mem_check(feat_k1, cur_v + feat_k1)
        res2["pts3d_in_other_view"] = res2.pop("pts3d")
        if preds is None:
            preds = [res1]
        else:
            res1["pts3d_in_other_view"] = res1.pop("pts3d")
            preds.append(res1)
        preds_all.append((res1, res2))
        if not todo:
            break
    preds.append(res2)
    return preds, preds_all, used
