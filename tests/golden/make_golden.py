#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference
(/root/reference, CPU fp32, torch-fallback RoPE) on seeded synthetic weights and frames.

Only runs in the build container (the GPU box has no /root/reference); the resulting
*.npz files are committed.  Usage:  python tests/golden/make_golden.py [tiny] [full] [memory]

What is dumped (SURVEY.md §8c "recommended dumps"):
  spann3r_tiny.npz    enc_depth=2 / dec_depth=10 model, 4 frames of 64x80, every stage tensor
  spann3r_full224.npz 24/12 model, 5 frames of 224x224 (BASELINE config 1), outputs subsampled
  memory_bank.npz     reference SpatialMemory driven stand-alone for 32 frames of P=196:
                      similarity skip, working->long-term hand-over and one prune (5096->4000)
  memory_bank_sliding.npz  the same class with long_mem_size == 0 (the sliding window of spann3r/model.py:132-137), 10 frames (`memory_sliding`)
  spann3r_offline.npz  tiny model, 5 frames of 48x64: DUSt3R pair graph (reference make_pairs + inference) and
                      Spann3R.offline_reconstruction on it (demo.py offline mode): visiting order, outputs
  spann3r_trueshape.npz  tiny model, batch 2, 4 frames of 48x80 WITH `true_shape` (landscape, and dataset-rotated portrait)
  spann3r_cfg2_224x10.npz  24/12 model, 10 frames of 224x224, eval policy (BASELINE config 2 = the bench workload):
                      subsampled outputs, per-step feat_fuse / feat_k / cur_v, final mem_attn / mem_count
  spann3r_cfg3_512x13.npz  24/12 model, 13 frames of 512x512, train memory policy with dropout off (BASELINE config 3,
                      growing bank: 11 reads, bank up to 11264 tokens), same dumps
  spann3r_cfg3_512x50.npz  the same at the benched length of config 3 (50 frames, last read over 49152 bank tokens): the frames
                      and steps around T = 24 and T = 48, the final mem_attn / mem_count (`cfg3long`)
  crop_plan.npz       the reference's OWN BaseStereoViewDataset._crop_resize_if_necessary (base_stereo_view_dataset.py:
                      140-194) + cropping.py:54-121 run on 12 input shapes (cv2 / torchvision, absent from the image and
                      unused by these functions' image path, are stubbed): every crop box, the resize target and the
                      uint8 image that comes out
"""
import argparse
import hashlib
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.serialization.add_safe_globals([argparse.Namespace])

from spann3r_amd.config import TINY, FULL  # noqa: E402
from spann3r_amd.weights import (synth_state_dict, synth_frames, state_dict_fingerprint,  # noqa: E402
                                 hash_uniform, _stream_id, stress_state_dict)


def build_reference(cfg, sd, tag, mem_pos_enc=False, use_feat=False):
    """Instantiate the reference through its real loader (dust3r/model.py:27-51)."""
    from spann3r.model import Spann3R
    path = "/tmp/golden_dust3r_%s.pth" % tag
    torch.save({"args": argparse.Namespace(model=cfg.ctor_string()),
                "model": {k[len("dust3r."):]: v for k, v in sd.items() if k.startswith("dust3r.")}}, path)
    m = Spann3R(dus3r_name=path, use_feat=use_feat, mem_pos_enc=mem_pos_enc)
    missing = m.load_state_dict(sd, strict=True)
    print(missing)
    os.remove(path)
    return m.eval()


def run_with_taps(m, frames):
    """Wrap the reference's stage methods (no edits to the reference) and record their outputs."""
    from spann3r.model import SpatialMemory
    steps = []
    cur = {}

    def wrap(name, fn, rec):
        def inner(*a, **k):
            out = fn(*a, **k)
            rec(out, a)
            return out
        return inner

    m.encode_frames = wrap("encode_frames", m.encode_frames,
                           lambda o, a: cur.update(feat1=o[0], feat2=o[1], pos1=o[2], pos2=o[3]))

    def rec_decode(o, a):
        d1, d2 = list(o[0]), list(o[1])
        cur.update(feat_fuse=a[0], dec1=d1, dec2=d2)
    orig_decode = m.decode

    def decode(*a):
        d1, d2 = orig_decode(*a)
        d1, d2 = list(d1), list(d2)
        rec_decode((d1, d2), a)
        return d1, d2
    m.decode = decode

    orig_key = m.encode_feat_key

    def key(f1, f2, num=1):
        o = orig_key(f1, f2, num)
        cur["feat_k%d" % num] = o
        return o
    m.encode_feat_key = key

    orig_head = m.downstream_head

    def head(dec, shape, num=1):
        o = orig_head(dec, shape, num)
        cur["pts%d" % num] = o["pts3d"].clone()
        cur["conf%d" % num] = o["conf"].clone()
        return o
    m.downstream_head = head

    orig_val = m.encode_cur_value

    def val(*a):
        o = orig_val(*a)
        cur["cur_v"] = o
        steps.append(dict(cur))
        cur.clear()
        return o
    m.encode_cur_value = val

    reads = []
    orig_read = SpatialMemory.memory_read

    def read(self, feat, res=True):
        o = orig_read(self, feat, res)
        reads.append(o)
        return o
    SpatialMemory.memory_read = read
    try:
        with torch.no_grad():
            preds, preds_all, sp = m(frames, return_memory=True)
    finally:
        SpatialMemory.memory_read = orig_read
    return preds, preds_all, sp, steps


def npf(t):
    return t.detach().float().cpu().numpy().astype(np.float32)


def make_tiny():
    cfg, H, W, NF = TINY, 64, 80, 4
    sd = synth_state_dict(0, cfg)
    m = build_reference(cfg, sd, "tiny")
    frames = synth_frames(NF, H, W)
    preds, preds_all, sp, steps = run_with_taps(m, frames)
    out = {"meta_hw": np.array([H, W]), "meta_frames": np.array(NF), "meta_seed": np.array(0),
           "fingerprint": np.array(state_dict_fingerprint(sd))}
    hooks = cfg.hooks
    for i, s in enumerate(steps):
        for k in ("feat1", "feat2", "feat_fuse", "feat_k1", "feat_k2", "cur_v", "pts1", "conf1", "pts2", "conf2"):
            out["s%d_%s" % (i, k)] = npf(s[k])
        for side in ("dec1", "dec2"):
            for h in hooks:
                if i == 0 or h == hooks[-1]:
                    out["s%d_%s_%d" % (i, side, h)] = npf(s[side][h])
    for j, p in enumerate(preds):
        out["pred%d_pts" % j] = npf(p["pts3d" if j == 0 else "pts3d_in_other_view"])
        out["pred%d_conf" % j] = npf(p["conf"])
    out["mem_k"], out["mem_v"] = npf(sp.mem_k), npf(sp.mem_v)
    out["mem_count"], out["mem_attn"] = npf(sp.mem_count), npf(sp.mem_attn)
    out["mem_wm_lm"] = np.array([sp.wm, sp.lm])
    # growing-bank (train-mode memory policy, dropout off) variant: final predictions only
    m2 = build_reference(cfg, sd, "tiny")
    m2.train()
    m2.mem_dropout.eval()
    with torch.no_grad():
        preds_t, _, sp_t = m2(frames, return_memory=True)
    for j, p in enumerate(preds_t):
        out["train_pred%d_pts" % j] = npf(p["pts3d" if j == 0 else "pts3d_in_other_view"])
        out["train_pred%d_conf" % j] = npf(p["conf"])
    out["train_mem_attn"] = npf(sp_t.mem_attn)
    # Spann3R(mem_pos_enc=True): RoPE inside the value encoder (spann3r/model.py:232-234): final predictions + memory values
    m3 = build_reference(cfg, sd, "tiny", mem_pos_enc=True)
    with torch.no_grad():
        preds_p, _, sp_p = m3(frames, return_memory=True)
    for j, p in enumerate(preds_p):
        out["mpe_pred%d_pts" % j] = npf(p["pts3d" if j == 0 else "pts3d_in_other_view"])
        out["mpe_pred%d_conf" % j] = npf(p["conf"])
    out["mpe_mem_v"] = npf(sp_p.mem_v)
    np.savez_compressed(os.path.join(HERE, "spann3r_tiny.npz"), **out)
    print("tiny: %d arrays" % len(out))


def make_usefeat():
    """Spann3R(use_feat=True) (spann3r/model.py:225,312-314: the value encoder works on dec1[-1]: 768-wide blocks, 16 heads of 48,
    no pos_patch_embed) on the tiny depths, 4 frames of 64x80, eval policy and the growing-bank policy: predictions, memory values"""
    import dataclasses
    cfg = dataclasses.replace(TINY, use_feat=True)
    H, W, NF = 64, 80, 4
    sd = synth_state_dict(0, cfg)
    frames = synth_frames(NF, H, W, batch=2, seed=41)
    out = {"meta_hw": np.array([H, W]), "meta_frames": np.array(NF), "meta_seed": np.array(41), "meta_batch": np.array(2),
           "fingerprint": np.array(state_dict_fingerprint(sd))}
    m = build_reference(cfg, sd, "usefeat", use_feat=True)
    for tag, train in (("eval", False), ("train", True)):
        if train:
            m.train()
            m.mem_dropout.eval()
        with torch.no_grad():
            preds, preds_all, sp = m(frames, return_memory=True)
        for j, p in enumerate(preds):
            out["%s_pred%d_pts" % (tag, j)] = npf(p["pts3d" if j == 0 else "pts3d_in_other_view"])
            out["%s_pred%d_conf" % (tag, j)] = npf(p["conf"])
        out[tag + "_mem_v"], out[tag + "_mem_k"] = npf(sp.mem_v), npf(sp.mem_k)
    np.savez_compressed(os.path.join(HERE, "spann3r_usefeat.npz"), **out)
    print("usefeat: %d arrays" % len(out))


def make_usefeat_mpe():
    """Spann3R(use_feat=True, mem_pos_enc=True): RoPE2D on the 48-wide heads of the 768-wide value encoder (spann3r/model.py:225-235,
    croco/models/pos_embed.py:112-159 with D = 24 per axis) -- the one constructor combination the MI355X build used to refuse"""
    import dataclasses
    cfg = dataclasses.replace(TINY, use_feat=True, mem_pos_enc=True)
    H, W, NF = 64, 80, 4
    sd = synth_state_dict(0, cfg)
    frames = synth_frames(NF, H, W, batch=2, seed=41)
    out = {"meta_hw": np.array([H, W]), "meta_frames": np.array(NF), "meta_seed": np.array(41), "meta_batch": np.array(2),
           "fingerprint": np.array(state_dict_fingerprint(sd))}
    m = build_reference(cfg, sd, "usefeatmpe", use_feat=True, mem_pos_enc=True)
    with torch.no_grad():
        preds, preds_all, sp = m(frames, return_memory=True)
    for j, p in enumerate(preds):
        out["eval_pred%d_pts" % j] = npf(p["pts3d" if j == 0 else "pts3d_in_other_view"])
        out["eval_pred%d_conf" % j] = npf(p["conf"])
    out["eval_mem_v"], out["eval_mem_k"] = npf(sp.mem_v), npf(sp.mem_k)
    np.savez_compressed(os.path.join(HERE, "spann3r_usefeat_mpe.npz"), **out)
    print("usefeat_mpe: %d arrays" % len(out))


def make_full():
    cfg, H, W, NF = FULL, 224, 224, 5
    sd = synth_state_dict(0, cfg)
    m = build_reference(cfg, sd, "full")
    frames = synth_frames(NF, H, W)
    t = time.time()
    preds, preds_all, sp, steps = run_with_taps(m, frames)
    dt = time.time() - t
    print("reference forward: %.2f s  (%.2f frames/s, %d threads)" % (dt, NF / dt, torch.get_num_threads()))
    out = {"meta_hw": np.array([H, W]), "meta_frames": np.array(NF), "meta_seed": np.array(0),
           "fingerprint": np.array(state_dict_fingerprint(sd)), "ref_seconds": np.array(dt),
           "ref_threads": np.array(torch.get_num_threads())}
    S = 4
    if keep is not None:                  # long sequences: only the listed frames / steps are dumped (keeps the file small)
        out["meta_keep"] = np.array(sorted(keep))
    for j, p in enumerate(preds):
        if keep is not None and j not in keep:
            continue
        pts = p["pts3d" if j == 0 else "pts3d_in_other_view"]
        out["pred%d_pts_sub" % j] = npf(pts[:, ::S, ::S])
        out["pred%d_conf_sub" % j] = npf(p["conf"][:, ::S, ::S])
        out["pred%d_stats" % j] = np.array([float(pts.double().mean()), float(pts.double().abs().mean()),
                                            float(p["conf"].double().mean())])
    for i, s in enumerate(steps):
        for k in ("feat2", "feat_fuse", "feat_k1", "feat_k2", "cur_v"):
            out["s%d_%s_sub" % (i, k)] = npf(s[k][:, ::7, ::16])
        out["s%d_dec1_last_sub" % i] = npf(s["dec1"][-1][:, ::7, ::16])
        out["s%d_dec2_last_sub" % i] = npf(s["dec2"][-1][:, ::7, ::16])
    out["mem_k_sub"], out["mem_v_sub"] = npf(sp.mem_k[:, ::7, ::16]), npf(sp.mem_v[:, ::7, ::16])
    out["mem_count"], out["mem_attn"] = npf(sp.mem_count), npf(sp.mem_attn)
    out["mem_wm_lm"] = np.array([sp.wm, sp.lm])
    np.savez_compressed(os.path.join(HERE, "spann3r_full224.npz"), **out)
    print("full224: %d arrays" % len(out))


def make_sequence_fixture(name, H, W, NF, train_policy, S, keep=None, stress=False, batch=1):
    """A whole benched configuration: outputs subsampled by S pixels, token tensors by (7, 16)."""
    cfg = FULL
    sd = synth_state_dict(0, cfg)
    if stress:                            # trained-like statistics (spann3r_amd.weights.stress_state_dict)
        sd = stress_state_dict(7, cfg)
    m = build_reference(cfg, sd, name)
    if train_policy:                      # growing bank, deterministic
        m.train()
        m.mem_dropout.eval()
    frames = synth_frames(NF, H, W, batch=batch)
    t = time.time()
    preds, preds_all, sp, steps = run_with_taps(m, frames)
    dt = time.time() - t
    print("%s: reference forward %.2f s (%.2f frames/s, %d threads)" % (name, dt, NF / dt, torch.get_num_threads()))
    out = {"meta_hw": np.array([H, W]), "meta_frames": np.array(NF), "meta_seed": np.array(0), "meta_sub": np.array(S), "meta_batch": np.array(batch),
           "meta_train_policy": np.array(int(train_policy)), "fingerprint": np.array(state_dict_fingerprint(sd)),
           "ref_seconds": np.array(dt), "ref_threads": np.array(torch.get_num_threads())}
    if keep is not None:                  # long sequences: only the listed frames / steps are dumped (keeps the file small)
        out["meta_keep"] = np.array(sorted(keep))
    for j, p in enumerate(preds):
        if keep is not None and j not in keep:
            continue
        pts = p["pts3d" if j == 0 else "pts3d_in_other_view"]
        out["pred%d_pts_sub" % j] = npf(pts[:, ::S, ::S])
        out["pred%d_conf_sub" % j] = npf(p["conf"][:, ::S, ::S])
        out["pred%d_stats" % j] = np.array([float(pts.double().mean()), float(pts.double().abs().mean()),
                                            float(p["conf"].double().mean())])
    for i, (r1, r2) in enumerate(preds_all):      # the view-2 result of every step (only the last one is in preds)
        if keep is not None and i not in keep:
            continue
        out["step%d_pts2_sub" % i] = npf(r2["pts3d_in_other_view"][:, ::S, ::S])
        out["step%d_conf2_sub" % i] = npf(r2["conf"][:, ::S, ::S])
    for i, s in enumerate(steps):
        if keep is not None and i not in keep:
            continue
        for k in ("feat_fuse", "feat_k1", "feat_k2", "cur_v"):
            out["s%d_%s_sub" % (i, k)] = npf(s[k][:, ::7, ::16])
    out["mem_k_sub"], out["mem_v_sub"] = npf(sp.mem_k[:, ::7, ::16]), npf(sp.mem_v[:, ::7, ::16])
    out["mem_count"], out["mem_attn"] = npf(sp.mem_count), npf(sp.mem_attn)
    out["mem_wm_lm"] = np.array([sp.wm, sp.lm])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%s: %d arrays" % (name, len(out)))


def make_autocast():
    """The reference ITSELF under torch's bf16 autocast (CPU), on the config-2 and the stress workloads: its max-norm errors against
    its own fp32 run (the committed fixtures) anchor the bf16 tolerance of tests/test_model_gpu.py -- the bf16 mode of the MI355X
    build must not be worse than 1.5x what bf16 autocast does to the reference (the DPT heads stay fp32 there too: model.py:327-329)."""
    out = {}
    for tag, name, H, W, NF, S, stress in (("cfg2", "spann3r_cfg2_224x10", 224, 224, 10, 4, False),
                                           ("stress", "spann3r_stress_224x6", 224, 224, 6, 4, True)):
        g = np.load(os.path.join(HERE, name + ".npz"))
        sd = stress_state_dict(7, FULL) if stress else synth_state_dict(0, FULL)
        m = build_reference(FULL, sd, name)
        frames = synth_frames(NF, H, W)
        t = time.time()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            preds, preds_all, sp, steps = run_with_taps(m, frames)
        print("%s: reference forward under bf16 autocast %.1f s" % (name, time.time() - t))

        def rel(a, b):
            a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
            return float((a - b).abs().max() / b.abs().max())
        err = {"pts": 0.0, "conf": 0.0, "pts2": 0.0, "fuse": 0.0, "k": 0.0}
        for j, p in enumerate(preds):
            pts = p["pts3d" if j == 0 else "pts3d_in_other_view"]
            err["pts"] = max(err["pts"], rel(npf(pts[:, ::S, ::S]), g["pred%d_pts_sub" % j]))
            err["conf"] = max(err["conf"], rel(npf(p["conf"][:, ::S, ::S]), g["pred%d_conf_sub" % j]))
        for i, (_, r2) in enumerate(preds_all):
            err["pts2"] = max(err["pts2"], rel(npf(r2["pts3d_in_other_view"][:, ::S, ::S]), g["step%d_pts2_sub" % i]),
                              rel(npf(r2["conf"][:, ::S, ::S]), g["step%d_conf2_sub" % i]))
        for i, st in enumerate(steps):
            if i > 0:
                err["fuse"] = max(err["fuse"], rel(npf(st["feat_fuse"].float()[:, ::7, ::16]), g["s%d_feat_fuse_sub" % i]))
            err["k"] = max(err["k"], rel(npf(st["feat_k1"].float()[:, ::7, ::16]), g["s%d_feat_k1_sub" % i]),
                           rel(npf(st["feat_k2"].float()[:, ::7, ::16]), g["s%d_feat_k2_sub" % i]))
        err["mem_attn"] = rel(npf(sp.mem_attn.float()), g["mem_attn"])
        print(tag, {k: "%.2e" % v for k, v in err.items()})
        for k, v in err.items():
            out["%s_%s" % (tag, k)] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "reference_bf16_autocast.npz"), **out)
    print("reference_bf16_autocast: %d arrays" % len(out))


def make_trueshape():
    """What real callers send (dust3r/datasets/base/base_stereo_view_dataset.py:89,215-220; demo.py:109): every view carries
    `true_shape` as a CPU int32 tensor.  Case L: landscape image, true_shape = image shape.  Case P: a portrait image the
    dataset rotated to landscape, true_shape = (W_img, H_img): the patch embed ignores it, the heads regroup the tokens."""
    cfg, H, W, NF = TINY, 48, 80, 4
    sd = synth_state_dict(0, cfg)
    frames = synth_frames(NF, H, W, batch=2, seed=31)
    out = {"meta_hw": np.array([H, W]), "meta_frames": np.array(NF), "meta_seed": np.array(31), "meta_batch": np.array(2),
           "fingerprint": np.array(state_dict_fingerprint(sd))}
    for tag, ts in (("L", (H, W)), ("P", (W, H))):
        m = build_reference(cfg, sd, "ts" + tag)
        fr = [dict(f, true_shape=torch.tensor([ts, ts], dtype=torch.int32)) for f in frames]
        with torch.no_grad():
            preds, preds_all, sp = m(fr, return_memory=True)
        for j, p in enumerate(preds):
            out["%s_pred%d_pts" % (tag, j)] = npf(p["pts3d" if j == 0 else "pts3d_in_other_view"])
            out["%s_pred%d_conf" % (tag, j)] = npf(p["conf"])
        for i, (r1, r2) in enumerate(preds_all):
            out["%s_step%d_conf2" % (tag, i)] = npf(r2["conf"])
        out["%s_mem_attn" % tag] = npf(sp.mem_attn)
    np.savez_compressed(os.path.join(HERE, "spann3r_trueshape.npz"), **out)
    print("trueshape: %d arrays" % len(out))


def make_mixedshape():
    """A training-style batch that MIXES orientations (spann3r/training.py:216 batches do): sample 0 landscape, sample 1 a portrait
    the dataset rotated to landscape -- dust3r/utils/misc.py:80-94 runs the head once per orientation and scatters the results."""
    cfg, H, W, NF = TINY, 48, 80, 4
    sd = synth_state_dict(0, cfg)
    frames = synth_frames(NF, H, W, batch=2, seed=31)
    out = {"meta_hw": np.array([H, W]), "meta_frames": np.array(NF), "meta_seed": np.array(31), "meta_batch": np.array(2),
           "fingerprint": np.array(state_dict_fingerprint(sd))}
    m = build_reference(cfg, sd, "tsM")
    fr = [dict(f, true_shape=torch.tensor([(H, W), (W, H)], dtype=torch.int32)) for f in frames]
    with torch.no_grad():
        preds, preds_all, sp = m(fr, return_memory=True)
    for j, p in enumerate(preds):
        out["M_pred%d_pts" % j] = npf(p["pts3d" if j == 0 else "pts3d_in_other_view"])
        out["M_pred%d_conf" % j] = npf(p["conf"])
    for i, (r1, r2) in enumerate(preds_all):
        out["M_step%d_conf2" % i] = npf(r2["conf"])
    out["M_mem_attn"] = npf(sp.mem_attn)
    np.savez_compressed(os.path.join(HERE, "spann3r_mixedshape.npz"), **out)
    print("mixedshape: %d arrays" % len(out))


def make_offline():
    """demo.py's offline mode on the tiny model: the DUSt3R pair graph through the reference's own make_pairs / inference
    (complete graph, symmetrised, batch 2) and Spann3R.offline_reconstruction on it."""
    from dust3r.inference import inference
    from dust3r.image_pairs import make_pairs
    cfg, H, W, NF = TINY, 48, 64, 5
    sd = synth_state_dict(0, cfg)
    m = build_reference(cfg, sd, "offline")
    frames = synth_frames(NF, H, W, seed=41)
    imgs_all = [dict(img=f["img"], true_shape=torch.tensor(f["img"].shape[2:]).unsqueeze(0), idx=j, instance=str(j))
                for j, f in enumerate(frames)]                                   # demo.py:100-112
    pairs = make_pairs(imgs_all, scene_graph="complete", prefilter=None, symmetrize=True)
    graph = inference(pairs, m.dust3r, "cpu", batch_size=2, verbose=False)
    with torch.no_grad():
        preds, preds_all, idx_used = m.offline_reconstruction(frames, graph)
    out = {"meta_hw": np.array([H, W]), "meta_frames": np.array(NF), "meta_seed": np.array(41),
           "fingerprint": np.array(state_dict_fingerprint(sd)), "idx_used": np.array(idx_used),
           "graph_idx1": np.array(graph["view1"]["idx"]), "graph_idx2": np.array(graph["view2"]["idx"]),
           "graph_conf1": npf(graph["pred1"]["conf"]), "graph_conf2": npf(graph["pred2"]["conf"]),
           "graph_pts1_sub": npf(graph["pred1"]["pts3d"][:, ::4, ::4]), "graph_pts2_sub": npf(graph["pred2"]["pts3d_in_other_view"][:, ::4, ::4])}
    for j, p in enumerate(preds):
        out["pred%d_pts" % j] = npf(p["pts3d" if j == 0 else "pts3d_in_other_view"])
        out["pred%d_conf" % j] = npf(p["conf"])
    for i, (r1, r2) in enumerate(preds_all):
        out["step%d_conf2" % i] = npf(r2["conf"])
    np.savez_compressed(os.path.join(HERE, "spann3r_offline.npz"), **out)
    print("offline: idx_used", idx_used, "-", len(out), "arrays")


from memory_inputs import memory_inputs  # noqa: E402  (shared with the tests)


def make_memory():
    from spann3r.model import SpatialMemory
    sd = synth_state_dict(0, TINY)
    norms = {}
    for n in ("norm_q", "norm_k", "norm_v"):
        ln = torch.nn.LayerNorm(1024)
        ln.weight.data.copy_(sd[n + ".weight"])
        ln.bias.data.copy_(sd[n + ".bias"])
        norms[n] = ln
    sp = SpatialMemory(norms["norm_q"], norms["norm_k"], norms["norm_v"])
    out = {"n_steps": np.array(32)}
    events = []
    with torch.no_grad():
        for step in range(32):
            k, v, q = memory_inputs(step)
            if sp.mem_k is not None:
                o = sp.memory_read(q, res=True)
                out["read%d_sub" % step] = npf(o[:, ::7, ::16])
                out["read%d_mean" % step] = np.array(float(o.double().mean()))
            before = None if sp.mem_k is None else sp.mem_k.shape[1]
            sp.add_mem_check(k, v)
            after = sp.mem_k.shape[1]
            events.append([step, -1 if before is None else before, after, sp.wm, sp.lm])
    out["events"] = np.array(events)
    out["mem_count"], out["mem_attn"] = npf(sp.mem_count), npf(sp.mem_attn)
    out["mem_k_sub"], out["mem_v_sub"] = npf(sp.mem_k[:, :, ::64]), npf(sp.mem_v[:, :, ::64])
    np.savez_compressed(os.path.join(HERE, "memory_bank.npz"), **out)
    print("memory: events\n", np.array(events))


def make_memory_sliding():
    """memory_bank_sliding.npz: the reference SpatialMemory with long_mem_size == 0 (spann3r/model.py:132-137: the bank keeps the last
    work_mem_size frames), 10 frames of P = 196, every read and the final bank"""
    from spann3r.model import SpatialMemory
    sd = synth_state_dict(0, TINY)
    norms = {}
    for n in ("norm_q", "norm_k", "norm_v"):
        ln = torch.nn.LayerNorm(1024)
        ln.weight.data.copy_(sd[n + ".weight"])
        ln.bias.data.copy_(sd[n + ".bias"])
        norms[n] = ln
    sp = SpatialMemory(norms["norm_q"], norms["norm_k"], norms["norm_v"], long_mem_size=0, work_mem_size=3, sim_thresh=1.0)
    out = {"n_steps": np.array(10)}
    events = []
    with torch.no_grad():
        for step in range(10):
            k, v, q = memory_inputs(step)
            if sp.mem_k is not None:
                o = sp.memory_read(q, res=True)
                out["read%d_sub" % step] = npf(o[:, ::7, ::16])
            sp.add_mem_check(k, v)
            events.append([step, sp.mem_k.shape[1], sp.wm, sp.lm])
    out["events"] = np.array(events)
    out["mem_count"], out["mem_attn"] = npf(sp.mem_count), npf(sp.mem_attn)
    out["mem_k_sub"], out["mem_v_sub"] = npf(sp.mem_k[:, :, ::64]), npf(sp.mem_v[:, :, ::64])
    np.savez_compressed(os.path.join(HERE, "memory_bank_sliding.npz"), **out)
    print("memory (sliding window): events\n", np.array(events))


def make_loss():
    """spann3r/loss.py ConfLoss_t(Regr3D_t(L21, norm_mode='avg_dis', fix_first=...), alpha) on seeded inputs: loss, details,
    factor loss and the autograd gradients of (loss + factor loss) w.r.t. every predicted pointmap / confidence."""
    from spann3r.loss import Regr3D_t, ConfLoss_t
    from dust3r.losses import L21
    from spann3r_amd.weights import synth_loss_case
    out = {}
    for tag, seed, fix_first, alpha, scale in (("a", 11, False, 0.4, 1.0), ("b", 12, True, 1.0, 1.0), ("c", 13, False, 0.4, 3.0)):
        gts, preds_all = synth_loss_case(seed)
        leaves = []
        for r1, r2 in preds_all:
            for r in (r1, r2):
                for k in r:
                    if scale != 1.0 and k != "conf":
                        r[k] = r[k] * scale                   # predictions larger than the ground truth: factor loss active
                    r[k].requires_grad_(True)
                    leaves.append(r[k])
        crit = ConfLoss_t(Regr3D_t(L21, norm_mode="avg_dis", fix_first=fix_first), alpha=alpha)
        loss, details, factor = crit.compute_frame_loss(gts, preds_all)
        total = loss + factor
        total.backward()
        out[tag + "_meta"] = np.array([seed, int(fix_first)], np.int64)
        out[tag + "_alpha"] = np.float32(alpha)
        out[tag + "_scale"] = np.float32(scale)
        out[tag + "_loss"] = npf(loss)
        out[tag + "_factor"] = np.float32(float(factor))
        for k, v in details.items():
            out[tag + "_detail_" + k] = np.float32(float(v))
        for j, t in enumerate(leaves):
            out[tag + "_grad%d" % j] = npf(t.grad)
        print("loss", tag, float(loss), float(factor), {k: round(float(v), 5) for k, v in details.items()})
    # the test criterion of spann3r/training.py:39
    from spann3r.loss import Regr3D_t_ScaleShiftInv
    for tag, seed, gsc in (("t1", 21, True), ("t2", 22, False)):
        gts, preds_all = synth_loss_case(seed)
        with torch.no_grad():
            loss, details, factor = Regr3D_t_ScaleShiftInv(L21, gt_scale=gsc).compute_frame_loss(gts, preds_all)
        out[tag + "_meta"] = np.array([seed, int(gsc)], np.int64)
        out[tag + "_loss"] = np.float32(float(loss))
        out[tag + "_factor"] = np.float32(float(factor))
        for k, v in details.items():
            out[tag + "_detail_" + k] = np.float32(float(v))
        print("ssi", tag, float(loss), float(factor), {k: round(float(v), 5) for k, v in details.items()})
    np.savez_compressed(os.path.join(HERE, "loss_conf.npz"), **out)


def make_postprocess():
    from dust3r.post_process import estimate_focal_knowing_depth
    from spann3r_amd.weights import synth_pointmaps
    out = {}
    for tag, seed in (("a", 21), ("b", 22)):
        pts = synth_pointmaps(seed)
        _, H, W, _ = pts.shape
        pp = torch.tensor((W / 2, H / 2))
        out[tag + "_seed"] = np.int64(seed)
        out[tag + "_focal"] = npf(estimate_focal_knowing_depth(pts, pp, focal_mode="weiszfeld"))
        out[tag + "_focal_clip"] = npf(estimate_focal_knowing_depth(pts, pp, focal_mode="weiszfeld", min_focal=1.2, max_focal=1.3))
        print("focal", tag, out[tag + "_focal"], out[tag + "_focal_clip"])
    np.savez_compressed(os.path.join(HERE, "postprocess.npz"), **out)


def make_traingrad():
    """f1 at FULL geometry: one training step of the UNMODIFIED reference (float32: its downstream_head casts the tokens with
    .float(), spann3r/model.py:329, so a float64 copy of the module does not run) -- Spann3R.forward in train mode (memory
    dropout 0 so that the step is deterministic), spann3r/loss.py ConfLoss_t(Regr3D_t(L21, avg_dis), 0.4).compute_frame_loss,
    (loss + factor).backward() -- on the 24/12-layer model and 3 frames of 64x80, batch 2.  Dumped: loss, factor and, for every
    parameter tensor, max |grad| plus a strided sample of its gradient (<= 256 elements): the device step is compared against
    the reference itself at full depth and width (the tiny-geometry test compares all 724 tensors against the oracle)."""
    from spann3r.loss import Regr3D_t, ConfLoss_t
    from dust3r.losses import L21
    cfg, H, W, NF, B = FULL, 64, 80, 3, 2
    sd = synth_state_dict(0, cfg)
    m = build_reference(cfg, sd, "traingrad")
    m.train()
    m.mem_dropout.p = 0.0
    frames = synth_frames(NF, H, W, batch=B, seed=77)
    g = torch.Generator().manual_seed(78)
    views = []
    for i, f in enumerate(frames):
        Q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
        pose = torch.eye(4).repeat(B, 1, 1)
        pose[:, :3, :3] = Q
        pose[:, :3, 3] = torch.randn(B, 3, generator=g) * 0.3
        views.append(dict(img=f["img"], true_shape=torch.tensor([[H, W]] * B, dtype=torch.int32),
                          pts3d=torch.randn(B, H, W, 3, generator=g) + torch.tensor([0.0, 0.0, 3.0]),
                          valid_mask=torch.rand(B, H, W, generator=g) < 0.85, camera_pose=pose))
    t = time.time()
    preds, preds_all = m(views)
    crit = ConfLoss_t(Regr3D_t(L21, norm_mode="avg_dis", fix_first=False), alpha=0.4)
    loss, details, factor = crit.compute_frame_loss(views, preds_all)
    (loss + factor).backward()
    print("traingrad: reference float32 step %.1f s, loss %.6f factor %.6f" % (time.time() - t, float(loss), float(factor)))
    out = {"meta": np.array([H, W, NF, B, 77, 78]), "loss": np.float64(float(loss)), "factor": np.float64(float(factor)),
           "fingerprint": np.array(state_dict_fingerprint(sd))}
    for k in ("pts3d", "valid_mask", "camera_pose"):
        out["gt_" + k] = np.stack([v[k].numpy() for v in views])
    names = []
    for name, p in m.named_parameters():
        if p.grad is None:
            continue
        gflat = p.grad.reshape(-1)
        step = max(1, gflat.numel() // 256)
        names.append(name)
        out["g_" + name + "_max"] = np.float64(float(gflat.abs().max()))
        out["g_" + name + "_sample"] = gflat[::step][:256].numpy().astype(np.float64)
        out["g_" + name + "_step"] = np.int64(step)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "train_grad_full.npz"), **out)
    print("train_grad_full.npz: %d parameter gradients" % len(names))


def synth_gt_views(frames, H, W, B, g):
    """ground-truth side of a training batch (pointmaps in front of the camera, 85 % valid, random rigid poses)"""
    views = []
    for f in frames:
        Q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
        pose = torch.eye(4).repeat(B, 1, 1)
        pose[:, :3, :3] = Q
        pose[:, :3, 3] = torch.randn(B, 3, generator=g) * 0.3
        views.append(dict(img=f["img"], true_shape=torch.tensor([[H, W]] * B, dtype=torch.int32),
                          pts3d=torch.randn(B, H, W, 3, generator=g) + torch.tensor([0.0, 0.0, 3.0]),
                          valid_mask=torch.rand(B, H, W, generator=g) < 0.85, camera_pose=pose))
    return views


def make_trainloop():
    """f1, the LOOP around the step: the reference's UNMODIFIED `train_one_epoch` (spann3r/training.py:168-262) run here on the CPU
    for two epochs of 4 iterations with accum_iter = 2 -- per-iteration learning rate (croco/utils/misc.py:464-479: epoch 0 is the
    linear warm-up starting at lr = 0, epoch 1 the cosine branch), losses divided by accum_iter and accumulated, clip at 1.0 and
    AdamW(get_parameter_groups(model, 0.05), betas=(0.9, 0.95)) every second iteration through NativeScalerWithGradNormCount (its
    GradScaler disables itself without CUDA).  `spann3r.training` imports tensorboard and the data sets (cv2, ...), neither needed
    by the function: both are stubbed in sys.modules before the module is imported.  Tiny model, memory dropout 0 (deterministic),
    3 frames of 64x80, batch 2.  Dumped: the ground truth of the 8 batches, per-iteration loss / lr / gradient norm (through a
    recording wrapper around the loss scaler) and, for every parameter, a strided sample of its value after each epoch."""
    import types
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    ds = types.ModuleType("spann3r.datasets")
    ds.__all__ = []
    sys.modules.setdefault("spann3r.datasets", ds)
    import croco.utils.misc as misc
    import spann3r.training as T
    from spann3r.loss import Regr3D_t, ConfLoss_t
    from dust3r.losses import L21
    cfg, H, W, NF, B, NIT, NEP = TINY, 64, 80, 3, 2, 4, 2
    sd = synth_state_dict(0, cfg)
    m = build_reference(cfg, sd, "trainloop")
    m.train()
    m.mem_dropout.p = 0.0
    args = argparse.Namespace(accum_iter=2, epochs=10, warmup_epochs=1, lr=2e-5, min_lr=1e-6, print_freq=1, weight_decay=0.05)
    g = torch.Generator().manual_seed(91)
    batches = [synth_gt_views(synth_frames(NF, H, W, batch=B, seed=200 + i), H, W, B, g) for i in range(NIT * NEP)]

    class Loader:
        class dataset:
            set_epoch = staticmethod(lambda e: None)
            set_ratio = staticmethod(lambda r: None)

        def __init__(self, items):
            self.items = items

        def __len__(self):
            return len(self.items)

        def __iter__(self):
            return iter([[dict(v) for v in b] for b in self.items])

    rec = {"loss": [], "norm": [], "lr": [], "untouched": []}
    inner = misc.NativeScalerWithGradNormCount()

    def scaler(loss, optimizer, **kw):
        n = inner(loss, optimizer, **kw)
        rec["loss"].append(float(loss) * args.accum_iter)          # the loop divided it by accum_iter just before (:228)
        rec["norm"].append(float("nan") if n is None else float(n))
        rec["lr"].append(optimizer.param_groups[0]["lr"])
        if n is not None:                                          # parameters AdamW passed over in this update (grad None)
            rec["untouched"] = [name for name, p in m.named_parameters() if p.requires_grad and p.grad is None]
        return n

    crit = ConfLoss_t(Regr3D_t(L21, norm_mode="avg_dis", fix_first=False), alpha=0.4)
    opt = torch.optim.AdamW(misc.get_parameter_groups(m, args.weight_decay), lr=args.lr, betas=(0.9, 0.95))      # training.py:326-327
    torch.backends.cuda.matmul.allow_tf32 = True                   # asserted at the top of train_one_epoch
    out = {"meta": np.array([H, W, NF, B, NIT, NEP, 91, 200]), "fingerprint": np.array(state_dict_fingerprint(sd)),
           "args": np.array([args.accum_iter, args.epochs, args.warmup_epochs, args.lr, args.min_lr, args.weight_decay])}
    for i, b in enumerate(batches):
        out["gt%d_pts3d" % i] = np.stack([v["pts3d"].numpy() for v in b]).astype(np.float32)
        out["gt%d_valid" % i] = np.packbits(np.stack([v["valid_mask"].numpy() for v in b]))
        out["gt%d_pose" % i] = np.stack([v["camera_pose"].numpy() for v in b]).astype(np.float32)
    t = time.time()
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    for ep in range(NEP):
        stats = T.train_one_epoch(m, crit, Loader(batches[ep * NIT:(ep + 1) * NIT]), opt, torch.device("cpu"), ep, scaler, args)
        print("trainloop: epoch %d  %s" % (ep, {k: round(v, 6) for k, v in stats.items()}))
        for name, p in m.named_parameters():
            if p.requires_grad:
                flat = p.detach().reshape(-1)
                step = max(1, flat.numel() // 64)
                out["p%d_%s" % (ep, name)] = flat[::step][:64].numpy().astype(np.float64)
    for k in ("loss", "norm", "lr"):
        out["it_" + k] = np.array(rec[k], dtype=np.float64)
    out["names"] = np.array(names)
    out["untouched"] = np.array(rec["untouched"])
    print("trainloop: %.1f s; lr %s; loss %s; norms %s; %d of %d parameters without a gradient: %s"
          % (time.time() - t, rec["lr"], rec["loss"], rec["norm"], len(rec["untouched"]), len(names), rec["untouched"][:8]))
    np.savez_compressed(os.path.join(HERE, "train_loop_tiny.npz"), **out)
    print("train_loop_tiny.npz: %d arrays" % len(out))


def make_lrsched():
    """the reference's per-iteration schedule (croco/utils/misc.py:464-479) sampled through the UNMODIFIED function on a stand-in
    optimizer with and without per-group lr_scale: 3 argument sets x 64 fractional epochs"""
    import croco.utils.misc as misc

    class Opt:
        def __init__(self):
            self.param_groups = [{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.65}]

    sets = [(5e-5, 1e-6, 1, 120), (1e-4, 0.0, 5, 40), (2e-5, 1e-6, 0, 10)]
    out = {"sets": np.array(sets, dtype=np.float64)}
    for k, (lr, min_lr, warm, epochs) in enumerate(sets):
        args = argparse.Namespace(lr=lr, min_lr=min_lr, warmup_epochs=warm, epochs=epochs)
        eps = np.linspace(0.0, epochs, 64)
        opt, rows = Opt(), []
        for e in eps:
            ret = misc.adjust_learning_rate(opt, float(e), args)
            rows.append((ret, opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]))
        out["epochs%d" % k], out["lr%d" % k] = eps, np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "lr_schedule.npz"), **out)
    print("lr_schedule.npz: %d arrays" % len(out))


def make_crop():
    """f3 pin: the crop / resize plan of the reference's own functions.  `cropping.py` imports cv2 (used only for the depth map,
    cropping.py:73-75) and `dust3r.utils.image` imports torchvision (ImgNorm); neither is installed here and neither touches the
    image path of `_crop_resize_if_necessary`, so both are stubbed in sys.modules before the UNMODIFIED modules are imported.
    The crop boxes are recorded by wrapping `cropping.crop_image_depthmap` / `ImageList.resize`; the function's uint8 output
    image is dumped too (Pillow does the pixel work: the installed Pillow is the third-party dependency)."""
    import types
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.IMREAD_UNCHANGED, cv2.IMREAD_COLOR = 0, -1, 1

    def _resize(a, size, fx=None, fy=None, interpolation=None):      # nearest, depth map only (not part of the fixture)
        w, h = int(size[0]), int(size[1])
        yy = (np.arange(h) * a.shape[0] / h).astype(int).clip(0, a.shape[0] - 1)
        xx = (np.arange(w) * a.shape[1] / w).astype(int).clip(0, a.shape[1] - 1)
        return a[yy][:, xx]
    cv2.resize = _resize
    tv, tvf = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    for n in ("Compose", "ToTensor", "Normalize", "ColorJitter"):
        setattr(tvf, n, lambda *a, **k: None)
    tv.transforms = tvf
    sys.modules.setdefault("cv2", cv2)
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvf)
    import dust3r.datasets.utils.cropping as cropping
    from dust3r.datasets.base.base_stereo_view_dataset import BaseStereoViewDataset
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_preprocess import _image

    rec = {}
    real_crop, real_resize = cropping.crop_image_depthmap, cropping.ImageList.resize

    def crop(image, depthmap, K, bbox):
        rec.setdefault("crops", []).append(tuple(int(v) for v in bbox))
        return real_crop(image, depthmap, K, bbox)

    def resize(self, size, **kw):
        rec["resize"] = tuple(int(v) for v in size)
        return real_resize(self, size, **kw)
    cropping.crop_image_depthmap, cropping.ImageList.resize = crop, resize
    cases = [((480, 640), (224, 224)), ((640, 480), (224, 224)), ((375, 500), (512, 384)), ((700, 500), (512, 384)),
             ((1080, 1920), (512, 384)), ((481, 641), (224, 224)), ((1000, 751), (512, 384)), ((97, 131), (224, 224)),
             ((720, 1280), (512, 288)), ((333, 517), (512, 336)),
             # near-square input, non-square resolution: the reference draws rng.integers(2) (:174-177); both outcomes
             ((1000, 1000), (512, 384)), ((528, 500), (512, 384))]
    out = {"n": np.int32(len(cases))}
    dummy = types.SimpleNamespace(aug_crop=0)
    for i, (hw, res) in enumerate(cases):
        H, W = hw
        rgb = _image(H, W, seed=H + i)
        K = np.array([[1.0, 0, W // 2], [0, 1.0, H // 2], [0, 0, 1]], dtype=np.float32)      # spann3r/datasets/demo.py:77-78
        for seed in ((0, 1, 2, 3) if i >= 10 else (0,)):
            rec.clear()
            rng = np.random.default_rng(seed)
            img, _, K2 = BaseStereoViewDataset._crop_resize_if_necessary(dummy, rgb, np.ones((H, W), np.float32), K.copy(), res,
                                                                       rng=rng, info="case%d" % i)
            tag = "c%d_s%d_" % (i, seed)
            out[tag + "hw"], out[tag + "res"] = np.int32(hw), np.int32(res)
            out[tag + "crop0"], out[tag + "crop1"] = np.int32(rec["crops"][0]), np.int32(rec["crops"][1])
            out[tag + "resize"] = np.int32(rec["resize"])
            a = np.ascontiguousarray(np.asarray(img))
            # the whole image as a digest (bit-exact comparison without 8 MB of noise in the repository) + its centre patch
            out[tag + "img_shape"] = np.int32(a.shape)
            out[tag + "img_sha256"] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), np.uint8)
            out[tag + "img_centre"] = a[a.shape[0] // 2 - 8: a.shape[0] // 2 + 8, a.shape[1] // 2 - 8: a.shape[1] // 2 + 8].copy()
            out[tag + "K"] = np.float32(K2)
            out[tag + "coin"] = np.int32(np.random.default_rng(seed).integers(2))
    np.savez_compressed(os.path.join(HERE, "crop_plan.npz"), **out)
    print("crop_plan.npz:", len(cases), "shapes")


def make_pairs_fixture():
    """missing #6 of VERDICT r4: dust3r/image_pairs.py:11-46 make_pairs for every scene_graph / prefilter it knows, on index-only
    "images" (the function only reads img['idx']): the pair lists spann3r_amd.runner.pair_indices must reproduce, order included."""
    import json
    from dust3r.image_pairs import make_pairs
    out = {}
    for n in (2, 3, 5, 8):
        imgs = [dict(idx=i) for i in range(n)]
        for sg in ("complete", "swin", "swin-2", "swin-5", "oneref", "oneref-2", "prev"):
            if sg == "oneref-2" and n < 3:
                continue
            for pf in (None, "seq1", "seq3", "cyc1", "cyc2"):
                for sym in (True, False):
                    pr = make_pairs(imgs, scene_graph=sg, prefilter=pf, symmetrize=sym)
                    out["%d|%s|%s|%d" % (n, sg, pf, sym)] = [[a["idx"], b["idx"]] for a, b in pr]
    with open(os.path.join(HERE, "pairs.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("pairs.json:", len(out), "graphs")


def make_pnp():
    """f4: the poses demo.py:170-186 computes -- cv2.solvePnPRansac(points, pixel grid, K, zeros(4)) with OpenCV's defaults,
    Rodrigues, inverse -- on the synthetic scenes of tests/test_postprocess.py.  Needs OpenCV,
    which the build image does not have: run where it is installed (`python tests/golden/make_golden.py pnp`) and commit
    tests/golden/pnp_cv2.npz; until then test_estimate_poses_vs_opencv skips and f4 stays 'parity unpinned'."""
    import cv2
    sys.path.insert(0, os.path.dirname(HERE))
    from test_postprocess import _scene
    out = {}
    for tag, (seed, kw) in {"clean": (5, {}), "outliers": (6, {"outliers": 0.3}), "four": (7, {"F": 4}), "clear": (8, {"F": 4, "clear_px": 40.0})}.items():
        pts, poses, f, (cx, cy) = _scene(seed, **kw)
        F, H, W, _ = pts.shape
        K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]], np.float64)
        uu, vv = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
        pix = np.stack((uu, vv), -1).reshape(-1, 2)
        got = []
        for j in range(F):
            X = pts[j].reshape(-1, 3)
            ok = np.isfinite(X).all(-1)                 # (the demo's pointmaps are finite; the scenes carry two non-finite probes)
            _, rvec, tvec, inl = cv2.solvePnPRansac(X[ok].astype(np.float32), pix[ok], K.astype(np.float32), np.zeros(4, np.float32))
            R, _ = cv2.Rodrigues(rvec)
            P = np.eye(4)
            P[:3, :3], P[:3, 3] = R, tvec[:, 0]
            got.append(np.linalg.inv(P))                # camera-to-world, as demo.py stores it
        out[tag + "_poses"] = np.stack(got)
        out[tag + "_seed"] = np.array(seed)
    np.savez_compressed(os.path.join(HERE, "pnp_cv2.npz"), **out)
    print("pnp_cv2:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    what = sys.argv[1:] or ["tiny", "full", "memory"]
    if "pairs" in what:
        make_pairs_fixture()
    if "pnp" in what:
        make_pnp()
    if "autocast" in what:
        make_autocast()
    if "crop" in what:
        make_crop()
    if "usefeat" in what:
        make_usefeat()
    if "usefeatmpe" in what:
        make_usefeat_mpe()
    if "traingrad" in what:
        make_traingrad()
    if "trainloop" in what:
        make_trainloop()
    if "lrsched" in what:
        make_lrsched()
    if "postprocess" in what:
        make_postprocess()
    if "loss" in what:
        make_loss()
    if "tiny" in what:
        make_tiny()
    if "memory" in what:
        make_memory()
    if "memory_sliding" in what:
        make_memory_sliding()
    if "full" in what:
        make_full()
    if "offline" in what:
        make_offline()
    if "trueshape" in what:
        make_trueshape()
    if "mixedshape" in what:
        make_mixedshape()
    if "cfg2" in what:
        make_sequence_fixture("spann3r_cfg2_224x10", 224, 224, 10, False, 4)
    if "cfg3" in what:
        make_sequence_fixture("spann3r_cfg3_512x13", 512, 512, 13, True, 8)
    if "stress" in what:
        # trained-like weight statistics through the reference: the parity claim of the fast fp32 mode (f32x3) is tested here
        make_sequence_fixture("spann3r_stress_224x6", 224, 224, 6, False, 4, stress=True)
    if "cfg3long" in what:
        # BASELINE config 3 at its benched length: 50 frames, bank up to 49152 tokens at the last read (split-K 16 plan);
        # frames / steps around T = 24 and T = 48 and the final state are kept
        make_sequence_fixture("spann3r_cfg3_512x50", 512, 512, 50, True, 8, keep={0, 23, 24, 25, 46, 47, 48, 49})
    if "demo160" in what:
        # what demo.py feeds for 4:3 photos (load_images(size=224): long side 224, short side a multiple of 16): 160 x 224 = 10 x 14 = 140
        # tokens per frame -- fewer rows than the square benchmark geometry, a token count that is no multiple of 16; eval policy
        make_sequence_fixture("spann3r_demo_160x224x6", 160, 224, 6, False, 4)
    if "mid288" in what:
        # a 16:9 geometry between the two benched ones: 288 x 512 = 18 x 32 = 576 tokens per frame (the 257..1535-row instances of the
        # many-row families and the > 256-row memory read in the FULL model, which neither 196 nor 1024 tokens reach); growing bank
        make_sequence_fixture("spann3r_mid_288x512x5", 288, 512, 5, True, 8)
    if "portrait224" in what:
        # the same 140-token geometry held upright (224 x 160): the FULL model through the landscape_only transposition of the heads
        # (dust3r/utils/misc.py:79-80) and the axis-swapped pointmap the value encoder sees; eval policy
        make_sequence_fixture("spann3r_portrait_224x160x4", 224, 160, 4, False, 4)
    if "batch4" in what:
        # the bench line's `batch4` workload through the reference: four sequences per call (784 rows per launch: the many-row instances
        # of the 224 x 224 step, per-sample banks, ONE similarity decision for the whole batch, spann3r/model.py:113-117); eval policy
        make_sequence_fixture("spann3r_b4_224x5", 224, 224, 5, False, 4, batch=4)
