"""Pins the CPU oracle (oracle/spann3r_oracle.py) against dumps of the unmodified reference
(tests/golden/*.npz, produced by tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import spann3r_oracle as O
from spann3r_amd.config import TINY, FULL
from spann3r_amd.weights import synth_frames, state_dict_fingerprint

TOL = 2e-5   # oracle and reference are both torch-CPU fp32; only op grouping differs


def test_fingerprint(tiny_sd):
    g = load_golden("spann3r_tiny.npz")
    assert state_dict_fingerprint(tiny_sd) == float(g["fingerprint"])


def test_tiny_stages_and_outputs(tiny_sd):
    g = load_golden("spann3r_tiny.npz")
    H, W = map(int, g["meta_hw"])
    frames = synth_frames(int(g["meta_frames"]), H, W)
    taps = {}
    preds, preds_all, mem = O.forward(frames, tiny_sd, TINY, return_memory=True, taps=taps)
    worst = 0.0
    for i, s in enumerate(taps["steps"]):
        for k in ("feat1", "feat2", "feat_fuse", "feat_k1", "feat_k2", "cur_v", "pts1", "conf1", "pts2", "conf2"):
            e = rel_err(s[k], g["s%d_%s" % (i, k)])
            worst = max(worst, e)
            assert e < TOL, (i, k, e)
        for side in ("dec1", "dec2"):
            for h in TINY.hooks:
                key = "s%d_%s_%d" % (i, side, h)
                if key in g:
                    e = rel_err(s[side][h], g[key])
                    assert e < TOL, (key, e)
    for j, p in enumerate(preds):
        assert rel_err(p["pts3d" if j == 0 else "pts3d_in_other_view"], g["pred%d_pts" % j]) < TOL
        assert rel_err(p["conf"], g["pred%d_conf" % j]) < TOL
    assert rel_err(mem.mem_k, g["mem_k"]) < TOL and rel_err(mem.mem_v, g["mem_v"]) < TOL
    assert np.array_equal(mem.mem_count.numpy(), g["mem_count"])
    assert rel_err(mem.mem_attn, g["mem_attn"]) < 1e-4
    assert [mem.wm, mem.lm] == list(g["mem_wm_lm"])
    # aliasing contract of the boundary: preds[j] is preds_all[j][0]
    assert preds[0] is preds_all[0][0] and preds[-1] is preds_all[-1][1]


def test_tiny_training_policy(tiny_sd):
    g = load_golden("spann3r_tiny.npz")
    H, W = map(int, g["meta_hw"])
    frames = synth_frames(int(g["meta_frames"]), H, W)
    preds, _, mem = O.forward(frames, tiny_sd, TINY, training_policy=True, return_memory=True)
    for j, p in enumerate(preds):
        assert rel_err(p["pts3d" if j == 0 else "pts3d_in_other_view"], g["train_pred%d_pts" % j]) < TOL
        assert rel_err(p["conf"], g["train_pred%d_conf" % j]) < TOL
    assert rel_err(mem.mem_attn, g["train_mem_attn"]) < 1e-4


def test_tiny_mem_pos_enc(tiny_sd):
    """Spann3R(mem_pos_enc=True): RoPE inside the value-encoder blocks (spann3r/model.py:232-234)"""
    import dataclasses
    g = load_golden("spann3r_tiny.npz")
    H, W = map(int, g["meta_hw"])
    frames = synth_frames(int(g["meta_frames"]), H, W)
    preds, _, mem = O.forward(frames, tiny_sd, dataclasses.replace(TINY, mem_pos_enc=True), return_memory=True)
    for j, p in enumerate(preds):
        assert rel_err(p["pts3d" if j == 0 else "pts3d_in_other_view"], g["mpe_pred%d_pts" % j]) < TOL
        assert rel_err(p["conf"], g["mpe_pred%d_conf" % j]) < TOL
    assert rel_err(mem.mem_v, g["mpe_mem_v"]) < TOL and rel_err(mem.mem_v, g["mem_v"]) > 1e-2       # (it does change the values)


def test_memory_bank(tiny_sd):
    """Stand-alone SpatialMemory: similarity skips, working->long-term hand-over, one prune."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location(
        "make_golden_inputs", os.path.join(os.path.dirname(__file__), "golden", "memory_inputs.py"))
    mi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mi)
    g = load_golden("memory_bank.npz")
    mem = O.SpatialMemoryOracle(tiny_sd)
    events = []
    for step in range(int(g["n_steps"])):
        k, v, q = mi.memory_inputs(step)
        if mem.mem_k is not None:
            o = mem.memory_read(q)
            assert rel_err(o[:, ::7, ::16], g["read%d_sub" % step]) < 1e-4, step
        before = -1 if mem.mem_k is None else mem.mem_k.shape[1]
        mem.add_mem_check(k, v)
        events.append([step, before, mem.mem_k.shape[1], mem.wm, mem.lm])
    assert np.array_equal(np.array(events), g["events"])
    assert np.array_equal(mem.mem_count.numpy(), g["mem_count"])
    assert rel_err(mem.mem_attn, g["mem_attn"]) < 1e-4
    assert rel_err(mem.mem_k[:, :, ::64], g["mem_k_sub"]) < 1e-6


def test_memory_bank_sliding_window(tiny_sd):
    """long_mem_size == 0 (spann3r/model.py:132-137): the bank keeps the last work_mem_size frames.  The oracle against a dump of the
    reference's own SpatialMemory (tests/golden/make_golden.py memory_sliding); tests/test_model_gpu.py pins the HIP path to the oracle"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location(
        "make_golden_inputs", os.path.join(os.path.dirname(__file__), "golden", "memory_inputs.py"))
    mi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mi)
    g = load_golden("memory_bank_sliding.npz")
    mem = O.SpatialMemoryOracle(tiny_sd, long_mem_size=0, work_mem_size=3, sim_thresh=1.0)
    events = []
    for step in range(int(g["n_steps"])):
        k, v, q = mi.memory_inputs(step)
        if mem.mem_k is not None:
            o = mem.memory_read(q)
            assert rel_err(o[:, ::7, ::16], g["read%d_sub" % step]) < 1e-4, step
        mem.add_mem_check(k, v)
        events.append([step, mem.mem_k.shape[1], mem.wm, mem.lm])
    assert np.array_equal(np.array(events), g["events"])
    assert np.array_equal(mem.mem_count.numpy(), g["mem_count"])
    assert rel_err(mem.mem_attn, g["mem_attn"]) < 1e-4
    assert rel_err(mem.mem_k[:, :, ::64], g["mem_k_sub"]) < 1e-6 and rel_err(mem.mem_v[:, :, ::64], g["mem_v_sub"]) < 1e-6


@pytest.mark.slow
def test_full224(full_sd):
    g = load_golden("spann3r_full224.npz")
    assert state_dict_fingerprint(full_sd) == float(g["fingerprint"])
    frames = synth_frames(int(g["meta_frames"]), 224, 224)
    preds, _, mem = O.forward(frames, full_sd, FULL, return_memory=True)
    for j, p in enumerate(preds):
        pts = p["pts3d" if j == 0 else "pts3d_in_other_view"]
        assert rel_err(pts[:, ::4, ::4], g["pred%d_pts_sub" % j]) < 1e-4
        assert rel_err(p["conf"][:, ::4, ::4], g["pred%d_conf_sub" % j]) < 1e-4
    assert rel_err(mem.mem_k[:, ::7, ::16], g["mem_k_sub"]) < 1e-4


def test_true_shape_views(tiny_sd):
    """Views WITH `true_shape` (what every reference caller sends): landscape, and a portrait the dataset rotated to
    landscape (true_shape = transposed image shape: the heads regroup the tokens, the patch embed does not)."""
    g = load_golden("spann3r_trueshape.npz")
    H, W = map(int, g["meta_hw"])
    frames = synth_frames(int(g["meta_frames"]), H, W, batch=int(g["meta_batch"]), seed=int(g["meta_seed"]))
    for tag, ts in (("L", (H, W)), ("P", (W, H))):
        fr = [dict(f, true_shape=torch.tensor([ts, ts], dtype=torch.int32)) for f in frames]
        preds, preds_all, mem = O.forward(fr, tiny_sd, TINY, return_memory=True)
        for j, p in enumerate(preds):
            assert rel_err(p["pts3d" if j == 0 else "pts3d_in_other_view"], g["%s_pred%d_pts" % (tag, j)]) < TOL
            assert rel_err(p["conf"], g["%s_pred%d_conf" % (tag, j)]) < TOL
        for i, (_, r2) in enumerate(preds_all):
            assert rel_err(r2["conf"], g["%s_step%d_conf2" % (tag, i)]) < TOL
        assert rel_err(mem.mem_attn, g["%s_mem_attn" % tag]) < 1e-4


def _check_sequence_fixture(name, full_sd, tol=1e-4):
    g = load_golden(name)
    assert state_dict_fingerprint(full_sd) == float(g["fingerprint"])
    H, W = map(int, g["meta_hw"])
    S = int(g["meta_sub"])
    frames = synth_frames(int(g["meta_frames"]), H, W, batch=int(g["meta_batch"]) if "meta_batch" in g.files else 1)
    taps = {}
    preds, preds_all, mem = O.forward(frames, full_sd, FULL, training_policy=bool(g["meta_train_policy"]),
                                      return_memory=True, taps=taps)
    for j, p in enumerate(preds):
        pts = p["pts3d" if j == 0 else "pts3d_in_other_view"]
        assert rel_err(pts[:, ::S, ::S], g["pred%d_pts_sub" % j]) < tol
        assert rel_err(p["conf"][:, ::S, ::S], g["pred%d_conf_sub" % j]) < tol
    for i, s in enumerate(taps["steps"]):
        for k in ("feat_fuse", "feat_k1", "feat_k2", "cur_v"):
            assert rel_err(s[k][:, ::7, ::16], g["s%d_%s_sub" % (i, k)]) < tol, (i, k)
    assert np.array_equal(mem.mem_count.numpy(), g["mem_count"])
    assert rel_err(mem.mem_attn, g["mem_attn"]) < tol
    assert [mem.wm, mem.lm] == list(g["mem_wm_lm"])


@pytest.mark.slow
def test_cfg2_224x10(full_sd):
    """BASELINE config 2 (the bench workload): 10 frames of 224x224, eval policy."""
    _check_sequence_fixture("spann3r_cfg2_224x10.npz", full_sd)


@pytest.mark.slow
@pytest.mark.parametrize("name", ["spann3r_demo_160x224x6.npz", "spann3r_portrait_224x160x4.npz", "spann3r_mid_288x512x5.npz", "spann3r_b4_224x5.npz"])
def test_other_geometries_and_batch4(full_sd, name):
    """Round-6 fixtures: the FULL model at 140 tokens per frame (landscape and upright: the landscape_only transposition), at 576
    tokens with a growing bank, and the bench line's batch-4 workload (one similarity decision for the whole batch)."""
    _check_sequence_fixture(name, full_sd)


@pytest.mark.slow
def test_cfg3_512x13(full_sd):
    """BASELINE config 3: 13 frames of 512x512, growing bank (train policy, dropout off): 11 reads, up to 11264 tokens."""
    _check_sequence_fixture("spann3r_cfg3_512x13.npz", full_sd)


def test_offline_reconstruction(tiny_sd):
    """demo.py's offline mode: the DUSt3R pair graph (pair_graph mirrors make_pairs + inference) and the next-best-view
    loop, against the dump of the reference's own run."""
    from spann3r_amd.runner import pair_graph
    g = load_golden("spann3r_offline.npz")
    H, W = map(int, g["meta_hw"])
    frames = synth_frames(int(g["meta_frames"]), H, W, seed=int(g["meta_seed"]))
    graph = pair_graph(lambda v1, v2: O.dust3r_forward(v1, v2, tiny_sd, TINY), frames)
    assert graph["view1"]["idx"] == g["graph_idx1"].tolist() and graph["view2"]["idx"] == g["graph_idx2"].tolist()
    assert rel_err(graph["pred1"]["conf"], g["graph_conf1"]) < TOL and rel_err(graph["pred2"]["conf"], g["graph_conf2"]) < TOL
    assert rel_err(graph["pred2"]["pts3d_in_other_view"][:, ::4, ::4], g["graph_pts2_sub"]) < TOL
    assert O.find_initial_pair(graph, len(frames)) == tuple(g["idx_used"][:2].tolist())
    preds, preds_all, used = O.offline_reconstruction(frames, graph, tiny_sd, TINY)
    assert used == g["idx_used"].tolist()
    for j, p in enumerate(preds):
        assert rel_err(p["pts3d" if j == 0 else "pts3d_in_other_view"], g["pred%d_pts" % j]) < TOL
        assert rel_err(p["conf"], g["pred%d_conf" % j]) < TOL
    for i, (_, r2) in enumerate(preds_all):
        assert rel_err(r2["conf"], g["step%d_conf2" % i]) < TOL


def test_tiny_use_feat():
    """oracle with cfg.use_feat (value encoder on dec1[-1], 16 heads of 48) against the reference dump of Spann3R(use_feat=True)"""
    import dataclasses
    from spann3r_amd.weights import synth_state_dict
    g = load_golden("spann3r_usefeat.npz")
    cfg = dataclasses.replace(TINY, use_feat=True)
    sd = synth_state_dict(0, cfg)
    assert state_dict_fingerprint(sd) == float(g["fingerprint"])
    H, W = map(int, g["meta_hw"])
    frames = synth_frames(int(g["meta_frames"]), H, W, batch=int(g["meta_batch"]), seed=int(g["meta_seed"]))
    for tag, tp in (("eval", False), ("train", True)):
        preds, _, mem = O.forward(frames, sd, cfg, training_policy=tp, return_memory=True)
        for j, p in enumerate(preds):
            assert rel_err(p["pts3d" if j == 0 else "pts3d_in_other_view"], g["%s_pred%d_pts" % (tag, j)]) < TOL
            assert rel_err(p["conf"], g["%s_pred%d_conf" % (tag, j)]) < TOL
        assert rel_err(mem.mem_v, g[tag + "_mem_v"]) < TOL


def test_tiny_use_feat_with_mem_pos_enc():
    """oracle with cfg.use_feat AND cfg.mem_pos_enc (RoPE2D on the 48-wide heads of the 768-wide value encoder, spann3r/model.py:225-235,
    :313) against the reference dump of Spann3R(use_feat=True, mem_pos_enc=True)"""
    import dataclasses
    from spann3r_amd.weights import synth_state_dict
    g = load_golden("spann3r_usefeat_mpe.npz")
    cfg = dataclasses.replace(TINY, use_feat=True, mem_pos_enc=True)
    sd = synth_state_dict(0, cfg)
    assert state_dict_fingerprint(sd) == float(g["fingerprint"])
    H, W = map(int, g["meta_hw"])
    frames = synth_frames(int(g["meta_frames"]), H, W, batch=int(g["meta_batch"]), seed=int(g["meta_seed"]))
    preds, _, mem = O.forward(frames, sd, cfg, return_memory=True)
    for j, p in enumerate(preds):
        assert rel_err(p["pts3d" if j == 0 else "pts3d_in_other_view"], g["eval_pred%d_pts" % j]) < TOL
        assert rel_err(p["conf"], g["eval_pred%d_conf" % j]) < TOL
    assert rel_err(mem.mem_v, g["eval_mem_v"]) < TOL and rel_err(mem.mem_k, g["eval_mem_k"]) < TOL
    # the dump exercises the rotation: without it the stored values move far beyond the tolerance
    _, _, mem0 = O.forward(frames, sd, dataclasses.replace(cfg, mem_pos_enc=False), return_memory=True)
    assert rel_err(mem0.mem_v, g["eval_mem_v"]) > 100 * TOL


def test_mixed_orientation_batch(tiny_sd):
    """A batch mixing a landscape view and a portrait the dataset rotated to landscape: the heads run once per orientation and the
    results are scattered back (dust3r/utils/misc.py:80-94); against the reference dump (make_golden.py mixedshape)"""
    g = load_golden("spann3r_mixedshape.npz")
    H, W = map(int, g["meta_hw"])
    frames = synth_frames(int(g["meta_frames"]), H, W, batch=int(g["meta_batch"]), seed=int(g["meta_seed"]))
    frames = [dict(f, true_shape=torch.tensor([(H, W), (W, H)], dtype=torch.int32)) for f in frames]
    preds, preds_all, mem = O.forward(frames, tiny_sd, TINY, return_memory=True)
    for j, p in enumerate(preds):
        assert rel_err(p["pts3d" if j == 0 else "pts3d_in_other_view"], g["M_pred%d_pts" % j]) < TOL
        assert rel_err(p["conf"], g["M_pred%d_conf" % j]) < TOL
    for i, (_, r2) in enumerate(preds_all):
        assert rel_err(r2["conf"], g["M_step%d_conf2" % i]) < TOL
    assert rel_err(mem.mem_attn, g["M_mem_attn"]) < 1e-4
