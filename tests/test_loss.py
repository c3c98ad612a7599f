"""Training criterion: the CPU oracle against the golden dump of the unmodified reference
(tests/golden/loss_conf.npz: loss, details, factor loss, autograd gradients), and the HIP forward / backward against the
oracle (float64 + autograd) on the same seeded inputs."""
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import loss_oracle as LO
from spann3r_amd.weights import synth_loss_case

CASES = ("a", "b", "c")


def _case(g, tag, dtype=torch.float32, device="cpu"):
    seed, fix_first = [int(v) for v in g[tag + "_meta"]]
    alpha, scale = float(g[tag + "_alpha"]), float(g[tag + "_scale"])
    gts, preds_all = synth_loss_case(seed)
    leaves = []
    for r1, r2 in preds_all:
        for r in (r1, r2):
            for k in list(r):
                if scale != 1.0 and k != "conf":
                    r[k] = r[k] * scale
                r[k] = r[k].to(dtype).to(device).requires_grad_(True)
                leaves.append(r[k])
    gts = [{k: (v.to(dtype) if v.is_floating_point() else v).to(device) for k, v in gt.items()} for gt in gts]
    return gts, preds_all, leaves, alpha, bool(fix_first)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_dump(tag):
    g = load_golden("loss_conf.npz")
    gts, preds_all, leaves, alpha, fix_first = _case(g, tag)
    loss, details, factor = LO.conf_loss_t(gts, preds_all, alpha, fix_first)
    assert abs(float(loss) - float(g[tag + "_loss"])) < 2e-6 * abs(float(g[tag + "_loss"]))
    assert abs(float(factor) - float(g[tag + "_factor"])) < 1e-6 + 2e-6 * abs(float(g[tag + "_factor"]))
    for k, v in details.items():
        assert abs(v - float(g[tag + "_detail_" + k])) < 1e-5 * max(1.0, abs(v)), k
    (loss + factor).backward()
    for j, t in enumerate(leaves):
        assert rel_err(t.grad, g[tag + "_grad%d" % j]) < 1e-5, j


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_loss_forward_backward(tag):
    from spann3r_amd.loss import ConfLoss_t, Regr3D_t, L21
    g = load_golden("loss_conf.npz")
    gts, preds_all, leaves, alpha, fix_first = _case(g, tag, device="cuda")
    crit = ConfLoss_t(Regr3D_t(L21, norm_mode="avg_dis", fix_first=fix_first), alpha=alpha)
    loss, details, factor = crit.compute_frame_loss(gts, preds_all)
    assert loss.is_cuda and loss.requires_grad
    # against the reference dump (fp32 CPU) ...
    assert abs(float(loss) - float(g[tag + "_loss"])) < 1e-5 * abs(float(g[tag + "_loss"]))
    assert abs(float(factor) - float(g[tag + "_factor"])) < 1e-6 + 1e-5 * abs(float(g[tag + "_factor"]))
    for k in ("conf_loss_1", "conf_loss2", "conf_mean", "Regr3D_t_pts3d_1", "Regr3D_t_pts3d_2", "Regr3D_tloss_left",
              "Regr3D_tloss_right", "Regr3D_tconf_left", "Regr3D_tconf_right"):
        assert abs(details[k] - float(g[tag + "_detail_" + k])) < 2e-5 * max(1.0, abs(details[k])), k
    (loss + factor).backward()
    for j, t in enumerate(leaves):
        assert rel_err(t.grad.cpu(), g[tag + "_grad%d" % j]) < 2e-5, j
    # ... and against the oracle in float64 with autograd
    gts64, preds64, leaves64, _, _ = _case(g, tag, dtype=torch.float64)
    l64, _, f64 = LO.conf_loss_t(gts64, preds64, alpha, fix_first)
    (l64 + f64).backward()
    assert abs(float(loss) - float(l64)) < 2e-6 * abs(float(l64))
    for a, b in zip(leaves, leaves64):
        assert rel_err(a.grad.cpu(), b.grad) < 5e-6


@pytest.mark.gpu
def test_hip_loss_scales_with_upstream_gradient_and_empty_masks():
    from spann3r_amd.loss import ConfLoss_t, Regr3D_t, L21
    g = load_golden("loss_conf.npz")
    gts, preds_all, leaves, alpha, fix_first = _case(g, "a", device="cuda")
    gts[2]["valid_mask"][:] = False                                   # a frame without ground truth: its entries count 0
    crit = ConfLoss_t(Regr3D_t(L21, norm_mode="avg_dis", fix_first=fix_first), alpha=alpha)
    loss, _, factor = crit.compute_frame_loss(gts, preds_all)
    ((loss + factor) * 0.25).backward()
    gts64, preds64, leaves64, _, _ = _case(g, "a", dtype=torch.float64)
    gts64[2]["valid_mask"][:] = False
    l64, _, f64 = LO.conf_loss_t(gts64, preds64, alpha, fix_first)
    ((l64 + f64) * 0.25).backward()
    assert abs(float(loss) - float(l64)) < 2e-6 * abs(float(l64))
    for a, b in zip(leaves, leaves64):
        assert rel_err(a.grad.cpu(), b.grad) < 5e-6 and torch.isfinite(a.grad).all()


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_test_criterion_oracle_matches_reference_dump(tag):
    """Regr3D_t_ScaleShiftInv (the validation criterion of spann3r/training.py:39)"""
    g = load_golden("loss_conf.npz")
    seed, gsc = [int(v) for v in g[tag + "_meta"]]
    gts, preds_all = synth_loss_case(seed)
    loss, details, factor = LO.regr3d_t_scale_shift_inv(gts, preds_all, gt_scale=bool(gsc))
    assert abs(float(loss) - float(g[tag + "_loss"])) < 3e-6 * abs(float(g[tag + "_loss"]))
    assert abs(float(factor) - float(g[tag + "_factor"])) < 1e-6
    for k, v in details.items():
        assert abs(v - float(g[tag + "_detail_" + k])) < 1e-5 * max(1.0, abs(v)), k


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_hip_test_criterion(tag):
    from spann3r_amd.loss import Regr3D_t_ScaleShiftInv, L21
    g = load_golden("loss_conf.npz")
    seed, gsc = [int(v) for v in g[tag + "_meta"]]
    gts, preds_all = synth_loss_case(seed)
    dev = lambda d: {k: v.cuda() for k, v in d.items()}
    loss, details, factor = Regr3D_t_ScaleShiftInv(L21, gt_scale=bool(gsc)).compute_frame_loss(
        [dev(x) for x in gts], [(dev(a), dev(b)) for a, b in preds_all])
    assert abs(float(loss) - float(g[tag + "_loss"])) < 1e-5 * abs(float(g[tag + "_loss"]))
    assert abs(float(factor) - float(g[tag + "_factor"])) < 1e-6
    for k in ("Regr3D_t_ScaleShiftInv_pts3d_1", "Regr3D_t_ScaleShiftInv_pts3d_2", "Regr3D_t_ScaleShiftInvloss_left",
              "Regr3D_t_ScaleShiftInvloss_right", "Regr3D_t_ScaleShiftInvconf_left", "Regr3D_t_ScaleShiftInvconf_right",
              "gt_shift_z", "pred_shift_z", "gt_scale", "pred_scale"):
        assert abs(details[k] - float(g[tag + "_detail_" + k])) < 2e-5 * max(1.0, abs(details[k])), (k, details[k], float(g[tag + "_detail_" + k]))
