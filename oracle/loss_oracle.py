"""CPU oracle of the training criterion ConfLoss_t(Regr3D_t(L21, norm_mode='avg_dis', fix_first=...), alpha)
(SURVEY.md §8f-1).  TEST INFRASTRUCTURE ONLY: imported by tests/ (never by the product path).

Restates, in plain differentiable torch (any dtype; the tests run it in float64 and let autograd give the reference
gradients of the HIP backward):
  * spann3r/loss.py:129-178  Regr3D_t.get_all_pts3d_t   (points into view 1's camera, validity, which prediction is which)
  * spann3r/loss.py:20-84    get_norm_factor / normalize_pointcloud_t ('avg_dis': mean distance to the origin; the
                             denominator is the number of valid points of the WHOLE batch, :50)
  * spann3r/loss.py:182-243  Regr3D_t.compute_frame_loss (entry order L0, L1, R1, L2, R2, ..., R_{n-1}; factor loss, whose
                             `filter_factor - gt_factor` broadcasts the k filtered factors against all B ground-truth factors)
  * spann3r/loss.py:246-285  ConfLoss_t.compute_frame_loss (loss * conf - alpha * log conf, mean per entry, x 2, mean)
  * dust3r/losses.py:52-56   L21 = Euclidean distance;  dust3r/inference.py:87-109 get_pred_pts3d;
    dust3r/utils/geometry.py:40-68 geotrf;  dust3r/utils/misc.py:112-121 invalid_to_zeros
Pinned against tests/golden/loss_conf.npz (the unmodified reference on seeded inputs: loss, details, factor loss and the
autograd gradients w.r.t. every prediction), tests/test_loss.py."""
import torch


def entry_plan(n):
    """(frame index, side, step) of every loss entry in the reference's order; side 0 = preds_all[step][0] ('left')"""
    plan = []
    for i in range(n):
        if i != n - 1:
            plan.append((i, 0, i))
        if i != 0:
            plan.append((i, 1, i - 1))
    return plan


def pred_points(pred):
    return pred["pts3d"] if "pts3d" in pred else pred["pts3d_in_other_view"]


def conf_loss_t(gts, preds_all, alpha, fix_first=False, dist_clip=None):
    """-> (loss, details, factor_loss) as ConfLoss_t(Regr3D_t(L21, 'avg_dis', fix_first=fix_first), alpha)
    .compute_frame_loss(gts, preds_all)"""
    n = len(gts)
    dt = pred_points(preds_all[0][0]).dtype
    T0 = torch.linalg.inv(gts[0]["camera_pose"].to(dt))
    gt = [torch.einsum("bij,bhwj->bhwi", T0[:, :3, :3], g["pts3d"].to(dt)) + T0[:, None, None, :3, 3] for g in gts]
    valid = []
    for g in gts:
        v = g["valid_mask"].clone()
        if dist_clip is not None:
            v = v & (g["pts3d"].norm(dim=-1) <= dist_clip)
        valid.append(v)
    pl = [pred_points(preds_all[i][0]) for i in range(n - 1)]
    pr = [pred_points(preds_all[i - 1][1]) for i in range(1, n)]

    def factor(pts):
        tot, cnt = 0, 0
        for i, p in enumerate(pts):
            tot = tot + (p.norm(dim=-1) * valid[i]).flatten(1).sum(1)
            cnt = cnt + valid[i].sum()
            if fix_first:
                break
        return (tot / (cnt + 1e-8)).clip(min=1e-8)[:, None, None, None]
    fp = factor(pl + [pr[-1]])
    fg = factor(gt)
    losses, confs = [], []
    for f, side, step in entry_plan(n):
        p = (pl[f] if side == 0 else pr[f - 1]) / fp
        m = valid[f]
        losses.append((p[m] - (gt[f] / fg)[m]).norm(dim=-1))
        confs.append(preds_all[step][side]["conf"][m])
    filt = fp[fp > fg]
    factor_loss = (filt - fg).abs().mean() if len(filt) > 0 else 0.0
    cl, conf_sum = [], 0
    for e, c in zip(losses, confs):
        conf_sum = conf_sum + c.mean()
        x = e * c - alpha * torch.log(c)
        cl.append(x.mean() if x.numel() > 0 else x.sum())
    cl = torch.stack(cl) * 2.0
    f = lambda t: float(t.detach())
    details = dict(conf_loss_1=f(cl[0]), conf_loss2=f(cl[1]), conf_mean=f(conf_sum / len(losses)),
                   Regr3D_t_pts3d_1=f(losses[0].mean()), Regr3D_t_pts3d_2=f(losses[1].mean()),
                   Regr3D_tloss_left=sum(f(losses[e].mean()) for e, (fr, side, _) in enumerate(entry_plan(n)) if side == 0 and fr != 0),
                   Regr3D_tloss_right=sum(f(losses[e].mean()) for e, (fr, side, _) in enumerate(entry_plan(n)) if side == 1 and fr != n - 1),
                   Regr3D_tconf_left=sum(f(preds_all[st][0]["conf"].mean()) for fr, side, st in entry_plan(n) if side == 0 and fr != 0),
                   Regr3D_tconf_right=sum(f(preds_all[st][1]["conf"].mean()) for fr, side, st in entry_plan(n) if side == 1 and fr != n - 1))
    return cl.mean(), details, factor_loss


def regr3d_t_scale_shift_inv(gts, preds_all, gt_scale=True, fix_first=True):
    """Regr3D_t_ScaleShiftInv(L21, gt_scale=...).compute_frame_loss (spann3r/loss.py:182-243,292-368; the test criterion of
    spann3r/training.py:39): avg_dis normalisation, joint median depth shift (:301-315), joint median-centre / median-norm
    scale (:338-357, get_joint_pointcloud_center_scale :108-126), then the SUM over the entries of the mean Euclidean error.
    -> (loss, details, factor_loss)"""
    n = len(gts)
    dt = pred_points(preds_all[0][0]).dtype
    T0 = torch.linalg.inv(gts[0]["camera_pose"].to(dt))
    gt = [torch.einsum("bij,bhwj->bhwi", T0[:, :3, :3], g["pts3d"].to(dt)) + T0[:, None, None, :3, 3] for g in gts]
    valid = [g["valid_mask"].clone() for g in gts]
    pl = [pred_points(preds_all[i][0]) for i in range(n - 1)]
    pr = [pred_points(preds_all[i - 1][1]) for i in range(1, n)]

    def factor(pts):
        tot, cnt = 0, 0
        for i, p in enumerate(pts):
            tot = tot + (p.norm(dim=-1) * valid[i]).flatten(1).sum(1)
            cnt = cnt + valid[i].sum()
            if fix_first:
                break
        return (tot / (cnt + 1e-8)).clip(min=1e-8)[:, None, None, None]
    fp = factor(pl + [pr[-1]])
    pl, pr = [p / fp for p in pl], [p / fp for p in pr]
    factor_loss = 0.0
    if not gt_scale:                                  # :170-173: with gt_scale the ground truth keeps its metric scale
        fg = factor(gt)
        filt = fp[fp > fg]
        factor_loss = (filt - fg).abs().mean() if len(filt) > 0 else 0.0
        gt = [g / fg for g in gt]
    nan = lambda t, m: torch.where(m.unsqueeze(-1) if t.dim() == 4 else m, t, torch.full_like(t, float("nan")))

    def joint_depth(zs):
        return torch.nanmedian(torch.cat([nan(z, valid[i]).flatten(1) for i, z in enumerate(zs)], -1), dim=-1).values
    gsz = joint_depth([g[..., 2] for g in gt])[:, None, None]
    psz = joint_depth([p[..., 2] for p in pl] + [pr[-1][..., 2]])[:, None, None]
    shift = lambda t, s: torch.cat((t[..., :2], t[..., 2:] - s[..., None]), -1)
    gt = [shift(g, gsz) for g in gt]
    pl, pr = [shift(p, psz) for p in pl], [shift(p, psz) for p in pr]

    def center_scale(pts):
        allp = torch.cat([nan(p, valid[i]).flatten(1, 2) for i, p in enumerate(pts)], 1)
        c = torch.nanmedian(allp, dim=1, keepdim=True).values
        return torch.nanmedian((allp - c).norm(dim=-1), dim=1).values[:, None, None, None]
    gs = center_scale(gt)
    ps = center_scale(pl + [pr[-1]]).clip(min=1e-3, max=1e3)
    if gt_scale:
        pl, pr = [p * (gs / ps) for p in pl], [p * (gs / ps) for p in pr]
    else:
        pl, pr = [p * (ps / gs) for p in pl], [p * (ps / gs) for p in pr]
        gt = [g * (gs / ps) for g in gt]
    losses = []
    for f, side, step in entry_plan(n):
        p = pl[f] if side == 0 else pr[f - 1]
        m = valid[f]
        losses.append((p[m] - gt[f][m]).norm(dim=-1).mean())
    fl = lambda t: float(t.detach())
    details = dict(Regr3D_t_ScaleShiftInv_pts3d_1=fl(losses[0]), Regr3D_t_ScaleShiftInv_pts3d_2=fl(losses[1]),
                   gt_shift_z=fl(gsz.mean()), pred_shift_z=fl(psz.mean()), gt_scale=fl(gs.mean()), pred_scale=fl(ps.mean()))
    return sum(losses), details, factor_loss
