"""Drop-in for the training criterion of `spann3r.loss` (reference: /root/reference/spann3r/loss.py, dust3r/losses.py):

    criterion = ConfLoss_t(Regr3D_t(L21, norm_mode='avg_dis', fix_first=False), alpha=0.4)     # spann3r/training.py:37
    loss, details, loss_factor = criterion.compute_frame_loss(batch, preds_all)                 # training.py:217
    (loss + loss_factor).backward()

The forward and the backward run in the HIP kernels of csrc/loss.hip (sp3_conf_loss_forward / _backward) on stacked
buffers; torch only stacks the per-frame tensors, owns the autograd edge back to the predictions and computes the two
unmasked `conf_left / conf_right` monitoring means.  Supported: L21, norm_mode 'avg_dis', gt_scale False -- the
configuration the reference trains with -- and `Regr3D_t_ScaleShiftInv(L21, gt_scale=...)`, the validation criterion of
training.py:39 (forward only: it runs under torch.no_grad there); anything else raises NotImplementedError.  There is no CPU path."""
import torch

from . import lib as L


class L21Loss:
    """dust3r/losses.py:52-59: Euclidean distance between 3-D points (the only pixel criterion the reference trains with)"""

    def __repr__(self):
        return "L21Loss()"


L21 = L21Loss()


def _pts(pred):
    return pred["pts3d"] if "pts3d" in pred else pred["pts3d_in_other_view"]


def _entries(n):
    """(step, side) of every loss entry in the reference's order L0, L1, R1, ..., R_{n-1} (spann3r/loss.py:199-223)"""
    out = []
    for i in range(n):
        if i != n - 1:
            out.append((i, 0))
        if i != 0:
            out.append((i - 1, 1))
    return out


class _ConfRegr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, P, Cf, G, V, pose0, alpha, fix_first):
        E, B = P.shape[:2]
        n, HW = G.shape[0], G.shape[2]
        lib = L.load()
        ws = torch.empty(int(lib.sp3_conf_loss_ws_bytes(n, B)) // 8 + 1, dtype=torch.float64, device=P.device)
        out = torch.empty(7, device=P.device)
        ent = torch.empty(E, 2, device=P.device)
        L.check(lib.sp3_conf_loss_forward(P.data_ptr(), Cf.data_ptr(), G.data_ptr(), V.data_ptr(), pose0.data_ptr(), n, B, HW,
                                          float(alpha), int(fix_first), ws.data_ptr(), out.data_ptr(), ent.data_ptr(), L.stream_ptr()),
                "sp3_conf_loss_forward")
        ctx.save_for_backward(P, Cf, G, V, pose0, ws)
        ctx.cfg = (n, B, HW, float(alpha), int(fix_first))
        ctx.mark_non_differentiable(ent)
        return out[1].clone(), out[0].clone(), out.detach(), ent

    @staticmethod
    def backward(ctx, g_loss, g_factor, _g_out, _g_ent):
        P, Cf, G, V, pose0, ws = ctx.saved_tensors
        n, B, HW, alpha, fix_first = ctx.cfg
        z = torch.zeros((), device=P.device)
        gs = torch.stack(((g_loss if g_loss is not None else z).float().reshape(()), (g_factor if g_factor is not None else z).float().reshape(())))
        dP, dC = torch.empty_like(P), torch.empty_like(Cf)
        L.check(L.load().sp3_conf_loss_backward(P.data_ptr(), Cf.data_ptr(), G.data_ptr(), V.data_ptr(), pose0.data_ptr(), n, B, HW, alpha,
                                                fix_first, ws.data_ptr(), gs.data_ptr(), dP.data_ptr(), dC.data_ptr(), L.stream_ptr()),
                "sp3_conf_loss_backward")
        return dP, dC, None, None, None, None, None


class Regr3D_t:
    """spann3r/loss.py:129-243.  Holds the configuration; the arithmetic lives in ConfLoss_t.compute_frame_loss (the
    reference's per-pixel intermediate lists never exist here)."""

    def __init__(self, criterion, norm_mode="avg_dis", gt_scale=False, fix_first=True):
        if not isinstance(criterion, L21Loss):
            raise NotImplementedError("Regr3D_t: only the L21 pixel criterion is implemented")
        if norm_mode != "avg_dis" or gt_scale:
            raise NotImplementedError("Regr3D_t: norm_mode='avg_dis', gt_scale=False (the training configuration) only")
        self.criterion, self.norm_mode, self.gt_scale, self.fix_first = criterion, norm_mode, gt_scale, fix_first

    def to(self, *_a, **_k):
        return self


class ConfLoss_t:
    """spann3r/loss.py:246-285: pixel loss weighted by the learned confidence, loss * conf - alpha * log(conf)."""

    def __init__(self, pixel_loss, alpha=1):
        assert alpha > 0
        if not isinstance(pixel_loss, Regr3D_t):
            raise NotImplementedError("ConfLoss_t wraps Regr3D_t")
        self.alpha = alpha                      # mutable: training.py:411 anneals it per epoch
        self.pixel_loss = pixel_loss

    def to(self, *_a, **_k):
        return self

    def get_name(self):
        return "ConfLoss(Regr3D_t(%r))" % (self.pixel_loss.criterion,)

    def compute_frame_loss(self, gts, preds, dist_clip=None, monitor=True):
        """gts: the batch (list of n views with pts3d [B,H,W,3], valid_mask [B,H,W] bool, camera_pose [B,4,4]);
        preds: Spann3R.forward's preds_all (n-1 pairs).  -> (loss, details, loss_factor), as the reference.
        monitor=False: `details` stays empty and nothing is read back to the host (a step captured in a hipGraph)."""
        n = len(gts)
        if n < 2 or len(preds) != n - 1:
            raise ValueError("compute_frame_loss: %d views need %d prediction pairs, got %d" % (n, n - 1, len(preds)))
        ent = _entries(n)
        P = torch.stack([_pts(preds[s][side]).float() for s, side in ent])          # [E,B,H,W,3]
        Cf = torch.stack([preds[s][side]["conf"].float() for s, side in ent])       # [E,B,H,W]
        dev = P.device
        if dev.type != "cuda":
            raise RuntimeError("ConfLoss_t runs on the GPU (HIP kernels); there is no CPU path")
        E, B, H, W = Cf.shape
        G = torch.stack([g["pts3d"].to(dev, torch.float32) for g in gts]).reshape(n, B, H * W, 3).contiguous()
        valid = [g["valid_mask"].to(dev) for g in gts]
        if dist_clip is not None:                                                   # :153-156
            valid = [v & (g["pts3d"].to(dev).norm(dim=-1) <= dist_clip) for v, g in zip(valid, gts)]
        V = torch.stack(valid).reshape(n, B, H * W).contiguous().view(torch.uint8)
        pose0 = gts[0]["camera_pose"].to(dev, torch.float32).reshape(B, 16).contiguous()
        loss, factor, out, per = _ConfRegr.apply(P.reshape(E, B, H * W, 3), Cf.reshape(E, B, H * W), G, V, pose0, self.alpha,
                                                 self.pixel_loss.fix_first)
        if not monitor:
            return loss, {}, factor
        o, per = out.tolist(), per.tolist()                                         # (one sync: the reference floats them too)
        name = "Regr3D_t"
        left = [e for e, (s, side) in enumerate(ent) if side == 0 and s != 0]      # i != 0 (:207)
        right = [e for e, (s, side) in enumerate(ent) if side == 1 and s + 1 != n - 1]   # i != n-1 (:219)
        cm = Cf.detach().flatten(1).mean(1).tolist()                                # unmasked monitoring means (:209,:221)
        details = dict(conf_loss_1=o[2], conf_loss2=o[3], conf_mean=o[4],
                       **{name + "_pts3d_1": o[5], name + "_pts3d_2": o[6],
                          name + "loss_left": sum(per[e][0] for e in left), name + "loss_right": sum(per[e][0] for e in right),
                          name + "conf_left": sum(cm[e] for e in left), name + "conf_right": sum(cm[e] for e in right)})
        return loss, details, factor


def _stack_inputs(gts, preds, with_conf):
    n = len(gts)
    if n < 2 or len(preds) != n - 1:
        raise ValueError("compute_frame_loss: %d views need %d prediction pairs, got %d" % (n, n - 1, len(preds)))
    ent = _entries(n)
    P = torch.stack([_pts(preds[s][side]).float() for s, side in ent])
    if P.device.type != "cuda":
        raise RuntimeError("the criteria run on the GPU (HIP kernels); there is no CPU path")
    E, B, H, W = P.shape[:4]
    dev = P.device
    G = torch.stack([g["pts3d"].to(dev, torch.float32) for g in gts]).reshape(n, B, H * W, 3).contiguous()
    V = torch.stack([g["valid_mask"].to(dev) for g in gts]).reshape(n, B, H * W).contiguous().view(torch.uint8)
    pose0 = gts[0]["camera_pose"].to(dev, torch.float32).reshape(B, 16).contiguous()
    Cf = torch.stack([preds[s][side]["conf"].float() for s, side in ent]) if with_conf else None
    return ent, P.reshape(E, B, H * W, 3).contiguous(), Cf, G, V, pose0


class Regr3D_t_ScaleShiftInv:
    """spann3r/loss.py:292-368 (Regr3D_t_ScaleInv over Regr3D_t_ShiftInv over Regr3D_t): the validation criterion
    `Regr3D_t_ScaleShiftInv(L21, gt_scale=True)` of training.py:39.  Forward only."""

    def __init__(self, criterion, norm_mode="avg_dis", gt_scale=False, fix_first=True):
        if not isinstance(criterion, L21Loss) or norm_mode != "avg_dis":
            raise NotImplementedError("Regr3D_t_ScaleShiftInv: L21 and norm_mode='avg_dis' only")
        self.criterion, self.norm_mode, self.gt_scale, self.fix_first = criterion, norm_mode, gt_scale, fix_first

    def to(self, *_a, **_k):
        return self

    @torch.no_grad()
    def compute_frame_loss(self, gts, preds):
        ent, P, _, G, V, pose0 = _stack_inputs(gts, preds, False)
        E, B, HW = P.shape[:3]
        n = len(gts)
        lib = L.load()
        ws = torch.empty(int(lib.sp3_ssi_loss_ws_bytes(n, B, HW)) // 8 + 1, dtype=torch.float64, device=P.device)
        out = torch.empty(6 + E, device=P.device)
        L.check(lib.sp3_ssi_loss_forward(P.data_ptr(), G.data_ptr(), V.data_ptr(), pose0.data_ptr(), n, B, HW, int(self.fix_first),
                                         int(self.gt_scale), ws.data_ptr(), out.data_ptr(), L.stream_ptr()), "sp3_ssi_loss_forward")
        o = out.tolist()
        name = "Regr3D_t_ScaleShiftInv"
        left = [e for e, (s, side) in enumerate(ent) if side == 0 and s != 0]
        right = [e for e, (s, side) in enumerate(ent) if side == 1 and s + 1 != n - 1]
        cm = [float(preds[s][side]["conf"].float().mean()) for s, side in ent]
        details = {name + "_pts3d_1": o[6], name + "_pts3d_2": o[7], name + "loss_left": sum(o[6 + e] for e in left),
                   name + "loss_right": sum(o[6 + e] for e in right), name + "conf_left": sum(cm[e] for e in left),
                   name + "conf_right": sum(cm[e] for e in right), "gt_shift_z": o[2], "pred_shift_z": o[3], "gt_scale": o[4], "pred_scale": o[5]}
        return out[0].clone(), details, out[1].clone()
