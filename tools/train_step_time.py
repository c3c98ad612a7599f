#!/usr/bin/env python
"""Wall time of one training step of the full model (ViT-L / ViT-B / DPT, seeded random weights) on one MI355X:
train-mode Spann3R.forward (HIP autograd ops) -> ConfLoss_t -> backward -> AdamW.  The step is the fp32 parity build."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import Spann3R, FULL, TINY
from spann3r_amd.loss import ConfLoss_t, Regr3D_t, L21
from spann3r_amd.train import AdamW
from spann3r_amd.weights import synth_state_dict, synth_frames

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--size", type=int, default=224)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--tiny", action="store_true")
ap.add_argument("--steps", type=int, default=2)
a = ap.parse_args()
cfg = TINY if a.tiny else FULL
m = Spann3R(dus3r_name=None, cfg=cfg, init_weights=False)
m.load_state_dict(synth_state_dict(0, cfg))
m = m.cuda().train()
crit = ConfLoss_t(Regr3D_t(L21, norm_mode="avg_dis", fix_first=False), alpha=0.4)
opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=5e-5, betas=(0.9, 0.95), weight_decay=0.05)
frames = [{"img": f["img"].cuda()} for f in synth_frames(a.frames, a.size, a.size, batch=a.batch, seed=1)]
g = torch.Generator().manual_seed(0)
gts = []
for i in range(a.frames):
    pose = torch.eye(4).repeat(a.batch, 1, 1)
    gts.append(dict(pts3d=(torch.randn(a.batch, a.size, a.size, 3, generator=g) + torch.tensor([0., 0., 3.])).cuda(),
                    valid_mask=(torch.rand(a.batch, a.size, a.size, generator=g) < 0.9).cuda(), camera_pose=pose.cuda()))
for it in range(a.steps):
    torch.cuda.synchronize(); t0 = time.time()
    preds, preds_all = m(frames)
    torch.cuda.synchronize(); t1 = time.time()
    loss, det, fac = crit.compute_frame_loss(gts, preds_all)
    (loss + fac).backward()
    torch.cuda.synchronize(); t2 = time.time()
    opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t3 = time.time()
    print("step %d: loss %.4f  forward %.2f s  loss+backward %.2f s  AdamW %.3f s  (batch %d, %d frames of %dx%d, peak memory %.1f GB)" %
          (it, float(loss.detach()) + float(fac), t1 - t0, t2 - t1, t3 - t2, a.batch, a.frames, a.size, a.size, torch.cuda.max_memory_allocated() / 2**30))
