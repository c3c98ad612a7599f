#!/usr/bin/env python
"""Wall time of one training step of the full model (ViT-L / ViT-B / DPT, seeded random weights) on one MI355X, split into its phases:
train-mode Spann3R.forward (HIP autograd ops) -> ConfLoss_t -> backward -> global-norm clip + AdamW on the flat buckets.
  python tools/train_step_time.py [--precision bf16|fp32] [--batch 4] [--frames 5]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import Spann3R, FULL, TINY
from spann3r_amd import train as T
from spann3r_amd.weights import synth_state_dict
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--size", type=int, default=224)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--tiny", action="store_true")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--precision", default="bf16")
a = ap.parse_args()
cfg = TINY if a.tiny else FULL
m = Spann3R(dus3r_name=None, cfg=cfg, init_weights=False)
m.load_state_dict(synth_state_dict(0, cfg))
m = m.cuda()
ts = T.TrainStep(m, precision=a.precision)
frames, gts = bench.synth_training_batch(1, a.frames, a.size, a.batch, "cuda")
sync = torch.cuda.synchronize
for it in range(a.steps):
    sync(); t0 = time.time()
    ts.reducer.zero_grad(); ts.reducer.prepare()
    preds, preds_all = m(frames)
    sync(); t1 = time.time()
    loss, det, fac = ts.crit.compute_frame_loss(gts, preds_all)
    (loss + fac).backward()
    sync(); t2 = time.time()
    ts.reducer.finish()
    norm = ts.opt.step(max_norm=1.0)
    sync(); t3 = time.time()
    print("step %d (%s): loss %.4f |g| %.3f  forward %.3f s  loss+backward %.3f s  clip+AdamW %.4f s  total %.3f s  (batch %d, %d frames of %dx%d, peak memory %.1f GB)" %
          (it, a.precision, float(loss.detach()) + float(fac), float(norm), t1 - t0, t2 - t1, t3 - t2, t3 - t0, a.batch, a.frames, a.size, a.size,
           torch.cuda.max_memory_allocated() / 2**30))

# the same step captured once into a hipGraph and replayed (TrainStep(graph=True)): what bench.py --train times on one rank
del ts
torch.cuda.empty_cache()
m2 = Spann3R(dus3r_name=None, cfg=cfg, init_weights=False)
m2.load_state_dict(synth_state_dict(0, cfg))
tg = T.TrainStep(m2.cuda(), precision=a.precision, graph=True)
for it in range(a.steps + 1):
    sync(); t0 = time.time()
    loss, norm = tg.run(frames, gts)
    sync(); t1 = time.time()
    print("graph step %d (%s): loss %.4f |g| %.3f  total %.3f s%s" % (it, a.precision, float(loss), float(norm), t1 - t0, "  (warm-up + capture)" if it == 0 else ""))
