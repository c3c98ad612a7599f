#!/usr/bin/env python
"""The timed region of the reference's demo.py (lines 81-132) on synthetic frames: build the model, load a state dict,
move only view['img'] to the device, time model.forward(batch), report FPS = len(batch) / seconds (with the device sync the
reference omits).  The real demo.py cannot be imported in this image (cv2 / open3d / torchvision are not installed)."""
import argparse
import time

import torch

from spann3r_amd import Spann3R, FULL, TINY
from spann3r_amd.weights import synth_state_dict, synth_frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--tiny", action="store_true", help="2-layer encoder / 10-layer decoder geometry (quick look)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--demo_path", default=None, help="folder of images (demo.py --demo_path): decoded with PIL on the host, "
                    "then cropped / Lanczos-resized / normalised on the GPU (spann3r_amd.preprocess)")
    ap.add_argument("--kf_every", type=int, default=10)
    ap.add_argument("--save_ply", default=None, help="write the confidence-filtered cloud here (demo.py:205-212) and print the "
                    "estimated focal of the first camera (demo.py:147-151)")
    ap.add_argument("--conf_thresh", type=float, default=1e-3)
    ap.add_argument("--offline", action="store_true", help="demo.py --offline: DUSt3R pair graph + next-best-view order")
    args = ap.parse_args()
    cfg = TINY if args.tiny else FULL
    device = "cuda"
    model = Spann3R(dus3r_name=None, cfg=cfg, init_weights=False).to(device)      # demo.py:81-82
    model.load_state_dict(synth_state_dict(0, cfg))                               # demo.py:84 (no checkpoint ships)
    model.eval().set_precision(args.precision)
    if args.demo_path:                                                            # demo.py:86-92 (Demo dataset, full_video)
        import os
        import numpy as np
        from PIL import Image
        from spann3r_amd.preprocess import frames_from_images
        names = sorted(n for n in os.listdir(args.demo_path) if n.lower().endswith((".jpg", ".jpeg", ".png")))
        batch = frames_from_images((np.asarray(Image.open(os.path.join(args.demo_path, n)).convert("RGB")) for n in names),
                                   resolution=args.size, device=device, kf_every=args.kf_every)
    else:
        batch = synth_frames(args.frames, args.size, args.size)
    for view in batch if not args.demo_path else []:                                                            # demo.py:94-95
        view["img"] = view["img"].to(device, non_blocking=True)
        view["true_shape"] = torch.tensor(view["img"].shape[2:]).unsqueeze(0)     # demo.py:109 (stays on the CPU)
    if args.offline:                                                              # demo.py:98-121
        from spann3r_amd.runner import pair_graph
        torch.cuda.synchronize()
        start = time.time()
        output = pair_graph(model.dust3r, batch)          # = make_pairs(complete, symmetrize) + inference(batch_size=2)
        preds, preds_all, idx_used = model.offline_reconstruction(batch, output)
        torch.cuda.synchronize()
        end = time.time()
        print("Time: %.4f s, FPS: %.1f  (offline, order %s)" % (end - start, len(batch) / (end - start), idx_used))
        return
    for it in range(args.repeat):
        torch.cuda.synchronize()
        start = time.time()
        preds, preds_all = model.forward(batch)                                   # demo.py:125
        torch.cuda.synchronize()
        end = time.time()
        print("Time: %.4f s, FPS: %.1f  (%d frames, call %d: %s)" %
              (end - start, len(batch) / (end - start), len(batch), it,
               ["eager warm-up", "hipGraph capture", "hipGraph replay"][min(it, 2)]))
    if args.save_ply:
        from spann3r_amd.postprocess import (estimate_focal_knowing_depth, confident_points, write_ply, estimate_poses,
                                             transforms_json, save_transforms)
        _, H, W, _ = preds[0]["pts3d"].shape
        focal = estimate_focal_knowing_depth(preds[0]["pts3d"], torch.tensor((W / 2, H / 2)), focal_mode="weiszfeld")
        print("Estimated focal of first camera: %.3f (%dx%d)" % (focal.item(), W, H))
        pts_all = torch.cat([p["pts3d" if j == 0 else "pts3d_in_other_view"] for j, p in enumerate(preds)])
        conf_all = torch.cat([p["conf"] for p in preds])
        images_all = (torch.cat([v["img"] for v in batch]).permute(0, 2, 3, 1) + 1.0) / 2.0
        if images_all.shape[1:3] != pts_all.shape[1:3]:
            images_all = images_all.swapaxes(1, 2)                                # portrait frames were rectified to landscape
        points, colours = confident_points(pts_all, conf_all, args.conf_thresh, images_all)
        write_ply(args.save_ply, points, colours)
        print("wrote %d of %d points to %s" % (len(points), conf_all.numel(), args.save_ply))
        if focal.item() > 1e-3:                                                 # (seeded random weights give no geometry at all)
            poses_all, inliers = estimate_poses(pts_all, focal, (W / 2, H / 2))     # demo.py:170-186 (PnP-RANSAC per frame)
            import os
            tj = os.path.join(os.path.dirname(os.path.abspath(args.save_ply)), "transforms.json")
            save_transforms(tj, transforms_json(H, W, focal, list(poses_all), os.path.basename(args.save_ply)))
            print("wrote %d camera poses to %s (inlier fractions %.2f..%.2f)" % (len(poses_all), tj, inliers.min(), inliers.max()))
        else:
            print("degenerate pointmaps (no checkpoint loaded): poses / transforms.json skipped")
    pts = preds[-1]["pts3d_in_other_view"]
    print("last frame: pts3d_in_other_view", tuple(pts.shape), "conf mean %.3f" % float(preds[-1]["conf"].mean()))


if __name__ == "__main__":
    main()
