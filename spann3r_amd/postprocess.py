"""Post-forward geometry of demo.py (reference: demo.py:43-72,147-253, dust3r/post_process.py:12-60).

What runs on the device (csrc/postproc.hip): the focal length of the first camera (`estimate_focal_knowing_depth`,
'weiszfeld' mode -- the one demo.py:150 uses) and the confidence filtering of the reconstructed cloud (demo.py:205-211).
What stays on the host, as in the reference: the .ply / transforms.json writers (file formats; Open3D and json there).
NOT built: the per-frame camera poses (demo.py:170-186 calls cv2.solvePnPRansac: OpenCV is not in this image, so neither an
implementation nor an oracle of its RANSAC could be pinned) -- `transforms_json` takes poses from the caller."""
import json
import math
import struct

import numpy as np
import torch

from . import lib as L


def estimate_focal_knowing_depth(pts3d, pp, focal_mode="weiszfeld", min_focal=0.0, max_focal=np.inf):
    """pts3d fp32 [B,H,W,3] on the device, pp = (cx, cy) -> focal [B] (device).  dust3r/post_process.py:12-60."""
    if focal_mode != "weiszfeld":
        raise NotImplementedError("estimate_focal_knowing_depth: only focal_mode='weiszfeld' (what demo.py uses) is implemented")
    if not pts3d.is_cuda:
        raise RuntimeError("estimate_focal_knowing_depth runs on the GPU (HIP kernel); there is no CPU path")
    B, H, W, three = pts3d.shape
    assert three == 3
    pts3d = pts3d.contiguous().float()
    ppx, ppy = (float(v) for v in (pp.tolist() if hasattr(pp, "tolist") else pp))
    base = max(H, W) / (2 * math.tan(math.radians(60) / 2))
    fmax = float(min(max_focal * base, 3.0e38))
    out = torch.empty(B, device=pts3d.device)
    L.check(L.load().sp3_focal_weiszfeld(pts3d.data_ptr(), B, H, W, ppx, ppy, 10, float(min_focal * base), fmax, out.data_ptr(),
                                         L.stream_ptr()), "sp3_focal_weiszfeld")
    return out


def confident_points(pts_all, conf_all, conf_thresh, images_all=None):
    """pts_all [..., 3], conf_all [...], images_all [..., 3] or None (device, fp32) -> (points [N,3], colours [N,3] or None):
    `pts_all[conf_sig > t]`, `images_all[conf_sig > t]` with conf_sig = (conf - 1) / conf (demo.py:205-211), order kept."""
    if not pts_all.is_cuda:
        raise RuntimeError("confident_points runs on the GPU (HIP kernels); there is no CPU path")
    conf = conf_all.contiguous().float().reshape(-1)
    n = conf.numel()
    pts = pts_all.contiguous().float().reshape(n, 3)
    rgb = None if images_all is None else images_all.contiguous().float().reshape(n, 3)
    dev = pts.device
    scratch = torch.empty((n + 1023) // 1024, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    out_p = torch.empty(n, 3, device=dev)
    out_c = None if rgb is None else torch.empty(n, 3, device=dev)
    L.check(L.load().sp3_conf_filter(conf.data_ptr(), pts.data_ptr(), L.ptr(rgb), n, float(conf_thresh), scratch.data_ptr(), total.data_ptr(),
                                     out_p.data_ptr(), L.ptr(out_c), L.stream_ptr()), "sp3_conf_filter")
    k = int(total.item())
    return out_p[:k], (None if out_c is None else out_c[:k])


def write_ply(path, points, colors=None):
    """binary little-endian PLY as Open3D's write_point_cloud lays it out (double x/y/z, uchar red/green/blue from
    colours in [0, 1]); demo.py:208-212"""
    pts = np.asarray(points.detach().cpu() if torch.is_tensor(points) else points, dtype=np.float64).reshape(-1, 3)
    head = ["ply", "format binary_little_endian 1.0", "comment Created by spann3r_amd", "element vertex %d" % len(pts),
            "property double x", "property double y", "property double z"]
    if colors is not None:
        col = np.asarray(colors.detach().cpu() if torch.is_tensor(colors) else colors, dtype=np.float64).reshape(-1, 3)
        col = np.clip(np.round(col * 255.0), 0, 255).astype(np.uint8)
        head += ["property uchar red", "property uchar green", "property uchar blue"]
        rec = np.empty(len(pts), dtype=[("p", "<f8", 3), ("c", "u1", 3)])
        rec["p"], rec["c"] = pts, col
    else:
        rec = pts.astype("<f8")
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\nend_header\n").encode())
        f.write(rec.tobytes())


def transforms_json(H, W, focal, poses_all, ply_file_path, ori_path=None):
    """demo.py:43-72 get_transform_json (NeRF-studio style; flips the y / z camera axes of every pose IN PLACE, as there)"""
    f = float(focal.item() if hasattr(focal, "item") else focal)
    d = {"w": W, "h": H, "fl_x": f, "fl_y": f, "cx": W / 2, "cy": H / 2, "k1": 0, "k2": 0, "p1": 0, "p2": 0, "camera_model": "OPENCV"}
    frames = []
    for i, pose in enumerate(poses_all):
        pose[:3, 1] *= -1
        pose[:3, 2] *= -1
        frames.append({"file_path": ("imgs/img_%04d.png" % i) if ori_path is None else ori_path[i], "transform_matrix": pose.tolist()})
    d["frames"] = frames
    d["ply_file_path"] = ply_file_path
    return d


def save_transforms(path, transform_dict):
    with open(path, "w") as f:
        json.dump(transform_dict, f, indent=4)
