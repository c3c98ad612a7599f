"""Post-forward geometry of demo.py (reference: demo.py:43-72,147-253, dust3r/post_process.py:12-60).

What runs on the device (csrc/postproc.hip): the focal length of the first camera (`estimate_focal_knowing_depth`,
'weiszfeld' mode -- the one demo.py:150 uses) and the confidence filtering of the reconstructed cloud (demo.py:205-211).
What stays on the host, as in the reference: the .ply / transforms.json writers (file formats; Open3D and json there).
Camera poses (demo.py:170-186 calls cv2.solvePnPRansac with every pixel <-> point pair of a frame): OpenCV is not in this
image, so `estimate_poses` is an independent calibrated PnP-RANSAC -- 96 hypotheses from 8-point DLTs (seeded sampling),
scored on the device over all points (reprojection error < 8 px, OpenCV's default), DLT re-solved on the consensus set, then
Gauss-Newton on the inliers' reprojection error, which is what solvePnPRansac's final refinement minimises -- checked against
ground-truth poses of synthetic scenes with gross outliers and against its own numpy oracle.  PARITY
WITH OPENCV IS UNPINNED (no cv2 to compare with): expect the same pose up to the noise of the inlier set, not bit parity."""
import json
import math

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


def _sym4(v):
    m = np.zeros((4, 4))
    m[np.triu_indices(4)] = v
    return m + np.triu(m, 1).T


def _pose_from_dlt(acc, norm):
    """acc: 41 sums of sp3_pnp_dlt_accum; norm = (centroid, scale) -> (R, t) world->camera, or None if degenerate"""
    S, Sx, Sy, Sr = (_sym4(acc[10 * k:10 * k + 10]) for k in range(4))
    Z = np.zeros((4, 4))
    return _pose_from_normal(np.block([[S, Z, -Sx], [Z, S, -Sy], [-Sx, -Sy, Sr]]), norm)


def _pose_from_normal(A, norm):
    if not np.isfinite(A).all():
        return None
    w, V = np.linalg.eigh(A)
    P = V[:, 0].reshape(3, 4)
    c, sc = norm[:3], norm[3]
    T = np.eye(4)
    T[:3, :3] *= sc
    T[:3, 3] = -sc * c
    P = P @ T                                               # back from the Hartley-normalised points
    U, sv, Vt = np.linalg.svd(P[:, :3])
    if np.linalg.det(U @ Vt) < 0:
        P, U = -P, -U
    scale = sv.mean()
    if not np.isfinite(scale) or scale < 1e-12:
        return None
    return U @ Vt, P[:, 3] / scale


def _expm_so3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def estimate_poses(pts_all, focal, pp, reproj_thresh=8.0, gn_iters=10, n_hyp=96, seed=0):
    """pts_all fp32 [F,H,W,3] on the device (every frame's pointmap in the first camera's frame), focal, pp = (cx, cy) ->
    poses float64 [F,4,4] camera-to-world (= inv(extrinsic), what demo.py:186 appends) and inlier fractions [F].
    Stands where `cv2.solvePnPRansac(pts, pixel_grid, K, 0)` + Rodrigues + inv stand in demo.py:170-186 (see the module
    docstring: independent implementation, parity with OpenCV unpinned)."""
    if not pts_all.is_cuda:
        raise RuntimeError("estimate_poses runs on the GPU (HIP kernels); there is no CPU path")
    F, H, W, _ = pts_all.shape
    pts = pts_all.contiguous().float()
    dev = pts.device
    f = float(focal.item() if hasattr(focal, "item") else focal)
    cx, cy = (float(v) for v in (pp.tolist() if hasattr(pp, "tolist") else pp))
    if not (f > 0 and math.isfinite(f)):
        raise ValueError("estimate_poses: focal must be positive and finite, got %r" % f)
    lib = L.load()
    flat = pts.reshape(F, -1, 3)
    fin = torch.isfinite(flat).all(-1, keepdim=True)
    safe = torch.where(fin, flat, torch.zeros_like(flat))
    cnt = fin.sum(1).clamp(min=1)
    cen = safe.sum(1) / cnt
    dist = (torch.where(fin, flat - cen[:, None], torch.zeros_like(flat))).norm(dim=-1).sum(1) / cnt[:, 0]
    norm = torch.cat((cen, (1.7320508 / dist.clamp(min=1e-12))[:, None]), 1).contiguous()      # Hartley: mean distance sqrt(3)
    norm_h = norm.double().cpu().numpy()
    out41 = torch.empty(F, 41, dtype=torch.float64, device=dev)
    out29 = torch.empty(F, 29, dtype=torch.float64, device=dev)
    Rt = torch.zeros(F, 12, device=dev)
    poses = [None] * F

    def dlt(use_rt):
        L.check(lib.sp3_pnp_dlt_accum(pts.data_ptr(), F, H, W, f, cx, cy, norm.data_ptr(), Rt.data_ptr() if use_rt else None,
                                      float(reproj_thresh), out41.data_ptr(), L.stream_ptr()), "sp3_pnp_dlt_accum")
        acc = out41.cpu().numpy()
        for j in range(F):
            if acc[j, 40] >= 6:
                rt = _pose_from_dlt(acc[j], norm_h[j])
                if rt is not None:
                    poses[j] = rt
        push()

    def push():
        h = np.zeros((F, 12), np.float32)
        for j, rt in enumerate(poses):
            if rt is not None:
                h[j, :9], h[j, 9:] = rt[0].reshape(-1), rt[1]
        Rt.copy_(torch.from_numpy(h))

    # RANSAC: n_hyp minimal (8-point) DLTs per frame on the host, consensus counted on the device over ALL points
    rng = np.random.default_rng(seed)
    idx = torch.from_numpy(rng.integers(0, H * W, (F, n_hyp, 8))).to(dev)
    samp = torch.gather(flat, 1, idx.reshape(F, -1, 1).expand(-1, -1, 3)).reshape(F, n_hyp, 8, 3).double().cpu().numpy()
    iu, iv = (idx % W).cpu().numpy(), (idx // W).cpu().numpy()
    hyp = np.zeros((F, n_hyp, 12), np.float32)
    hyp[:, :, 0] = hyp[:, :, 4] = hyp[:, :, 8] = 1.0
    for j in range(F):
        c, sc = norm_h[j, :3], norm_h[j, 3]
        for h in range(n_hyp):
            X = samp[j, h]
            if not np.isfinite(X).all():
                continue
            Xh = np.concatenate(((X - c) * sc, np.ones((8, 1))), 1)
            x, y = (iu[j, h] - cx) / f, (iv[j, h] - cy) / f
            Z0 = np.zeros_like(Xh)
            Am = np.concatenate((np.concatenate((Xh, Z0, -x[:, None] * Xh), 1), np.concatenate((Z0, Xh, -y[:, None] * Xh), 1)), 0)
            rt = _pose_from_normal(Am.T @ Am, norm_h[j])
            if rt is not None:
                hyp[j, h, :9], hyp[j, h, 9:] = rt[0].reshape(-1), rt[1]
    hyp_d = torch.from_numpy(hyp).to(dev)
    counts = torch.zeros(F, n_hyp, dtype=torch.int32, device=dev)
    L.check(lib.sp3_pnp_score(pts.data_ptr(), F, H, W, f, cx, cy, hyp_d.data_ptr(), n_hyp, float(reproj_thresh), counts.data_ptr(), L.stream_ptr()),
            "sp3_pnp_score")
    best = counts.argmax(1).cpu().numpy()                    # first maximum: deterministic
    for j in range(F):
        poses[j] = (hyp[j, best[j], :9].reshape(3, 3).astype(np.float64), hyp[j, best[j], 9:].astype(np.float64))
    push()
    dlt(True)                                                # DLT re-solved on the consensus set of the best hypothesis
    inl = np.zeros(F)
    for _ in range(gn_iters):
        L.check(lib.sp3_pnp_gn_accum(pts.data_ptr(), F, H, W, f, cx, cy, Rt.data_ptr(), float(reproj_thresh), out29.data_ptr(), L.stream_ptr()),
                "sp3_pnp_gn_accum")
        acc = out29.cpu().numpy()
        for j in range(F):
            if poses[j] is None or acc[j, 28] < 6:
                continue
            Hm = np.zeros((6, 6))
            Hm[np.triu_indices(6)] = acc[j, :21]
            Hm = Hm + np.triu(Hm, 1).T
            try:
                d = -np.linalg.solve(Hm + 1e-9 * np.trace(Hm) / 6 * np.eye(6), acc[j, 21:27])
            except np.linalg.LinAlgError:
                continue
            R, t = poses[j]
            E = _expm_so3(d[:3])
            poses[j] = (E @ R, E @ t + d[3:])
            inl[j] = acc[j, 28] / (H * W)
        push()
    out = np.tile(np.eye(4), (F, 1, 1))
    for j, rt in enumerate(poses):
        if rt is None:
            raise RuntimeError("estimate_poses: frame %d has no usable points" % j)
        ext = np.eye(4)
        ext[:3, :3], ext[:3, 3] = rt
        out[j] = np.linalg.inv(ext)
    return out, inl


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
