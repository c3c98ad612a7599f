"""CPU oracle of the post-forward geometry that is built.  TEST INFRASTRUCTURE ONLY.

  * focal_weiszfeld: dust3r/post_process.py:12-60 estimate_focal_knowing_depth(focal_mode='weiszfeld'), numpy float64
    (the pixel grid of dust3r/utils/geometry.py:15-37 xy_grid: [j, i] = (i, j)).
  * confident: demo.py:205-211, conf_sig = (conf - 1) / conf > thresh, boolean indexing.
  * camera poses (demo.py:170-186, cv2.solvePnPRansac): oracle/pnp_oracle.py.
focal_weiszfeld is pinned against tests/golden/postprocess.npz = the unmodified reference function on seeded pointmaps."""
import numpy as np


def focal_weiszfeld(pts3d, pp, iters=10, min_focal=0.0, max_focal=np.inf):
    B, H, W, _ = pts3d.shape
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    pix = np.stack((u, v), -1).reshape(1, -1, 2) - np.asarray(pp, np.float64).reshape(1, 1, 2)
    p = pts3d.reshape(B, -1, 3).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy = p[..., :2] / p[..., 2:3]
    xy = np.where(np.isfinite(xy), xy, 0.0)
    dpx, dxx = (xy * pix).sum(-1), (xy * xy).sum(-1)
    f = dpx.mean(1) / dxx.mean(1)
    for _ in range(iters):
        dis = np.linalg.norm(pix - f.reshape(-1, 1, 1) * xy, axis=-1)
        w = 1.0 / np.clip(dis, 1e-8, None)
        f = (w * dpx).mean(1) / (w * dxx).mean(1)
    base = max(H, W) / (2 * np.tan(np.deg2rad(60) / 2))
    return np.clip(f, min_focal * base, max_focal * base)


def confident(pts_all, conf_all, thresh, images_all=None):
    m = (conf_all - 1) / conf_all > thresh
    return pts_all[m].reshape(-1, 3), (None if images_all is None else images_all[m].reshape(-1, 3))
