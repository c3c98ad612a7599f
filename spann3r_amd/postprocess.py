"""Post-forward geometry of demo.py (reference: demo.py:43-72,147-253, dust3r/post_process.py:12-60).

What runs on the device (csrc/postproc.hip): the focal length of the first camera (`estimate_focal_knowing_depth`,
'weiszfeld' mode -- the one demo.py:150 uses) and the confidence filtering of the reconstructed cloud (demo.py:205-211).
What stays on the host, as in the reference: the .ply / transforms.json writers (file formats; Open3D and json there).
Camera poses (demo.py:170-186 calls cv2.solvePnPRansac with every pixel <-> point pair of a frame): `estimate_poses` runs
OpenCV's pipeline -- cv::RNG-drawn 5-point subsets, EPnP hypotheses, consensus at 8 px with the adaptive iteration count, then
SOLVEPNP_ITERATIVE (DLT + Levenberg-Marquardt) on the consensus set -- with the O(H*W) parts as device reductions.  It is
pinned to oracle/pnp_oracle.py, a per-frame restatement of the same OpenCV source files; OpenCV itself is not in this image, so
the last link (oracle vs cv2) waits for tests/golden/make_golden.py pnp on a machine that has it."""
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


# ------------------------------------------------------------------------------------------------ camera poses (demo.py:170-186)
# cv2.solvePnPRansac(points, pixel grid, K, zeros(4)) with OpenCV's defaults, as OpenCV 4.x runs it (calib3d/src/solvepnp.cpp:
# solvePnPRansac; ptsetreg.cpp: RANSACPointSetRegistrator; epnp.cpp; calibration.cpp: cvFindExtrinsicCameraParams2 + CvLevMarq):
#   1. up to 100 hypotheses from 5-point subsets drawn by cv::RNG((uint64)-1), each solved by EPnP            -> host, batched numpy
#   2. consensus of every hypothesis over all H*W points ((float)err^2 <= 64), the best one and the adaptive
#      iteration count (RANSACUpdateNumIters, confidence 0.99) taken in OpenCV's sequential order               -> device (sp3_pnp_score)
#   3. SOLVEPNP_ITERATIVE on the consensus set of the best hypothesis: un-normalised DLT, then <= 20 Levenberg-
#      Marquardt iterations on the reprojection error of that FIXED set                                         -> device sums, 12x12 / 6x6 on the host
# The tests pin this to oracle/pnp_oracle.py (a per-frame restatement of the same OpenCV functions); OpenCV itself is not in the image.
_CV_RNG_COEFF = 4164903690
_FLT_EPS = 1.1920929e-07


def _cv_subsets(n, n_sets, k=5):
    """the first n_sets k-subsets getSubset() draws from cv::RNG((uint64)-1) over n points -> int64 [n_sets, k]"""
    state = 0xFFFFFFFFFFFFFFFF
    out = np.empty((n_sets, k), np.int64)
    for s_ in range(n_sets):
        row = []
        while len(row) < k:
            state = ((state & 0xFFFFFFFF) * _CV_RNG_COEFF + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
            i = (state & 0xFFFFFFFF) % n
            if i not in row:
                row.append(i)
        out[s_] = row
    return out


def _epnp_batch(X, px, f, cx, cy):
    """EPnP (Lepetit et al. 2009, as opencv/modules/calib3d/src/epnp.cpp implements it) for S point sets at once:
    X [S,n,3] object points, px [S,n,2] pixels -> R [S,3,3], t [S,3] (world -> camera).  float64 throughout."""
    S, n, _ = X.shape
    c0 = X.mean(1, keepdims=True)
    P0 = X - c0
    U, dc, _ = np.linalg.svd(np.einsum("sni,snj->sij", P0, P0))
    cws = np.concatenate((c0, c0 + np.sqrt(dc / n)[:, :, None] * np.swapaxes(U, 1, 2)), 1)              # [S,4,3] control points
    CC = np.swapaxes(cws[:, 1:] - cws[:, :1], 1, 2)
    al = np.empty((S, n, 4))
    al[:, :, 1:] = np.einsum("sij,snj->sni", np.linalg.pinv(CC), P0)                                    # barycentric coordinates
    al[:, :, 0] = 1.0 - al[:, :, 1:].sum(2)
    M = np.zeros((S, 2 * n, 12))
    for i in range(4):
        M[:, 0::2, 3 * i], M[:, 0::2, 3 * i + 2] = al[:, :, i] * f, al[:, :, i] * (cx - px[:, :, 0])
        M[:, 1::2, 3 * i + 1], M[:, 1::2, 3 * i + 2] = al[:, :, i] * f, al[:, :, i] * (cy - px[:, :, 1])
    Ue, _, _ = np.linalg.svd(np.einsum("sri,srj->sij", M, M))
    v = np.stack([Ue[:, :, 11 - i] for i in range(4)], 1).reshape(S, 4, 4, 3)                            # null-space end, as control points
    pa, pb = np.array([0, 0, 0, 1, 1, 2]), np.array([1, 2, 3, 2, 3, 3])
    dv = v[:, :, pa] - v[:, :, pb]                                                                       # [S,4,6,3]
    G = np.einsum("sirk,sjrk->srij", dv, dv)                                                             # [S,6,4,4]
    L = np.stack((G[..., 0, 0], 2 * G[..., 0, 1], G[..., 1, 1], 2 * G[..., 0, 2], 2 * G[..., 1, 2], G[..., 2, 2],
                  2 * G[..., 0, 3], 2 * G[..., 1, 3], 2 * G[..., 2, 3], G[..., 3, 3]), -1)               # [S,6,10]
    rho = ((cws[:, pa] - cws[:, pb]) ** 2).sum(-1)                                                       # [S,6]

    def lsq(A, b):
        return np.einsum("sij,sj->si", np.linalg.pinv(A), b)

    def sq(x, neg):                                   # sqrt(-x) where the branch says the sign is negative, sqrt(x) where positive, else 0
        return np.sqrt(np.where(neg, np.maximum(-x, 0), np.maximum(x, 0)))
    with np.errstate(divide="ignore", invalid="ignore"):
        b4 = lsq(L[:, :, [0, 1, 3, 6]], rho)
        n1 = b4[:, 0] < 0
        b0 = sq(b4[:, 0], n1)
        be1 = np.stack((b0, *[np.where(n1, -b4[:, k], b4[:, k]) / b0 for k in (1, 2, 3)]), 1)
        b3 = lsq(L[:, :, [0, 1, 2]], rho)
        n2 = b3[:, 0] < 0
        be2 = np.stack((sq(b3[:, 0], n2) * np.where(b3[:, 1] < 0, -1.0, 1.0), sq(b3[:, 2], n2), np.zeros(S), np.zeros(S)), 1)
        b5 = lsq(L[:, :, [0, 1, 2, 3, 4]], rho)
        n3 = b5[:, 0] < 0
        b50 = sq(b5[:, 0], n3) * np.where(b5[:, 1] < 0, -1.0, 1.0)
        be3 = np.stack((b50, sq(b5[:, 2], n3), b5[:, 3] / b50, np.zeros(S)), 1)
    best_R, best_t, best_e = np.tile(np.eye(3), (S, 1, 1)), np.zeros((S, 3)), np.full(S, np.inf)
    for be in (be1, be2, be3):
        be = np.where(np.isfinite(be), be, 0.0)
        for _ in range(5):                                                                               # gauss_newton
            A = np.stack((2 * L[..., 0] * be[:, None, 0] + L[..., 1] * be[:, None, 1] + L[..., 3] * be[:, None, 2] + L[..., 6] * be[:, None, 3],
                          L[..., 1] * be[:, None, 0] + 2 * L[..., 2] * be[:, None, 1] + L[..., 4] * be[:, None, 2] + L[..., 7] * be[:, None, 3],
                          L[..., 3] * be[:, None, 0] + L[..., 4] * be[:, None, 1] + 2 * L[..., 5] * be[:, None, 2] + L[..., 8] * be[:, None, 3],
                          L[..., 6] * be[:, None, 0] + L[..., 7] * be[:, None, 1] + L[..., 8] * be[:, None, 2] + 2 * L[..., 9] * be[:, None, 3]), -1)
            q = np.stack((be[:, 0] ** 2, be[:, 0] * be[:, 1], be[:, 1] ** 2, be[:, 0] * be[:, 2], be[:, 1] * be[:, 2], be[:, 2] ** 2,
                          be[:, 0] * be[:, 3], be[:, 1] * be[:, 3], be[:, 2] * be[:, 3], be[:, 3] ** 2), 1)
            be = be + lsq(A, rho - np.einsum("srk,sk->sr", L, q))
        ccs = np.einsum("si,sijk->sjk", be, v)                                                           # camera-frame control points
        pcs = np.einsum("snj,sjk->snk", al, ccs)
        sgn = np.where(pcs[:, 0, 2] < 0, -1.0, 1.0)[:, None, None]
        pcs = pcs * sgn
        pc0, pw0 = pcs.mean(1), X.mean(1)
        Uh, _, Vht = np.linalg.svd(np.einsum("sni,snj->sij", pcs - pc0[:, None], X - pw0[:, None]))      # Horn: R = U V^T
        R = Uh @ Vht
        neg = np.linalg.det(R) < 0
        R[neg, 2] = -R[neg, 2]
        t = pc0 - np.einsum("sij,sj->si", R, pw0)
        Xc = np.einsum("sij,snj->sni", R, X) + t[:, None]
        with np.errstate(divide="ignore", invalid="ignore"):
            e = np.sqrt((px[:, :, 0] - (cx + f * Xc[:, :, 0] / Xc[:, :, 2])) ** 2 + (px[:, :, 1] - (cy + f * Xc[:, :, 1] / Xc[:, :, 2])) ** 2).sum(1) / n
        e = np.where(np.isfinite(e), e, np.inf)
        take = e < best_e                                                                                # strict: the first candidate wins ties
        best_R[take], best_t[take], best_e[take] = R[take], t[take], e[take]
    return best_R, best_t


def _expm_so3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def _logm_so3(R):
    c = min(max((np.trace(R) - 1) * 0.5, -1.0), 1.0)
    th = math.acos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(w) * 0.5
    return w * 0.5 if s < 1e-9 else w * (0.5 * th / s)


def _update_num_iters(p, ep, model_points, max_iters):
    """RANSACUpdateNumIters (ptsetreg.cpp)"""
    num = max(1.0 - min(max(p, 0.0), 1.0), 2.2250738585072014e-308)
    denom = 1.0 - (1.0 - min(max(ep, 0.0), 1.0)) ** model_points
    if denom < 2.2250738585072014e-308:
        return 0
    num, denom = math.log(num), math.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(round(num / denom))


def estimate_poses(pts_all, focal, pp, reproj_thresh=8.0, iterations=100, confidence=0.99, chunk=16):
    """pts_all fp32 [F,H,W,3] on the device (every frame's pointmap in the first camera's frame), focal, pp = (cx, cy) ->
    poses float64 [F,4,4] camera-to-world (= inv(extrinsic), what demo.py:186 appends) and the consensus fractions [F].
    Stands where `cv2.solvePnPRansac(pts, pixel_grid, K, 0)` + Rodrigues + inv stand in demo.py:170-186 and follows OpenCV's
    pipeline stage by stage (see the block comment above): the same subsets, hypotheses, consensus test and iteration count, then a
    Levenberg-Marquardt refinement that reaches the same MINIMUM on the fixed consensus set but not through the same iterates --
    the pose is perturbed on the left (omega, delta) here, CvLevMarq damps (rvec, tvec), so the stopping point differs at the
    1e-5 level.  Parity with cv2 itself is unpinned (no OpenCV in the image: oracle/pnp_oracle.py is the anchor);
    non-finite points are left out, as a caller of OpenCV would have to."""
    if not pts_all.is_cuda:
        raise RuntimeError("estimate_poses runs on the GPU (HIP kernels); there is no CPU path")
    F, H, W, _ = pts_all.shape
    pts = pts_all.contiguous().float()
    dev = pts.device
    f = float(np.float32(focal.item() if hasattr(focal, "item") else focal))              # K is passed as float32 in demo.py:175
    cx, cy = (float(np.float32(v)) for v in (pp.tolist() if hasattr(pp, "tolist") else pp))
    if not (f > 0 and math.isfinite(f)):
        raise ValueError("estimate_poses: focal must be positive and finite, got %r" % f)
    lib = L.load()
    flat = pts.reshape(F, H * W, 3)
    fin = torch.isfinite(flat).all(-1)
    nfin = fin.sum(1).cpu().numpy()
    if (nfin <= 5).any():
        raise RuntimeError("estimate_poses: frame %d has no usable points" % int(np.argmax(nfin <= 5)))
    order = torch.argsort((~fin).to(torch.uint8), dim=1, stable=True)                      # finite points first, pixel order kept
    thr = float(reproj_thresh)
    # ---- 1 + 2: hypotheses in chunks, consensus on the device, OpenCV's sequential bookkeeping on the host
    best = [None] * F                           # (count, R, t)
    niters = [max(int(iterations), 1)] * F
    done_upto = 0
    while done_upto < max(niters):
        n_h = min(chunk, max(niters) - done_upto)
        sets = np.stack([_cv_subsets(int(nfin[j]), done_upto + n_h)[done_upto:] for j in range(F)])      # [F,n_h,5] ranks among finite points
        pix = torch.gather(order, 1, torch.from_numpy(sets.reshape(F, -1)).to(dev))                        # -> pixel indices
        X = torch.gather(flat, 1, pix[:, :, None].expand(-1, -1, 3)).double().cpu().numpy().reshape(F * n_h, 5, 3)
        pixh = pix.cpu().numpy().reshape(F * n_h, 5)
        px = np.stack((pixh % W, pixh // W), -1).astype(np.float64)
        R, t = _epnp_batch(X, px, f, cx, cy)
        hyp = np.concatenate((R.reshape(F, n_h, 9), t.reshape(F, n_h, 3)), 2)
        hyp = np.where(np.isfinite(hyp).all(2, keepdims=True), hyp, 0.0).astype(np.float32)
        hyp_d = torch.from_numpy(hyp).to(dev)
        counts = torch.zeros(F, n_h, dtype=torch.int32, device=dev)
        L.check(lib.sp3_pnp_score(pts.data_ptr(), F, H, W, f, cx, cy, hyp_d.data_ptr(), n_h, thr, 1, counts.data_ptr(), L.stream_ptr()),
                "sp3_pnp_score")
        cnt = counts.cpu().numpy()
        for j in range(F):
            for h in range(n_h):
                if done_upto + h >= niters[j]:
                    break
                good = int(cnt[j, h])
                if good > max(best[j][0] if best[j] else 0, 4):
                    best[j] = (good, hyp[j, h, :9].reshape(3, 3).astype(np.float64), hyp[j, h, 9:].astype(np.float64))
                    niters[j] = _update_num_iters(confidence, (nfin[j] - good) / nfin[j], 5, niters[j])
        done_upto += n_h
    if any(b is None for b in best):
        raise RuntimeError("estimate_poses: no consensus for frame %d" % [b is None for b in best].index(True))
    Rm = torch.from_numpy(np.stack([np.concatenate((b[1].reshape(-1), b[2])) for b in best]).astype(np.float32)).to(dev)
    inl = np.array([b[0] for b in best], np.float64) / (H * W)
    # ---- 3: SOLVEPNP_ITERATIVE on the consensus set: DLT ...
    norm = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev).repeat(F, 1).contiguous()
    out41 = torch.empty(F, 41, dtype=torch.float64, device=dev)
    out29 = torch.empty(F, 29, dtype=torch.float64, device=dev)
    L.check(lib.sp3_pnp_dlt_accum(pts.data_ptr(), F, H, W, f, cx, cy, norm.data_ptr(), Rm.data_ptr(), thr, 1, out41.data_ptr(), L.stream_ptr()),
            "sp3_pnp_dlt_accum")
    acc = out41.cpu().numpy()
    poses = []
    for j in range(F):
        S_, Sx, Sy, Sr = (_sym4(acc[j, 10 * k:10 * k + 10]) for k in range(4))
        Z = np.zeros((4, 4))
        _, _, Vt = np.linalg.svd(np.block([[S_, Z, -Sx], [Z, S_, -Sy], [-Sx, -Sy, Sr]]))
        RRt = Vt[11].reshape(3, 4).copy()
        if np.linalg.det(RRt[:, :3]) < 0:
            RRt = -RRt
        sc = np.linalg.norm(RRt[:, :3])
        U, _, Vt2 = np.linalg.svd(RRt[:, :3])
        R0 = U @ Vt2
        poses.append((R0, RRt[:, 3] * (np.linalg.norm(R0) / sc)))
    # ... then CvLevMarq (lambda = 10^k scaling the diagonal of J^T J, k from -3; accept / retry on the error norm; <= 20 iterations,
    # relative parameter change below FLT_EPSILON).  All frames advance in lock step: one device pass evaluates J^T J, J^T e and
    # the squared error of every frame's current or trial pose.
    def evaluate(ps):
        h = np.stack([np.concatenate((R.reshape(-1), t)) for R, t in ps]).astype(np.float64)
        Rt = torch.from_numpy(h).to(dev)
        L.check(lib.sp3_pnp_gn_accum(pts.data_ptr(), F, H, W, f, cx, cy, Rt.data_ptr(), Rm.data_ptr(), thr, out29.data_ptr(), L.stream_ptr()),
                "sp3_pnp_gn_accum")
        return out29.cpu().numpy()

    def lm_step(R, t, Hm, g, lam_lg10):
        A = Hm.copy()
        A[np.diag_indices(6)] *= 1.0 + math.exp(lam_lg10 * math.log(10.0))
        d = -np.linalg.lstsq(A, g, rcond=None)[0]
        E = _expm_so3(d[:3])
        return E @ R, E @ t + d[3:]
    st = [dict(state="J", lam=-3, iters=0, cur=poses[j], prev=None, prev_norm=None, H=None, g=None) for j in range(F)]
    while any(s_["state"] != "done" for s_ in st):
        acc = evaluate([s_["cur"] for s_ in st])
        for j, s_ in enumerate(st):
            if s_["state"] == "done":
                continue
            a = acc[j]
            if s_["state"] == "J":
                Hm = np.zeros((6, 6))
                Hm[np.triu_indices(6)] = a[:21]
                s_["H"], s_["g"] = Hm + np.triu(Hm, 1).T, a[21:27].copy()
                if s_["iters"] == 0:
                    s_["prev_norm"] = math.sqrt(a[27])
                s_["prev"] = s_["cur"]
                s_["cur"] = lm_step(*s_["prev"], s_["H"], s_["g"], s_["lam"])
                s_["state"] = "E"
                continue
            err_norm = math.sqrt(a[27])
            if err_norm > s_["prev_norm"]:
                s_["lam"] += 1
                if s_["lam"] <= 16:
                    s_["cur"] = lm_step(*s_["prev"], s_["H"], s_["g"], s_["lam"])
                    continue
            s_["lam"] = max(s_["lam"] - 1, -16)
            s_["iters"] += 1
            p_new = np.concatenate((_logm_so3(s_["cur"][0]), s_["cur"][1]))
            p_old = np.concatenate((_logm_so3(s_["prev"][0]), s_["prev"][1]))
            if s_["iters"] >= 20 or np.linalg.norm(p_new - p_old) / max(np.linalg.norm(p_old), 1e-300) < _FLT_EPS:
                s_["state"] = "done"
            else:
                s_["prev_norm"], s_["state"] = err_norm, "J"
    out = np.tile(np.eye(4), (F, 1, 1))
    for j, s_ in enumerate(st):
        ext = np.eye(4)
        ext[:3, :3], ext[:3, 3] = s_["cur"]
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
