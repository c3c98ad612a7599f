"""CPU restatement of `cv2.solvePnPRansac(objectPoints, imagePoints, K, distCoeffs=0)` with OpenCV's defaults, the call
demo.py:170-186 makes per frame.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else).

OpenCV is a third-party dependency of the reference that is absent from /root/reference and from this image (no network), so
its algorithm is restated here from the published OpenCV 4.x sources, function by function:

  modules/calib3d/src/solvepnp.cpp   solvePnPRansac(): model_points = 5, kernel SOLVEPNP_EPNP, iterationsCount = 100,
                                     reprojectionError = 8.0, confidence = 0.99, flags = SOLVEPNP_ITERATIVE; the consensus
                                     set of the best hypothesis is re-solved by solvePnP(..., SOLVEPNP_ITERATIVE);
                                     PnPRansacCallback::runKernel / computeError (squared pixel error, float)
  modules/calib3d/src/ptsetreg.cpp   RANSACPointSetRegistrator::run / getSubset / findInliers (err <= thresh^2),
                                     RANSACUpdateNumIters; the sampler is cv::RNG((uint64)-1) (modules/core/include/opencv2/core.hpp,
                                     multiply-with-carry, coefficient 4164903690)
  modules/calib3d/src/epnp.cpp       epnp::compute_pose: control points from the PCA of the object points, barycentric
                                     coordinates, M^T M null space, betas by approximations 1-3 + 5 Gauss-Newton steps each, the
                                     candidate with the smallest reprojection error (Lepetit, Moreno-Noguer, Fua, IJCV 2009)
  modules/calib3d/src/calibration.cpp cvFindExtrinsicCameraParams2: DLT initial pose (non-planar branch) and 20 iterations of
                                     CvLevMarq (modules/calib3d/src/compat_ptsetreg.cpp) on the reprojection error, eps = FLT_EPSILON;
                                     cvRodrigues2 / cvProjectPoints2 for the pose parametrisation and its Jacobian

PARITY UNPINNED against OpenCV itself: there is no cv2 here to run, and the reference holds no golden pose.  What pins this file:
ground-truth poses of synthetic scenes (tests/test_postprocess.py) and, on a machine that has OpenCV, tests/golden/make_golden.py pnp
-> tests/golden/pnp_cv2.npz (test_pnp_oracle_vs_opencv then compares the two bit-for-bit in the sampler and to 1e-6 in the pose).
"""
import math

import numpy as np

FLT_EPSILON = 1.1920929e-07
DBL_MIN = 2.2250738585072014e-308


class CvRNG:
    """cv::RNG: state = (uint32)state * 4164903690 + (state >> 32); next() returns the low 32 bits"""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state if state else 0xFFFFFFFF

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else a + self.next() % (b - a)


def rodrigues(r):
    """rotation vector -> (R, dR/dr as [3][3,3]); cvRodrigues2"""
    r = np.asarray(r, np.float64).reshape(3)
    th = float(np.linalg.norm(r))
    K = lambda v: np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], np.float64)
    if th < DBL_MIN ** 0.5:
        return np.eye(3), [K(e) for e in np.eye(3)]
    k = r / th
    c, s = math.cos(th), math.sin(th)
    R = c * np.eye(3) + (1 - c) * np.outer(k, k) + s * K(k)
    # d R / d r_i = (r_i [r]x + [r x (I - R) e_i]x) R / theta^2   (the closed form of the derivative cvRodrigues2 tabulates)
    dR = [(r[i] * K(r) + K(np.cross(r, (np.eye(3) - R)[:, i]))) @ R / (th * th) for i in range(3)]
    return R, dR


def rodrigues_inv(R):
    """rotation matrix -> rotation vector (cvRodrigues2, matrix branch: R is first projected on SO(3) by an SVD)"""
    U, _, Vt = np.linalg.svd(np.asarray(R, np.float64))
    R = U @ Vt
    rx, ry, rz = R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]
    s = math.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = min(max((R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5, -1.0), 1.0)
    th = math.acos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        x = math.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        y = math.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        z = math.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and (R[1, 2] > 0) != (y * z > 0):
            z = -z
        v = np.array([x, y, z])
        return v * (th / np.linalg.norm(v))
    return np.array([rx, ry, rz]) * (0.5 / s) * th


def project(X, r, t, K, jac=False):
    """cvProjectPoints2 without distortion: pixels [n,2] (and d pixel / d (r, t) [2n,6])"""
    R, dR = rodrigues(r)
    Xc = X @ R.T + t
    z = Xc[:, 2]
    z = np.where(z == 0, 1.0, z)                                   # (cvProjectPoints2: z = z ? 1./z : 1)
    iz = 1.0 / z
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x, y = Xc[:, 0] * iz, Xc[:, 1] * iz
    p = np.stack((x * fx + cx, y * fy + cy), 1)
    if not jac:
        return p
    n = len(X)
    J = np.zeros((2 * n, 6))
    # d / dt
    J[0::2, 3], J[0::2, 5] = fx * iz, -fx * x * iz
    J[1::2, 4], J[1::2, 5] = fy * iz, -fy * y * iz
    # d / dr: dXc = dR_i X
    for i in range(3):
        d = X @ dR[i].T
        J[0::2, i] = fx * (d[:, 0] - x * d[:, 2]) * iz
        J[1::2, i] = fy * (d[:, 1] - y * d[:, 2]) * iz
    return p, J


# ------------------------------------------------------------------------------------------------ EPnP (epnp.cpp)
def _lstsq_svd(A, b):
    """cvSolve(..., CV_SVD)"""
    return np.linalg.lstsq(A, b, rcond=None)[0]


def epnp(X, px, K):
    """epnp::compute_pose on n >= 4 correspondences: X [n,3] object points, px [n,2] pixels -> (R, t)"""
    X, px = np.asarray(X, np.float64), np.asarray(px, np.float64)
    n = len(X)
    fu, fv, uc, vc = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    # choose_control_points: centroid + principal directions scaled by sqrt(singular value / n)
    cws = np.zeros((4, 3))
    cws[0] = X.mean(0)
    P0 = X - cws[0]
    U, dc, _ = np.linalg.svd(P0.T @ P0)
    for i in range(3):
        cws[i + 1] = cws[0] + math.sqrt(dc[i] / n) * U[:, i]
    # compute_barycentric_coordinates
    CC = (cws[1:] - cws[0]).T
    al = np.zeros((n, 4))
    al[:, 1:] = (np.linalg.pinv(CC) @ (X - cws[0]).T).T                   # (cvInvert(&CC, &CC_inv, CV_SVD))
    al[:, 0] = 1.0 - al[:, 1:].sum(1)
    # fill_M
    M = np.zeros((2 * n, 12))
    for i in range(4):
        M[0::2, 3 * i], M[0::2, 3 * i + 2] = al[:, i] * fu, al[:, i] * (uc - px[:, 0])
        M[1::2, 3 * i + 1], M[1::2, 3 * i + 2] = al[:, i] * fv, al[:, i] * (vc - px[:, 1])
    Ue, _, _ = np.linalg.svd(M.T @ M)                              # columns sorted by descending singular value
    v = [Ue[:, 11 - i] for i in range(4)]                          # ut + 12 * 11, + 12 * 10, ...: the null-space end
    # compute_L_6x10 / compute_rho
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    dv = np.array([[v[i][3 * a:3 * a + 3] - v[i][3 * b:3 * b + 3] for (a, b) in pairs] for i in range(4)])    # [4][6][3]
    dot = lambda i, j, r: float(dv[i, r] @ dv[j, r])
    L = np.array([[dot(0, 0, r), 2 * dot(0, 1, r), dot(1, 1, r), 2 * dot(0, 2, r), 2 * dot(1, 2, r), dot(2, 2, r),
                   2 * dot(0, 3, r), 2 * dot(1, 3, r), 2 * dot(2, 3, r), dot(3, 3, r)] for r in range(6)])
    rho = np.array([((cws[a] - cws[b]) ** 2).sum() for (a, b) in pairs])

    def approx1():
        b4 = _lstsq_svd(L[:, [0, 1, 3, 6]], rho)
        if b4[0] < 0:
            b0 = math.sqrt(-b4[0])
            return np.array([b0, -b4[1] / b0, -b4[2] / b0, -b4[3] / b0])
        b0 = math.sqrt(b4[0])
        return np.array([b0, b4[1] / b0, b4[2] / b0, b4[3] / b0])

    def approx2():
        b3 = _lstsq_svd(L[:, [0, 1, 2]], rho)
        if b3[0] < 0:
            be = [math.sqrt(-b3[0]), math.sqrt(-b3[2]) if b3[2] < 0 else 0.0]
        else:
            be = [math.sqrt(b3[0]), math.sqrt(b3[2]) if b3[2] > 0 else 0.0]
        if b3[1] < 0:
            be[0] = -be[0]
        return np.array(be + [0.0, 0.0])

    def approx3():
        b5 = _lstsq_svd(L[:, [0, 1, 2, 3, 4]], rho)
        if b5[0] < 0:
            be = [math.sqrt(-b5[0]), math.sqrt(-b5[2]) if b5[2] < 0 else 0.0]
        else:
            be = [math.sqrt(b5[0]), math.sqrt(b5[2]) if b5[2] > 0 else 0.0]
        if b5[1] < 0:
            be[0] = -be[0]
        return np.array(be + [b5[3] / be[0], 0.0])

    def gauss_newton(be):
        be = be.copy()
        for _ in range(5):
            A = np.stack((2 * L[:, 0] * be[0] + L[:, 1] * be[1] + L[:, 3] * be[2] + L[:, 6] * be[3],
                          L[:, 1] * be[0] + 2 * L[:, 2] * be[1] + L[:, 4] * be[2] + L[:, 7] * be[3],
                          L[:, 3] * be[0] + L[:, 4] * be[1] + 2 * L[:, 5] * be[2] + L[:, 8] * be[3],
                          L[:, 6] * be[0] + L[:, 7] * be[1] + L[:, 8] * be[2] + 2 * L[:, 9] * be[3]), 1)
            q = np.array([be[0] * be[0], be[0] * be[1], be[1] * be[1], be[0] * be[2], be[1] * be[2], be[2] * be[2],
                          be[0] * be[3], be[1] * be[3], be[2] * be[3], be[3] * be[3]])
            be = be + np.linalg.lstsq(A, rho - L @ q, rcond=None)[0]       # (qr_solve: least squares of the 6 x 4 system)
        return be

    def pose(be):
        ccs = sum(be[i] * v[i].reshape(4, 3) for i in range(4))            # compute_ccs
        pcs = al @ ccs                                                     # compute_pcs
        if pcs[0, 2] < 0:                                                  # solve_for_sign
            ccs, pcs = -ccs, -pcs
        pc0, pw0 = pcs.mean(0), X.mean(0)                                  # estimate_R_and_t (Horn)
        ABt = (pcs - pc0).T @ (X - pw0)
        Uh, _, Vht = np.linalg.svd(ABt)
        R = Uh @ Vht
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw0
        Xc = X @ R.T + t                                                   # reprojection_error
        with np.errstate(divide="ignore", invalid="ignore"):
            ue, ve = uc + fu * Xc[:, 0] / Xc[:, 2], vc + fv * Xc[:, 1] / Xc[:, 2]
        return R, t, float(np.sqrt((px[:, 0] - ue) ** 2 + (px[:, 1] - ve) ** 2).sum() / n)

    cands = []
    for ap in (approx1, approx2, approx3):
        try:
            cands.append(pose(gauss_newton(ap())))
        except (ZeroDivisionError, ValueError, np.linalg.LinAlgError):
            cands.append((np.eye(3), np.zeros(3), float("inf")))
    N = 0
    if cands[1][2] < cands[0][2]:
        N = 1
    if cands[2][2] < cands[N][2]:
        N = 2
    return cands[N][0], cands[N][1]


# ------------------------------------------------------------------------------------------------ SOLVEPNP_ITERATIVE
def solve_pnp_iterative(X, px, K, max_iter=20):
    """cvFindExtrinsicCameraParams2(useExtrinsicGuess = false), non-planar branch: DLT + CvLevMarq -> (rvec, tvec)"""
    X, px = np.asarray(X, np.float64), np.asarray(px, np.float64)
    n = len(X)
    mn = np.stack(((px[:, 0] - K[0, 2]) / K[0, 0], (px[:, 1] - K[1, 2]) / K[1, 1]), 1)
    Mc = X.mean(0)
    W = np.linalg.svd((X - Mc).T @ (X - Mc), compute_uv=False)
    if W[2] / W[1] < 1e-3:
        raise ValueError("planar object points: the homography branch of cvFindExtrinsicCameraParams2 is not restated (pointmaps are not planar)")
    Lm = np.zeros((2 * n, 12))
    x, y = -mn[:, 0], -mn[:, 1]
    Lm[0::2, 0:3], Lm[0::2, 3] = X, 1.0
    Lm[1::2, 4:7], Lm[1::2, 7] = X, 1.0
    Lm[0::2, 8:11], Lm[0::2, 11] = x[:, None] * X, x
    Lm[1::2, 8:11], Lm[1::2, 11] = y[:, None] * X, y
    _, _, Vt = np.linalg.svd(Lm.T @ Lm)
    RRt = Vt[11].reshape(3, 4).copy()
    if np.linalg.det(RRt[:, :3]) < 0:
        RRt = -RRt
    sc = np.linalg.norm(RRt[:, :3])
    U, _, Vt2 = np.linalg.svd(RRt[:, :3])
    R = U @ Vt2
    t = RRt[:, 3] * (np.linalg.norm(R) / sc)
    param = np.concatenate((rodrigues_inv(R), t))
    # CvLevMarq(6, 2n, max_iter, FLT_EPSILON, completeSymmFlag = true): lambda = 10^lambdaLg10, J^T J diagonal scaled by (1 + lambda)
    m = px.reshape(-1)
    lam_lg10 = -3
    iters = 0

    def step(prev, JtJ, JtErr):
        A = JtJ.copy()
        A[np.diag_indices(6)] *= 1.0 + math.exp(lam_lg10 * math.log(10.0))
        return prev - _lstsq_svd(A, JtErr)
    while True:
        p, J = project(X, param[:3], param[3:], K, jac=True)
        err = p.reshape(-1) - m
        JtJ, JtErr = J.T @ J, J.T @ err
        prev = param.copy()
        if iters == 0:
            prev_norm = np.linalg.norm(err)
        param = step(prev, JtJ, JtErr)
        while True:                                                          # CHECK_ERR
            err_norm = np.linalg.norm(project(X, param[:3], param[3:], K).reshape(-1) - m)
            if err_norm > prev_norm:
                lam_lg10 += 1
                if lam_lg10 <= 16:
                    param = step(prev, JtJ, JtErr)
                    continue
            break
        lam_lg10 = max(lam_lg10 - 1, -16)
        iters += 1
        if iters >= max_iter or np.linalg.norm(param - prev) / max(np.linalg.norm(prev), DBL_MIN) < FLT_EPSILON:
            break
        prev_norm = err_norm
    return param[:3], param[3:]


# ------------------------------------------------------------------------------------------------ RANSAC (ptsetreg.cpp)
def ransac_update_num_iters(p, ep, model_points, max_iters):
    p, ep = min(max(p, 0.0), 1.0), min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, DBL_MIN)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < DBL_MIN:
        return 0
    num, denom = math.log(num), math.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(round(num / denom))      # cvRound


def solve_pnp_ransac(X, px, K, iterations=100, reproj_err=8.0, confidence=0.99, model_points=5):
    """-> (ok, rvec, tvec, inlier mask [n] bool).  X [n,3], px [n,2] as float32 (the dtypes demo.py passes)"""
    X32, px32 = np.asarray(X, np.float32), np.asarray(px, np.float32)
    K = np.asarray(K, np.float32).astype(np.float64)
    Xd, pd = X32.astype(np.float64), px32.astype(np.float64)
    n = len(X32)
    assert n > model_points
    rng = CvRNG()
    thr = np.float32(reproj_err * reproj_err)
    niters = max(iterations, 1)
    best_mask, best_model, max_good = None, None, 0
    it = 0
    while it < niters:
        it += 1
        # getSubset: model_points distinct indices, rng.uniform(0, count) redrawn on a repeat (checkSubset accepts everything)
        idx = []
        for _ in range(model_points):
            i = rng.uniform(0, n)
            while i in idx:
                i = rng.uniform(0, n)
            idx.append(i)
        try:
            R, t = epnp(Xd[idx], pd[idx], K)                                  # runKernel: solvePnP(..., SOLVEPNP_EPNP)
        except np.linalg.LinAlgError:
            continue
        r = rodrigues_inv(R)
        err = ((project(Xd, r, t, K) - pd) ** 2).sum(1).astype(np.float32)    # computeError: squared L2, float
        mask = err <= thr
        good = int(mask.sum())
        if good > max(max_good, model_points - 1):
            best_mask, best_model, max_good = mask, (r, t), good
            niters = ransac_update_num_iters(confidence, (n - good) / n, model_points, niters)
    if max_good <= 0:
        return False, None, None, np.zeros(n, bool)
    try:
        r, t = solve_pnp_iterative(Xd[best_mask], pd[best_mask], K)           # the consensus set, SOLVEPNP_ITERATIVE
    except (ValueError, np.linalg.LinAlgError):
        r, t = best_model
        return False, r, t, best_mask
    return True, r, t, best_mask


def pose_c2w(pts, focal, pp):
    """what demo.py:163-186 appends for one frame: pts [H,W,3] (pointmap in the first camera's frame), pixel grid, K from
    (focal, pp) -> camera-to-world 4x4 = inv([R | t]) and the inlier fraction"""
    H, W, _ = pts.shape
    uu, vv = np.meshgrid(np.arange(W), np.arange(H))
    px = np.stack((uu, vv), -1).reshape(-1, 2).astype(np.float32)
    X = pts.reshape(-1, 3).astype(np.float32)
    ok = np.isfinite(X).all(1)
    K = np.array([[focal, 0, pp[0]], [0, focal, pp[1]], [0, 0, 1]], np.float64)
    good, r, t, mask = solve_pnp_ransac(X[ok], px[ok], K)
    if r is None:
        raise RuntimeError("solvePnPRansac found no model")
    R, _ = rodrigues(r)
    E = np.eye(4)
    E[:3, :3], E[:3, 3] = R, t
    return np.linalg.inv(E), float(mask.sum()) / (H * W)
