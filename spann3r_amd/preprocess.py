"""Input pipeline: decoded RGB frames -> the `frames` list `Spann3R.forward` consumes, on the MI355X.

Mirrors what the reference's Demo dataset + BaseStereoViewDataset.__getitem__ + the default collate produce for a folder
of images (spann3r/datasets/demo.py:30-98; dust3r/datasets/base/base_stereo_view_dataset.py:63-119,140-194,215-220;
dust3r/datasets/utils/cropping.py:54-121; dust3r/utils/image.py:23): centre crop on the principal point, Lanczos
down-scale so the image covers the target resolution, centre crop to it, ToTensor + Normalize(0.5, 0.5), portraits
rotated to landscape, `true_shape` kept on the CPU.  Image DECODING stays with the caller (PIL / cv2, as in the reference);
everything after it runs in two HIP kernels per image (sp3_preprocess_image): the bytes cross PCIe once, as uint8.

Host part (this file): the integer / float bookkeeping of the crops and the resampling coefficient tables -- Pillow's
`precompute_coeffs` + `normalize_coeffs_8bpc` (src/libImaging/Resample.c), computed once per (input size, output size) in
float64 and cached on the device.  Device part: csrc/preproc.hip.  There is no CPU path: `oracle/preprocess_oracle.py`
(test infrastructure) holds the restatement the kernels are checked against."""
import itertools
import math

import numpy as np
import torch

from . import lib as L

_PRECISION_BITS = 32 - 8 - 2
_tables = {}


def resample_tables(in_size, out_size):
    """(bounds int32 [out, 2], coeffs int32 [out, ksize]) of Pillow's LANCZOS resampler for a whole-axis resize."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 3.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    arg = (x + xmin[:, None] - center[:, None] + 0.5) * (1.0 / fscale)
    with np.errstate(divide="ignore", invalid="ignore"):
        def sinc(v):
            pv = v * math.pi
            return np.where(v == 0.0, 1.0, np.sin(pv) / pv)
        w = np.where((arg >= -3.0) & (arg < 3.0), sinc(arg) * sinc(arg / 3.0), 0.0)
    w = np.where(x < xmax[:, None], w, 0.0)
    # the reference sums left to right in double; np.cumsum does the same sequential additions
    ww = np.cumsum(w, axis=1)[:, -1]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww[:, None] != 0.0, ww[:, None], 1.0), w)
    kk = np.where(k < 0, (-0.5 + k * (1 << _PRECISION_BITS)).astype(np.int64), (0.5 + k * (1 << _PRECISION_BITS)).astype(np.int64))
    kk = np.where(x < xmax[:, None], kk, 0).astype(np.int32)
    return np.stack((xmin, xmax), 1).astype(np.int32), kk


def _device_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    t = _tables.get(key)
    if t is None:
        b, k = resample_tables(in_size, out_size)
        t = _tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).contiguous().to(device), k.shape[1])
    return t


def _o2c(K):
    K = K.copy(); K[0, 2] += 0.5; K[1, 2] += 0.5
    return K


def _c2o(K):
    K = K.copy(); K[0, 2] -= 0.5; K[1, 2] -= 0.5
    return K


def _crop_matrix(K, in_res, out_res, scaling=1, offset_factor=0.5):
    margins = np.asarray(in_res) * scaling - out_res
    if not np.all(margins >= 0.0):
        raise ValueError("crop larger than the image")
    Kc = _o2c(K)
    Kc[:2, :] *= scaling
    Kc[:2, 2] -= offset_factor * margins
    return _c2o(Kc)


def plan_view(H, W, resolution, rng=None):
    """Where the reference's `_crop_resize_if_necessary` cuts and what it resizes to, for a Demo view (float32 pseudo
    intrinsics, principal point (W//2, H//2)).  resolution = (width, height) with width >= height, or an int (square).
    The camera-matrix arithmetic is kept in the reference's dtypes and order: the final offset is a ROUNDED difference.
    Pinned against the reference's own functions (tests/golden/crop_plan.npz).

    Near-square crops (0.9 < H/W < 1.1) with a non-square resolution: the reference picks portrait or landscape with
    `rng.integers(2)` (base_stereo_view_dataset.py:174-177; the dataset's per-item generator).  Pass that generator as `rng`
    to reproduce its draw; with rng=None (inference: demo.py always asks for a square 224 resolution, where the branch is
    never taken) the choice is deterministic: the resolution as given, i.e. landscape."""
    if isinstance(resolution, int):
        resolution = (resolution, resolution)
    K = np.array([[1.0, 0, W // 2], [0, 1.0, H // 2], [0, 0, 1]], dtype=np.float32)
    cx, cy = K[:2, 2].round().astype(int)
    mx, my = min(cx, W - cx), min(cy, H - cy)
    if not (mx > W / 5 and my > H / 5):
        raise ValueError("Bad principal point")
    l, t, r, b = int(cx - mx), int(cy - my), int(cx + mx), int(cy + my)
    K[0, 2] -= l
    K[1, 2] -= t
    W1, H1 = r - l, b - t
    res = tuple(resolution)
    if res[0] < res[1]:
        raise ValueError("resolution must be (width, height) with width >= height")
    if H1 > 1.1 * W1:
        res = res[::-1]
    elif 0.9 < H1 / W1 < 1.1 and res[0] != res[1]:
        if rng is not None and rng.integers(2):
            res = res[::-1]
    in_res = np.array((W1, H1))
    scale_final = max(np.array(res) / in_res) + 1e-8
    out_res = np.floor(in_res * scale_final).astype(int)
    K1 = _crop_matrix(K, in_res, out_res, scaling=scale_final)
    K2 = _crop_matrix(K1, tuple(out_res), res)
    l2, t2 = np.int32(np.round(K1[:2, 2] - K2[:2, 2]))
    return dict(crop0=(l, t, r, b), resize=(int(out_res[0]), int(out_res[1])), crop1=(int(l2), int(t2), int(l2) + res[0], int(t2) + res[1]), out=res)


def preprocess_image(rgb, resolution, device="cuda", rng=None):
    """rgb: uint8 [H, W, 3] (numpy array or torch tensor, host or device) -> (img fp32 [1, 3, h, w] on the device in
    [-1, 1], rectified to landscape; true_shape int32 [1, 2] on the CPU)."""
    t = torch.as_tensor(rgb)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise TypeError("expected a uint8 [H, W, 3] RGB frame")
    src = t.to(device).contiguous()                       # the only host -> device copy: the raw bytes
    H, W, _ = src.shape
    p = plan_view(H, W, resolution, rng)
    l, tt, r, b = p["crop0"]
    W1, H1 = r - l, b - tt
    W2, H2 = p["resize"]
    hb, hk, hks = _device_tables(W1, W2, src.device)
    vb, vk, vks = _device_tables(H1, H2, src.device)
    outW, outH = p["out"]
    l2, t2 = p["crop1"][:2]
    transpose = int(outH > outW)
    tmp = torch.empty(H1, W2, 3, dtype=torch.uint8, device=src.device)
    out = torch.empty((1, 3, outW, outH) if transpose else (1, 3, outH, outW), device=src.device)
    L.check(L.load().sp3_preprocess_image(src.data_ptr(), W * 3, l, tt, H1, W1, hb.data_ptr(), hk.data_ptr(), hks, W2,
                                          vb.data_ptr(), vk.data_ptr(), vks, H2, l2, t2, outW, outH, transpose,
                                          tmp.data_ptr(), out.data_ptr(), L.stream_ptr()), "sp3_preprocess_image")
    return out, torch.tensor([[outH, outW]], dtype=torch.int32)


def frames_from_images(images, resolution=224, device="cuda", kf_every=1):
    """A folder's worth of decoded RGB frames -> the batch demo.py feeds the model (spann3r/datasets/demo.py with
    full_video=True: every kf_every-th image; then the default collate with batch size 1): a list of dicts with
    img [1,3,h,w] on the device, true_shape int32 [1,2] on the CPU, idx, instance."""
    frames = []
    for j, rgb in enumerate(itertools.islice(images, 0, None, kf_every)):     # skipped images are never decoded by a generator
        img, ts = preprocess_image(rgb, resolution, device)
        frames.append(dict(img=img, true_shape=ts, idx=j, instance=str(j)))
    return frames
