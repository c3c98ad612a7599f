"""CPU oracle of the input pipeline that turns decoded RGB frames into the `frames` list Spann3R.forward consumes
(SURVEY.md §8f-3).  TEST INFRASTRUCTURE ONLY: imported by tests/ (never by the product path).

Restates, for the Demo dataset's case (no depth / metadata files, pseudo intrinsics with the principal point at the image
centre):
  * spann3r/datasets/demo.py:30-98  (view construction)
  * dust3r/datasets/base/base_stereo_view_dataset.py:140-194 `_crop_resize_if_necessary` and :63-119 `__getitem__`
    (true_shape, ImgNorm, transpose_to_landscape :215-220)
  * dust3r/datasets/utils/cropping.py:54-121 (rescale / crop bookkeeping; its image arithmetic is Pillow's)
  * dust3r/utils/image.py:23 `ImgNorm` = torchvision ToTensor + Normalize(0.5, 0.5)
The pixel arithmetic lives in a third-party dependency that is not vendored: Pillow (`requirements.txt`: pillow==10.3.0;
this image ships 12.2.0, same algorithm): `Image.resize(..., LANCZOS)` = src/libImaging/Resample.c: separable two-pass
(horizontal, then vertical) convolution, support 3 x max(scale, 1), coefficients normalised in double and rounded to
22-bit fixed point, accumulation in int32 from 2^21, 8-bit intermediate image.  `resample_u8` restates it and is pinned
bit for bit against the installed Pillow (tests/test_preprocess.py)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _sinc(x):
    return 1.0 if x == 0.0 else math.sin(x * math.pi) / (x * math.pi)


def _lanczos(x):
    return _sinc(x) * _sinc(x / 3.0) if -3.0 <= x < 3.0 else 0.0


def lanczos_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box: (bounds [out,2], coeffs [out,ksize] int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """one 8-bit pass along `axis` (0 = vertical, 1 = horizontal) of an [H, W, C] uint8 image"""
    src = img.astype(np.int64)
    out_n = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = out_n
    out = np.zeros(shape, np.uint8)
    for o in range(out_n):
        x0, n = bounds[o]
        acc = np.full(shape[:axis] + shape[axis + 1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc = acc + np.take(src, x0 + x, axis=axis) * int(kk[o, x])
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 0:
            out[o] = v
        else:
            out[:, o] = v
    return out


def resample_u8(img, out_w, out_h):
    """Pillow Image.resize((out_w, out_h), LANCZOS) on an [H, W, 3] uint8 array."""
    H, W, _ = img.shape
    x = img
    if out_w != W:
        x = _pass(x, *lanczos_coeffs(W, out_w), axis=1)
    if out_h != H:
        x = _pass(x, *lanczos_coeffs(H, out_h), axis=0)
    return x


def _o2c(K):                                   # dust3r/utils/geometry.py:233-243
    K = K.copy()
    K[0, 2] += 0.5
    K[1, 2] += 0.5
    return K


def _c2o(K):                                   # :220-230
    K = K.copy()
    K[0, 2] -= 0.5
    K[1, 2] -= 0.5
    return K


def _camera_matrix_of_crop(K, in_res, out_res, scaling=1, offset_factor=0.5):   # cropping.py:82-96
    margins = np.asarray(in_res) * scaling - out_res
    assert np.all(margins >= 0.0)
    offset = offset_factor * margins
    Kc = _o2c(K)
    Kc[:2, :] *= scaling
    Kc[:2, 2] -= offset
    return _c2o(Kc)


def demo_plan(H, W, resolution, rng=None):
    """Bookkeeping of `_crop_resize_if_necessary` (base_stereo_view_dataset.py:140-194) for a Demo view (demo.py:66-70:
    float32 pseudo intrinsics, principal point (W//2, H//2); aug_crop off; for near-square crops with a non-square
    resolution the reference draws `rng.integers(2)` (:174-177): pass the same generator to reproduce it, rng=None keeps the
    resolution as given = the coin landing on 0).  The same float32 / float64 operations in the same order, because the
    final crop offset is a rounded difference of two camera matrices.  resolution = (width, height), width >= height.
    Returns dict(crop0=(l,t,r,b), resize=(w,h), crop1=(l,t,r,b), out=(w,h))."""
    K = np.array([[1.0, 0, W // 2], [0, 1.0, H // 2], [0, 0, 1]], dtype=np.float32)
    cx, cy = K[:2, 2].round().astype(int)
    mx, my = min(cx, W - cx), min(cy, H - cy)
    assert mx > W / 5 and my > H / 5
    l, t, r, b = int(cx - mx), int(cy - my), int(cx + mx), int(cy + my)  # :160-166 crop centred on the principal point
    K = K.copy()
    K[0, 2] -= l                                                        # cropping.py:107-110
    K[1, 2] -= t
    W1, H1 = r - l, b - t
    res = tuple(resolution)
    assert res[0] >= res[1]
    if H1 > 1.1 * W1:                                                   # :170-173 portrait -> transposed resolution
        res = res[::-1]
    elif 0.9 < H1 / W1 < 1.1 and res[0] != res[1]:                      # :174-177 square: (portrait, landscape) at random
        if rng is not None and rng.integers(2):
            res = res[::-1]
    in_res = np.array((W1, H1))
    scale_final = max(np.array(res) / in_res) + 1e-8                    # cropping.py:67
    out_res = np.floor(in_res * scale_final).astype(int)                # :68
    K1 = _camera_matrix_of_crop(K, in_res, out_res, scaling=scale_final)          # :77-78 (no offset: margins are < 1 pixel)
    K2 = _camera_matrix_of_crop(K1, tuple(out_res), res, offset_factor=0.5)       # base_stereo_view_dataset.py:187
    l2, t2 = np.int32(np.round(K1[:2, 2] - K2[:2, 2]))                  # cropping.py:117
    return dict(crop0=(l, t, r, b), resize=(int(out_res[0]), int(out_res[1])),
                crop1=(int(l2), int(t2), int(l2) + res[0], int(t2) + res[1]), out=res)


def view_u8(rgb, resolution, rng=None):
    """the uint8 image `_crop_resize_if_necessary` returns (crop on the principal point, Lanczos resize, centre crop)"""
    H, W, _ = rgb.shape
    p = demo_plan(H, W, resolution, rng)
    l, t, r, b = p["crop0"]
    x = rgb[t:b, l:r]
    x = resample_u8(x, *p["resize"])
    l, t, r, b = p["crop1"]
    return x[t:b, l:r]


def preprocess_view(rgb, resolution, rng=None):
    """decoded RGB frame [H, W, 3] uint8 -> (img float32 [3, h, w] in [-1, 1] rectified to landscape, true_shape int32 [2])
    exactly as Demo -> BaseStereoViewDataset.__getitem__ hands it to the model."""
    x = view_u8(rgb, resolution, rng)
    true_shape = np.int32((x.shape[0], x.shape[1]))
    img = ((x.astype(np.float32) / np.float32(255.0)) - np.float32(0.5)) / np.float32(0.5)     # ToTensor + Normalize
    img = np.ascontiguousarray(img.transpose(2, 0, 1))
    if true_shape[1] < true_shape[0]:                                   # transpose_to_landscape (:215-220)
        img = np.ascontiguousarray(img.swapaxes(1, 2))
    return img, true_shape
