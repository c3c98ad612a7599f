"""Input pipeline.  CPU: the oracle's resampler is pinned bit for bit against the installed Pillow (the
dependency the reference's crop/resize calls into), and the product's host tables / crop plans against the oracle.
GPU: sp3_preprocess_image == the oracle, bit for bit (uint8 stage exact, float stage the same IEEE operations)."""
import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as PO
from spann3r_amd import preprocess as PP

SIZES = [((480, 640), 298, 224), ((375, 500), 512, 384), ((720, 1280), 398, 224), ((97, 131), 45, 33), ((64, 64), 224, 224),
         ((50, 70), 70, 25)]


def _image(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (np.sin(xx / 7.0) + np.cos(yy / 5.0)) * 60 + 128
    img = base[..., None] + rng.normal(0, 40, (h, w, 3))
    img[h // 3: h // 3 + 4] = 255; img[:, w // 2: w // 2 + 3] = 0          # hard edges: exercise the clamp
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("hw,ow,oh", SIZES)
def test_oracle_resampler_matches_pillow(hw, ow, oh):
    Image = pytest.importorskip("PIL.Image")
    img = _image(*hw, seed=ow)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.LANCZOS))
    assert np.array_equal(PO.resample_u8(img, ow, oh), ref)


def test_oracle_view_matches_pillow_pipeline():
    """crop -> resize -> crop with PIL calls in the order of cropping.py, on the oracle's plan"""
    Image = pytest.importorskip("PIL.Image")
    for hw, res in [((480, 640), (224, 224)), ((640, 480), (224, 224)), ((375, 500), (512, 384)), ((700, 500), (512, 384))]:
        rgb = _image(*hw, seed=hw[0])
        p = PO.demo_plan(*hw, res)
        im = Image.fromarray(rgb).crop(p["crop0"]).resize(p["resize"], resample=Image.LANCZOS).crop(p["crop1"])
        ref = (np.asarray(im).astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
        ref = ref.transpose(2, 0, 1)
        img, ts = PO.preprocess_view(rgb, res)
        h, w = ts
        assert (w, h) == p["out"]
        if h > w:
            ref = ref.transpose(0, 2, 1)
        assert img.shape == ref.shape and np.array_equal(img, ref)
        assert img.shape[1] <= img.shape[2]


def test_host_tables_and_plans_match_oracle():
    for a, b in [(640, 298), (480, 224), (500, 513), (374, 384), (1280, 398), (64, 224), (131, 45), (33, 33), (1920, 683)]:
        b1, k1 = PO.lanczos_coeffs(a, b)
        b2, k2 = PP.resample_tables(a, b)
        assert np.array_equal(b1, b2) and np.array_equal(k1, k2), (a, b)
    for hw, res in [((480, 640), 224), ((640, 480), (224, 224)), ((375, 500), (512, 384)), ((1080, 1920), (512, 384)),
                    ((481, 641), (224, 224)), ((1000, 751), (512, 384)), ((1000, 1000), (512, 384)), ((528, 500), (512, 384))]:
        r = (res, res) if isinstance(res, int) else res
        assert PO.demo_plan(*hw, r) == PP.plan_view(*hw, res)
    with pytest.raises(ValueError):
        PP.plan_view(480, 640, (224, 512))
    with pytest.raises(TypeError):
        PP.preprocess_image(np.zeros((4, 4, 3), np.float32), 224, device="cpu")


def _crop_cases():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "crop_plan.npz"))
    tags = sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_hw")})
    return g, tags


def test_crop_plan_equals_the_references_own_functions():
    """f3 pin: tests/golden/crop_plan.npz was written by the reference's unmodified `_crop_resize_if_necessary` +
    cropping.py (make_golden.py crop).  Integer work: every box, the resize target and the uint8 image must be EQUAL --
    for the oracle's restatement and for the product's plan (incl. the rng draw of near-square inputs)."""
    import hashlib
    g, tags = _crop_cases()
    assert len(tags) >= 12
    flips = 0
    for t in tags:
        H, W = (int(v) for v in g[t + "_hw"])
        res = tuple(int(v) for v in g[t + "_res"])
        seed = int(t.split("_s")[1])
        want = dict(crop0=tuple(g[t + "_crop0"].tolist()), resize=tuple(g[t + "_resize"].tolist()), crop1=tuple(g[t + "_crop1"].tolist()))
        for plan in (PO.demo_plan(H, W, res, np.random.default_rng(seed)), PP.plan_view(H, W, res, np.random.default_rng(seed))):
            assert {k: plan[k] for k in want} == want, (t, plan, want)
        flips += int(want["crop1"][2] - want["crop1"][0] != res[0])
        i = int(t[1:].split("_")[0])
        img = PO.view_u8(_image(H, W, seed=H + i), res, np.random.default_rng(seed))
        assert tuple(img.shape) == tuple(g[t + "_img_shape"].tolist())
        c = img[img.shape[0] // 2 - 8: img.shape[0] // 2 + 8, img.shape[1] // 2 - 8: img.shape[1] // 2 + 8]
        assert np.array_equal(c, g[t + "_img_centre"])
        assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).digest() == g[t + "_img_sha256"].tobytes(), t
    assert flips >= 3          # portraits and at least one coin flip are in the fixture


@pytest.mark.gpu
def test_preprocess_kernels_equal_the_reference_output():
    """the device pipeline against the uint8 images the reference's own function produced (same fixture): ImgNorm of a uint8
    image is exactly invertible, so the fp32 output is mapped back to bytes and hashed"""
    import hashlib
    g, tags = _crop_cases()
    for t in tags:
        H, W = (int(v) for v in g[t + "_hw"])
        res = tuple(int(v) for v in g[t + "_res"])
        seed, i = int(t.split("_s")[1]), int(t[1:].split("_")[0])
        img, ts = PP.preprocess_image(_image(H, W, seed=H + i), res, rng=np.random.default_rng(seed))
        x = img[0].cpu().numpy()
        h, w = ts[0].tolist()
        if h > w:                                              # portraits come back rotated to landscape
            x = x.transpose(0, 2, 1)
        u8 = np.rint((x * np.float32(0.5) + np.float32(0.5)) * 255.0).astype(np.uint8).transpose(1, 2, 0)
        assert tuple(u8.shape) == tuple(g[t + "_img_shape"].tolist())
        assert hashlib.sha256(np.ascontiguousarray(u8).tobytes()).digest() == g[t + "_img_sha256"].tobytes(), t


@pytest.mark.gpu
@pytest.mark.parametrize("hw,res", [((480, 640), 224), ((640, 480), 224), ((375, 500), (512, 384)), ((700, 500), (512, 384)),
                                    ((97, 131), 224), ((1080, 1920), (512, 288))])
def test_preprocess_kernels_bit_exact(hw, res):
    rgb = _image(*hw, seed=hw[1])
    r = (res, res) if isinstance(res, int) else res
    ref, ts_ref = PO.preprocess_view(rgb, r)
    img, ts = PP.preprocess_image(rgb, res)
    assert ts.dtype == torch.int32 and not ts.is_cuda and ts.tolist() == [list(ts_ref)]
    assert tuple(img.shape) == (1,) + ref.shape
    assert np.array_equal(img[0].cpu().numpy(), ref)


@pytest.mark.gpu
def test_frames_from_images_feed_the_model(tiny_sd):
    from spann3r_amd import Spann3R, TINY
    rgbs = [_image(120, 160, seed=s) for s in range(5)]
    frames = PP.frames_from_images(rgbs, resolution=64, kf_every=2)
    assert [f["idx"] for f in frames] == [0, 1, 2] and frames[0]["img"].shape == (1, 3, 64, 64)
    model = Spann3R(dus3r_name=None, cfg=TINY, init_weights=False)
    model.load_state_dict(tiny_sd, strict=True)
    model = model.cuda().eval()
    preds, preds_all = model.forward(frames)
    assert len(preds) == 3 and preds[0]["pts3d"].shape == (1, 64, 64, 3) and torch.isfinite(preds[-1]["conf"]).all()
