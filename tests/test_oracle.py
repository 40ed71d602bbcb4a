"""CPU tests: the oracle (oracle/wb_oracle.py) against the golden fixtures generated from the
reference's own source (tests/golden/make_golden.py) and against SciPy known answers."""
import hashlib
import os

import numpy as np
import pytest
import scipy.ndimage as ndi

import waldboost_amd as wb
from oracle import wb_oracle as orc
from waldboost_amd.synth import synth_image
from util import GOLDEN, f2_cases, f2_meta, golden_meta, oracle_detect, small_cases


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("case", list(small_cases()), ids=lambda c: c[0])
def test_pyramid_matches_reference_bit_exact(case):
    name, img, info, levels = case
    opts = dict(shrink=info["shrink"], n_per_oct=info["n_per_oct"], smooth=info["smooth"], channels=orc.grad_hist)
    got = list(orc.channel_pyramid(img, opts))
    assert len(got) == info["n_levels"]
    for (c, s), ref, rs in zip(got, levels, info["scales"]):
        assert c.dtype == np.float32 and c.shape == ref.shape
        assert s == rs
        assert np.array_equal(c.view(np.uint32), ref.view(np.uint32))


def test_cfg1_detect_matches_reference():
    meta = golden_meta()["cfg1"]
    g = np.load(os.path.join(GOLDEN, "cfg1_640x480.npz"))
    M = wb.load(os.path.join(GOLDEN, "cfg1_d1_T32.pb"))
    img = synth_image(480, 640, 0)
    shape, opts = tuple(M.shape), dict(shrink=2, n_per_oct=8, smooth=1, channels=orc.grad_hist)
    hashes = [sha(c) for c, _ in orc.channel_pyramid(img, opts)]
    assert hashes == meta["chn_sha256"]
    res = oracle_detect(M, img)
    det = g["det"]
    assert res["n_loc"] == meta["n_loc"] == 407350          # SURVEY section 8 geometry
    assert res["n_weak"] == meta["n_weak"]
    assert np.array_equal(res["alive"], g["alive"])
    assert np.array_equal(res["level"], det["level"]) and np.array_equal(res["r"], det["r"]) and np.array_equal(res["c"], det["c"])
    assert np.array_equal(res["scores"].view(np.uint32), det["score"].view(np.uint32))
    assert np.array_equal(res["boxes"], np.stack([det["x1"], det["y1"], det["x2"], det["y2"]], 1))
    assert np.array_equal(np.array(res["scales"]), g["scales"])


def test_mixed_depth_model_matches_reference():
    g = np.load(os.path.join(GOLDEN, "mixed_200x264.npz"))
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    res = oracle_detect(M, g["image"])
    det = g["det"]
    assert res["n_loc"] == int(g["n_loc"]) and res["n_weak"] == int(g["n_weak"])
    assert np.array_equal(res["alive"], g["alive"])
    assert np.array_equal(res["r"], det["r"]) and np.array_equal(res["c"], det["c"]) and np.array_equal(res["level"], det["level"])
    assert np.array_equal(res["scores"].view(np.uint32), det["score"].view(np.uint32))


def test_all_rejecting_stage_breaks_early():
    g = np.load(os.path.join(GOLDEN, "reject_200x264.npz"))
    img = np.load(os.path.join(GOLDEN, "mixed_200x264.npz"))["image"]
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    M.theta = [float(t) for t in g["theta"]]
    res = oracle_detect(M, img)
    assert res["scores"].size == 0
    assert np.array_equal(res["alive"], g["alive"])
    assert res["n_weak"] == int(g["n_weak"]) and res["n_loc"] == int(g["n_loc"])
    assert (res["alive"][:, 5:] == 0).all()


def test_empty_model_keeps_every_window():
    g = np.load(os.path.join(GOLDEN, "empty_40x56.npz"))
    opts = dict(shrink=2, n_per_oct=8, smooth=1, channels=orc.grad_hist)
    res = orc.detect((12, 12, 4), opts, [], [], g["image"])
    assert np.array_equal(res["boxes"], g["boxes"]) and np.array_equal(res["scores"], g["scores"])
    assert res["n_loc"] == int(g["n_loc"]) and res["n_weak"] == 0


# ---- the other channel functions: fpga.grad_hist_4_u1 / grad_mag_u1, grad_mag ----
@pytest.mark.parametrize("case", list(f2_cases()), ids=lambda c: c[0])
def test_f2_pyramid_matches_reference_bit_exact(case):
    name, img, info, levels = case
    opts = dict(shrink=info["shrink"], n_per_oct=info["n_per_oct"], smooth=info["smooth"], channels=info["channels"])
    got = list(orc.channel_pyramid(img, opts))
    assert len(got) == info["n_levels"]
    for (c, s), ref, rs in zip(got, levels, info["scales"]):
        assert c.dtype == ref.dtype == np.dtype(info["dtype"]) and c.shape == ref.shape
        assert s == rs
        assert np.array_equal(c.view(np.uint8), ref.view(np.uint8))


def test_f2_uint8_channel_pool_wraps():
    """The 'edgy' fixtures exist to exercise the uint8 wrap of avg_pool_2 on channel values."""
    for name, img, info, levels in f2_cases():
        if "edgy" in name and info["shrink"] == 2 and info["smooth"] == 0:
            full = orc.CHANNEL_FUNCS[info["channels"]](orc.resize_bilinear(img, *[2 * x for x in levels[0].shape[:2]]))
            s = full[0::2, 0::2].astype(np.int32) + full[1::2, 0::2] + full[0::2, 1::2] + full[1::2, 1::2]
            assert (s > 255).any(), name
            return
    raise AssertionError("no wrapping fixture")


@pytest.mark.parametrize("fn", ["grad_hist_4_u1", "grad_mag_u1", "grad_mag"])
def test_f2_detect_matches_reference(fn):
    meta = f2_meta()
    g = np.load(os.path.join(GOLDEN, f"{fn}_200x264.npz"))
    proto = wb.model_pb2.Model()
    import zlib
    proto.ParseFromString(zlib.decompress(open(os.path.join(GOLDEN, f"{fn}_d2_T24.pb"), "rb").read()))
    assert proto.channel_opts.func == meta["names"][fn]
    trees = [orc.make_tree(np.array(w.feature, np.uint8).reshape(-1, 3), np.array(w.threshold, np.float32),
                           np.array(w.left, np.int8), np.array(w.right, np.int8), np.array(w.prediction, np.float32))
             for w in proto.classifier]
    opts = dict(shrink=proto.channel_opts.shrink, n_per_oct=proto.channel_opts.n_per_oct,
                smooth=proto.channel_opts.smooth, channels=fn)
    hashes = [sha(c) for c, _ in orc.channel_pyramid(g["image"], opts)]
    assert hashes == meta[fn]["chn_sha256"]
    res = orc.detect(tuple(proto.shape), opts, trees, list(proto.theta), g["image"])
    det = g["det"]
    assert res["n_loc"] == int(g["n_loc"]) and res["n_weak"] == int(g["n_weak"])
    assert np.array_equal(res["alive"], g["alive"])
    assert np.array_equal(res["r"], det["r"]) and np.array_equal(res["c"], det["c"]) and np.array_equal(res["level"], det["level"])
    assert np.array_equal(res["scores"].view(np.uint32), det["score"].view(np.uint32))
    assert np.array_equal(res["boxes"], np.stack([det["x1"], det["y1"], det["x2"], det["y2"]], 1))


def test_grad_mag_equals_scipy():
    from scipy.ndimage import convolve1d
    rng = np.random.default_rng(5)
    for shape in [(37, 53), (8, 9), (3, 3), (120, 7)]:
        img = rng.integers(0, 256, shape).astype(np.uint8)
        im = img.astype("f")
        H, D = np.array([1, 2, 1], "f4"), np.array([-1, 0, 1], "f4")
        gy = convolve1d(convolve1d(im, H, axis=1), D, axis=0)
        gx = convolve1d(convolve1d(im, H, axis=0), D, axis=1)
        mag = np.sqrt(gx ** 2 + gy ** 2)
        K = orc.triangle_kernel(5)
        nrm = convolve1d(mag, K, axis=0)
        convolve1d(nrm, K, axis=1, output=nrm)
        mag /= nrm + 1e-3
        assert np.array_equal(orc.grad_mag(img).view(np.uint32), mag[..., None].view(np.uint32))


# ---- training-time callers: gather_samples, Model.predict, DTree.apply ----
@pytest.mark.parametrize("tag,pb,npz", [("f32", "mixed_d2_T24.pb", "mixed_200x264.npz"),
                                        ("u8", "grad_hist_4_u1_d2_T24.pb", "grad_hist_4_u1_200x264.npz")])
def test_f4_samples_match_reference(tag, pb, npz):
    from util import oracle_model
    z = np.load(os.path.join(GOLDEN, "samples_f4.npz"))
    M = wb.load(os.path.join(GOLDEN, pb))
    shape, opts, trees, thetas = oracle_model(M)
    levels = list(orc.channel_pyramid(np.load(os.path.join(GOLDEN, npz))["image"], opts))
    for li in (0, 5):
        k = f"{tag}/L{li}"
        X = orc.gather_samples(levels[li][0], z[f"{k}/rs"], z[f"{k}/cs"], shape)
        assert X.dtype == z[f"{k}/X"].dtype and np.array_equal(X, z[f"{k}/X"])
        H, mask = orc.model_predict(shape, trees, thetas, X)
        assert np.array_equal(mask, z[f"{k}/mask"]) and mask.any() and not mask.all()
        assert np.array_equal(H.view(np.uint32), z[f"{k}/H"].view(np.uint32))
        for t in (0, 3, 10):
            assert np.array_equal(orc.tree_apply(trees[t], X), z[f"{k}/apply{t}"])
            assert np.array_equal(trees[t]["prediction"][orc.tree_apply(trees[t], X)], z[f"{k}/predict{t}"])
    assert orc.gather_samples(levels[0][0], np.zeros(0, int), np.zeros(0, int), shape).shape == (0,) + shape


# ---- SciPy known-answer tests (SciPy ships on the GPU box as well) -------------------------
@pytest.mark.parametrize("shape,out", [((1080 // 4, 1920 // 4), (990 // 4 * 1, 1760 // 4)), ((97, 131), (80, 110)),
                                       ((64, 96), (64, 96)), ((33, 60), (24, 42)), ((135, 240), (134, 240))])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_resize_equals_scipy_zoom(shape, out, dtype):
    img = synth_image(shape[0], shape[1], 7, dtype)
    src = img.astype(np.float64) if dtype == np.uint8 else img
    z = ndi.zoom(src, [o / i for o, i in zip(out, shape)], order=1, mode="mirror", grid_mode=True)
    assert z.shape == out
    z = np.clip(z, img.min(), img.max())
    want = z.astype(dtype)
    got = orc.resize_bilinear(img, out[0], out[1])
    assert got.dtype == dtype and np.array_equal(got, want)


def nonfinite_image(shape, dtype, kind, seed=5):
    """A float image with a few non-finite (or nearly overflowing) pixels: what reference channels.py:132 does with
    them is scipy's zoom -- every tap is multiplied, 0 * inf = NaN -- and NumPy's clip."""
    img = (synth_image(shape[0], shape[1], seed).astype(dtype) * dtype(0.25)).astype(dtype)
    big = np.finfo(np.float32).max
    if kind == "inf":
        img[7, 11] = np.inf
        img[shape[0] - 1, 3] = -np.inf                   # on the border: the mirrored tap
        img[20:22, 30:33] = np.inf
    elif kind == "nan":
        img[9, 14] = np.nan
    elif kind == "huge":
        img[5:9, 8:12] = big
        img[17, 40] = -big
        img[30, 2] = big * dtype(0.75)
    return img


@pytest.mark.parametrize("kind", ["inf", "nan", "huge"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape,out", [((64, 96), (64, 96)), ((97, 131), (80, 110)), ((66, 90), (36, 50))])
def test_resize_of_non_finite_pixels_equals_scipy_zoom(shape, out, dtype, kind):
    """Pins S3 off the beaten track: identity levels are NOT a copy next to an infinite pixel (its zero-weight taps
    give NaN), a NaN pixel makes min / max and with them the whole clipped level NaN."""
    img = nonfinite_image(shape, dtype, kind)
    with np.errstate(invalid="ignore", over="ignore"):
        z = ndi.zoom(img, [o / i for o, i in zip(out, shape)], order=1, mode="mirror", grid_mode=True)
        want = np.clip(z, img.min(), img.max())
        got = orc.resize_bilinear(img, out[0], out[1])
    assert got.dtype == dtype and np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.array_equal(got[ok], want[ok])
    if kind == "inf":
        assert np.isinf(want).any() and (np.isnan(want).any() or shape != out)      # identity: the zero-weight taps
    if kind == "nan":
        assert np.isnan(want).all()


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_gradients_equal_scipy_convolve1d(dtype):
    img = synth_image(75, 101, 3, dtype).astype("f")
    if dtype == np.float32:
        img = img * np.float32(1.2345) + np.float32(1e-3)
    H = np.array([1, 2, 1], "f4")
    D = np.array([-1, 0, 1], "f4")
    gy = ndi.convolve1d(ndi.convolve1d(img, H, axis=1), D, axis=0)
    gx = ndi.convolve1d(ndi.convolve1d(img, H, axis=0), D, axis=1)
    ogx, ogy = orc.gradients(img)
    assert np.array_equal(ogx, gx) and np.array_equal(ogy, gy)


def test_uint8_octave_wraps_like_numpy():
    img = np.full((16, 16), 200, np.uint8)
    o = list(orc.image_octaves(img))
    assert o[1][0, 0] == ((800 & 255) >> 2)          # bright regions corrupt octaves >= 1 (reference bug, S2)
    ref = ((img[0::2, 0::2] + img[1::2, 0::2] + img[0::2, 1::2] + img[1::2, 1::2]) / 4).astype(np.uint8)
    assert np.array_equal(o[1], ref)


def test_level_plan_geometry_of_the_survey():
    p = orc.level_plan(1080, 1920, 2, 8)
    assert len(p) == 64
    assert [(l["nh"], l["nw"]) for l in p[:3]] == [(1080, 1920), (990, 1760), (908, 1614)]
    n_loc = sum(max(l["nh"] // 2 - 12, 0) * max(l["nw"] // 2 - 12, 0) for l in p)
    assert n_loc == 3045278
    assert sum(l["nh"] * l["nw"] for l in p) == 13007444


@pytest.mark.parametrize("case", list(__import__("util").dtype_cases()), ids=lambda c: c[0])
def test_oracle_pyramids_of_other_image_dtypes_vs_reference_fixture(case):
    """float64 and integer images (reference channels.py:122 keeps image.dtype): every level bit-exact."""
    name, img, info, levels = case
    assert str(img.dtype) == info["dtype"]
    opts = dict(shrink=info["shrink"], n_per_oct=info["n_per_oct"], smooth=info["smooth"], channels=orc.grad_hist)
    with np.errstate(over="ignore"):
        got = list(orc.channel_pyramid(img, opts))
    assert len(got) == info["n_levels"]
    for (c, s), ref, rs in zip(got, levels, info["scales"]):
        assert s == rs and c.dtype == np.float32 and c.shape == ref.shape
        assert np.array_equal(c.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("case", list(__import__("util").chanfunc_arg_cases()), ids=lambda c: c[0])
def test_oracle_channel_functions_with_arguments_vs_reference_fixture(case):
    name, img, func, kwargs, ref = case
    got = getattr(orc, func)(img, **kwargs)
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
