"""GPU parity of the other channel functions the reference ships:
waldboost.fpga.grad_hist_4_u1 / grad_mag_u1 as channel_opts["channels"] -- uint8 channels through
the channel kernel and the cascade -- against fixtures generated from the reference's own source
(tests/golden/make_golden_f2.py) and against the CPU oracle.  Everything is integer: bit-exact.
"""
import os

import numpy as np
import pytest

import waldboost_amd as wb
from oracle import wb_oracle as orc
from waldboost_amd.synth import synth_image, random_tree_arrays
from util import GOLDEN, f2_cases, f2_meta, oracle_detect

pytestmark = pytest.mark.gpu

FUNCS = {"grad_hist_4_u1": wb.fpga.grad_hist_4_u1, "grad_mag_u1": wb.fpga.grad_mag_u1}
U1_CASES = [c for c in f2_cases() if c[2]["channels"] in FUNCS]
GM_CASES = [c for c in f2_cases() if c[2]["channels"] == "grad_mag"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("case", U1_CASES, ids=lambda c: c[0])
def test_u1_pyramid_bit_exact_vs_reference_fixture(case):
    name, img, info, levels = case
    opts = dict(shrink=info["shrink"], n_per_oct=info["n_per_oct"], smooth=info["smooth"], channels=FUNCS[info["channels"]])
    got = list(wb.channels.channel_pyramid(img, opts))
    assert len(got) == info["n_levels"]
    for i, ((c, s), ref, rs) in enumerate(zip(got, levels, info["scales"])):
        assert c.dtype == np.uint8 and c.shape == ref.shape, (name, i)
        assert s == rs
        assert np.array_equal(c, ref), (name, i, np.abs(c.astype(int) - ref).max())


@pytest.mark.parametrize("fn", sorted(FUNCS))
@pytest.mark.parametrize("shape,shrink", [((8, 8), 2), ((9, 23), 2), ((131, 97), 2), ((480, 640), 2), ((150, 211), 1),
                                          ((200, 300), 4)])
def test_u1_pyramid_vs_oracle(fn, shape, shrink):
    img = synth_image(shape[0], shape[1], 31)
    # half the pixels pushed to the extremes: gradients large enough to hit the 255 clamp and the pool wrap
    rng = np.random.default_rng(3)
    img = np.where(rng.random(shape) < 0.3, rng.choice(np.array([0, 255], np.uint8), shape), img).astype(np.uint8)
    o = dict(shrink=shrink, n_per_oct=4 if shrink != 2 else 8, smooth=1)
    got = list(wb.channels.channel_pyramid(img, dict(o, channels=FUNCS[fn])))
    ref = list(orc.channel_pyramid(img, dict(o, channels=fn)))
    assert len(got) == len(ref)
    for (c, s), (rc, rs) in zip(got, ref):
        assert s == rs and c.shape == rc.shape and c.dtype == rc.dtype == np.uint8
        assert np.array_equal(c, rc)


@pytest.mark.parametrize("fn", sorted(FUNCS))
def test_u1_function_on_a_bare_image(fn):
    img = synth_image(77, 103, 9)
    got = FUNCS[fn](img)
    ref = orc.CHANNEL_FUNCS[fn](img)
    assert got.dtype == np.uint8 and got.shape == ref.shape and np.array_equal(got, ref)
    assert (got[0] == 0).all() and (got[:, -1] == 0).all()         # numba stencil border
    with pytest.raises(NotImplementedError):
        FUNCS[fn](img.astype(np.float32))
    with pytest.raises(ValueError):
        FUNCS[fn](np.zeros((4, 4, 1), np.uint8))


@pytest.mark.parametrize("fn", sorted(FUNCS))
def test_u1_detect_vs_reference_fixture(fn):
    meta = f2_meta()
    g = np.load(os.path.join(GOLDEN, f"{fn}_200x264.npz"))
    M = wb.load(os.path.join(GOLDEN, f"{fn}_d2_T24.pb"))
    assert M.channel_opts["channels"] is FUNCS[fn]
    assert wb.model.symbol_name(M.channel_opts["channels"]) == meta["names"][fn]
    res = M.detect_raw(g["image"])
    det = g["det"]
    assert det.size > 0
    assert M.n_loc == int(g["n_loc"]) and M.n_weak == int(g["n_weak"])
    assert np.array_equal(res["alive"], g["alive"])
    assert np.array_equal(res["level"], det["level"]) and np.array_equal(res["r"], det["r"]) and np.array_equal(res["c"], det["c"])
    assert np.array_equal(bits(res["scores"]), bits(det["score"]))
    assert np.array_equal(bits(res["boxes"]), bits(np.stack([det["x1"], det["y1"], det["x2"], det["y2"]], 1)))
    # level-by-level through the reference-shaped surface: uint8 channel arrays into predict_on_image
    M.reset()
    rows = []
    for li, (chns, scale, (r, c, h)) in enumerate(M.scan_channels(g["image"])):
        assert chns.dtype == np.uint8
        rows += [(li, int(a), int(b)) for a, b in zip(r, c)]
    assert rows == [(int(d["level"]), int(d["r"]), int(d["c"])) for d in det]
    assert M.n_weak == int(g["n_weak"])


def u1_model(fn, seed, T, depth, lo, hi, sa=-0.45, sb=-0.15, window=(12, 12)):
    rng = np.random.default_rng(seed)
    C = 4 if fn == "grad_hist_4_u1" else 1
    shape = (window[0], window[1], C)
    M = wb.Model(shape, dict(shrink=2, n_per_oct=8, smooth=1, channels=FUNCS[fn]))
    acc = 0.0
    for t in range(T):
        d = depth if depth else int(rng.integers(1, 4))
        f, th, l, r, p = random_tree_arrays(rng, shape, d, lo, hi)
        if t % 3 == 0:
            th = np.round(th)                 # integer thresholds: the `<=` ties of uint8 values are exercised
        acc += sb if t % 3 else sa
        M.append(wb.DTree(f, th, l, r, p), float("-inf") if t % 5 == 4 else float(np.float32(acc)))
    return M


@pytest.mark.parametrize("fn", sorted(FUNCS))
@pytest.mark.parametrize("depth", [1, 2, 0])
def test_u1_detect_vs_oracle(fn, depth):
    img = synth_image(300, 420, 17)
    M = u1_model(fn, 5 + depth, 40, depth, 1.0, 20.0)
    res = M.detect_raw(img)
    ref = oracle_detect(M, img)
    assert ref["scores"].size > 0
    assert np.array_equal(res["alive"], ref["alive"])
    assert np.array_equal(res["level"], ref["level"])
    assert np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"])
    assert np.array_equal(bits(res["scores"]), bits(ref["scores"]))
    assert np.array_equal(bits(res["boxes"]), bits(ref["boxes"]))
    assert M.n_loc == ref["n_loc"] and M.n_weak == ref["n_weak"]


@pytest.mark.parametrize("fn", sorted(FUNCS))
@pytest.mark.parametrize("window", [(9, 17), (10, 10), (16, 8), (20, 31)])
def test_u1_detect_other_window_shapes(fn, window):
    """Window widths with 64 + n not a multiple of 4 take the per-pixel uint8 tile load, the others the
    16-byte group load; both must agree with the oracle at the right and bottom borders of every level."""
    img = synth_image(203, 277, 23)
    M = u1_model(fn, 3, 24, 2, 1.0, 20.0, window=window)
    res = M.detect_raw(img)
    ref = oracle_detect(M, img)
    assert ref["scores"].size > 0
    assert np.array_equal(res["alive"], ref["alive"]) and np.array_equal(res["level"], ref["level"])
    assert np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"])
    assert np.array_equal(bits(res["scores"]), bits(ref["scores"]))


@pytest.mark.parametrize("fn", sorted(FUNCS))
def test_u1_cascade_threshold_corner_values(fn):
    """The uint8 cascade compares pixels with integerised thresholds: negative, -0.0, fractional, exactly integral,
    254.5 / 255 / above 255, +-inf and NaN thresholds must route windows like NumPy's `uint8 <= float32`."""
    rng = np.random.default_rng(12)
    C = 4 if fn == "grad_hist_4_u1" else 1
    shape = (12, 12, C)
    special = np.array([-5.5, -0.0, 0.0, 0.5, 1.0, 2.999, 3.0, 17.25, 254.5, 255.0, 255.5, 300.0, np.inf, -np.inf, np.nan], np.float32)
    M = wb.Model(shape, dict(shrink=2, n_per_oct=8, smooth=1, channels=FUNCS[fn]))
    acc = 0.0
    for t in range(36):
        f, th, l, r, p = random_tree_arrays(rng, shape, 2 if t % 4 else 3, 1.0, 20.0)
        pick = rng.random(th.size) < 0.5
        th = np.where(pick, rng.choice(special, th.size), th).astype(np.float32)
        acc += -0.2
        M.append(wb.DTree(f, th, l, r, p), float("-inf") if t % 6 == 5 else float(np.float32(acc)))
    img = synth_image(180, 260, 5)
    img[40:90, 60:140] = 255                      # saturated block: channel values 0 and large
    img[100:140, 20:100] = 0
    with np.errstate(invalid="ignore"):
        ref = oracle_detect(M, img)
    res = M.detect_raw(img)
    assert ref["scores"].size > 0
    assert np.array_equal(res["alive"], ref["alive"]) and np.array_equal(res["level"], ref["level"])
    assert np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"])
    assert np.array_equal(bits(res["scores"]), bits(ref["scores"]))
    # the same trees on explicit window lists and on samples (tree_eval / tree_apply kernels: float compares)
    chns, _ = next(iter(wb.channels.channel_pyramid(img, M.channel_opts)))
    rs, cs = np.indices((chns.shape[0] - 12, chns.shape[1] - 12))
    rs, cs = rs.flatten()[::11], cs.flatten()[::11]
    for w in M.classifier[:6]:
        tree = orc.make_tree(w.feature, w.threshold, w.left, w.right, w.prediction)
        with np.errstate(invalid="ignore"):
            assert np.array_equal(bits(w.predict_on_image(chns, rs, cs)), bits(orc.tree_predict_on_image(tree, chns, rs, cs)))


def test_u1_batch_and_tree_eval():
    fn = "grad_hist_4_u1"
    M = u1_model(fn, 9, 24, 2, 1.0, 20.0)
    imgs = np.stack([synth_image(240, 320, s) for s in (1, 2, 3)])
    res = M.detect_batch_raw(imgs)
    for b in range(3):
        ref = oracle_detect(M, imgs[b])
        sel = res["image"] == b
        assert np.array_equal(res["r"][sel], ref["r"]) and np.array_equal(res["c"][sel], ref["c"])
        assert np.array_equal(bits(res["scores"][sel]), bits(ref["scores"]))
        assert np.array_equal(res["alive"][b], ref["alive"])
    # DTree.predict_on_image on a uint8 channel image
    chns, _ = next(iter(wb.channels.channel_pyramid(imgs[0], M.channel_opts)))
    rs, cs = np.indices((chns.shape[0] - 12, chns.shape[1] - 12))
    rs, cs = rs.flatten()[::7], cs.flatten()[::7]
    w = M.classifier[3]
    got = w.predict_on_image(chns, rs, cs)
    ref = orc.tree_predict_on_image(orc.make_tree(w.feature, w.threshold, w.left, w.right, w.prediction), chns, rs, cs)
    assert np.array_equal(bits(got), bits(ref))


def test_u1_1080p_vs_oracle():
    """Full-size frame, 64-stage depth-2 cascade over uint8 channels."""
    img = synth_image(1080, 1920, 0)
    M = u1_model("grad_hist_4_u1", 21, 64, 2, 1.0, 20.0, 0.05, 0.02)
    res = M.detect_raw(img)
    ref = oracle_detect(M, img)
    assert ref["n_loc"] == 3045278
    assert np.array_equal(res["alive"], ref["alive"])
    assert np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"]) and np.array_equal(res["level"], ref["level"])
    assert np.array_equal(bits(res["scores"]), bits(ref["scores"]))


# ------------------------------------------------------------------------------ grad_mag (float32, 1 channel)
@pytest.mark.parametrize("case", GM_CASES, ids=lambda c: c[0])
def test_grad_mag_pyramid_bit_exact_vs_reference_fixture(case):
    name, img, info, levels = case
    opts = dict(shrink=info["shrink"], n_per_oct=info["n_per_oct"], smooth=info["smooth"], channels=wb.channels.grad_mag)
    got = list(wb.channels.channel_pyramid(img, opts))
    assert len(got) == info["n_levels"]
    for i, ((c, s), ref, rs) in enumerate(zip(got, levels, info["scales"])):
        assert c.dtype == np.float32 and c.shape == ref.shape, (name, i)
        assert s == rs
        assert np.array_equal(bits(c), bits(ref)), (name, i, np.abs(c - ref).max())


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("shape,shrink", [((8, 8), 2), ((9, 23), 2), ((131, 97), 2), ((480, 640), 2), ((150, 211), 1),
                                          ((200, 300), 4)])
def test_grad_mag_pyramid_vs_oracle(shape, shrink, dtype):
    img = synth_image(shape[0], shape[1], 41, dtype)
    o = dict(shrink=shrink, n_per_oct=4 if shrink != 2 else 8, smooth=1)
    got = list(wb.channels.channel_pyramid(img, dict(o, channels=wb.channels.grad_mag)))
    ref = list(orc.channel_pyramid(img, dict(o, channels="grad_mag")))
    assert len(got) == len(ref)
    for (c, s), (rc, rs) in zip(got, ref):
        assert s == rs and c.shape == rc.shape and c.dtype == np.float32
        assert np.array_equal(bits(c), bits(rc))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_grad_mag_on_a_bare_image(dtype):
    for shape in [(77, 103), (5, 40), (3, 3)]:          # the small ones reflect more than once inside the 11-tap filter
        img = synth_image(shape[0], shape[1], 9, dtype)
        got = wb.channels.grad_mag(img)
        ref = orc.grad_mag(img)
        assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref))
    for kw in (dict(norm=3), dict(norm=None), dict(norm=7, eps=0.25)):      # other arguments: the plain kernels
        assert np.array_equal(bits(wb.channels.grad_mag(img, **kw)), bits(orc.grad_mag(img, **kw)))


def test_grad_mag_detect_vs_reference_fixture():
    fn = "grad_mag"
    meta = f2_meta()
    g = np.load(os.path.join(GOLDEN, f"{fn}_200x264.npz"))
    M = wb.load(os.path.join(GOLDEN, f"{fn}_d2_T24.pb"))
    assert M.channel_opts["channels"] is wb.channels.grad_mag
    assert wb.model.symbol_name(M.channel_opts["channels"]) == meta["names"][fn] == "waldboost.channels.grad_mag"
    res = M.detect_raw(g["image"])
    det = g["det"]
    assert det.size > 0
    assert M.n_loc == int(g["n_loc"]) and M.n_weak == int(g["n_weak"])
    assert np.array_equal(res["alive"], g["alive"])
    assert np.array_equal(res["level"], det["level"]) and np.array_equal(res["r"], det["r"]) and np.array_equal(res["c"], det["c"])
    assert np.array_equal(bits(res["scores"]), bits(det["score"]))
    assert np.array_equal(bits(res["boxes"]), bits(np.stack([det["x1"], det["y1"], det["x2"], det["y2"]], 1)))


@pytest.mark.parametrize("case", list(__import__("util").chanfunc_arg_cases()), ids=lambda c: c[0])
def test_channel_functions_with_arguments_vs_reference_fixture(case):
    """grad_hist(image, n_bins, full, bias) / grad_mag(image, norm, eps) called directly (reference channels.py:30-52)."""
    import waldboost_amd as wb
    name, img, func, kwargs, ref = case
    got = getattr(wb.channels, func)(img, **kwargs)
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_channel_function_arguments_that_change_the_arithmetic_dtype():
    """A float64 / int64 NumPy scalar is a strong type under NumPy-2 promotion: `np.abs(chns) - bias` (reference
    channels.py:51) then yields a float64 array, `mag /= norm + eps` (:36) divides in float64 and rounds once back to
    float32.  The oracle evaluates the reference's own expressions."""
    import waldboost_amd as wb
    from waldboost_amd.synth import synth_image
    for img in (synth_image(61, 83, 31), synth_image(50, 70, 32, np.float32)):
        for kw in (dict(bias=np.float64(1.25)), dict(n_bins=6, full=True, bias=np.float64(0.3)), dict(n_bins=3, bias=np.int64(2)),
                   dict(bias=np.float64(0.0))):
            got, ref = wb.channels.grad_hist(img, **kw), orc.grad_hist(img, **kw)
            assert got.dtype == ref.dtype == np.float64 and got.shape == ref.shape
            assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), kw
        for kw in (dict(eps=np.float64(1e-3)), dict(norm=3, eps=np.float64(0.01))):
            got, ref = wb.channels.grad_mag(img, **kw), orc.grad_mag(img, **kw)
            assert got.dtype == ref.dtype == np.float32 and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), kw
        narrow = wb.channels.grad_mag(img, norm=3, eps=0.01)
        assert not np.array_equal(narrow, wb.channels.grad_mag(img, norm=3, eps=np.float64(0.01))) or img.dtype == np.uint8
    img = np.zeros((20, 30), np.uint8)
    assert wb.channels.grad_hist(img, n_bins=3, bias=np.float32(1.0)).dtype == np.float32
    with pytest.raises(NotImplementedError):
        wb.channels.grad_hist(img, bias=np.complex64(1.0))


# ------------------------------------------------------------------------------ a channel function without a kernel
def _two_channels(im):
    """A caller's own channel function (reference channels.py:119,136 calls whatever channel_opts["channels"] holds)."""
    f = im.astype("f")
    return np.stack([f * np.float32(0.5), np.sqrt(np.abs(f) + np.float32(1.0))], -1)


def _inverted_u8(im):
    return (255 - im.astype(np.uint8))[..., None]


@pytest.mark.parametrize("func,dtype,shrink,smooth", [(_two_channels, np.uint8, 2, 1), (_two_channels, np.float32, 1, 1),
                                                      (_two_channels, np.int16, 2, 0), (_inverted_u8, np.uint8, 2, 1),
                                                      (_inverted_u8, np.uint8, 1, 0)])
def test_pyramid_around_a_callable_without_a_kernel_vs_oracle(func, dtype, shrink, smooth):
    """The GPU computes the octaves, every level's resize and cast back, avg_pool_2 and smooth_image_3d; the caller's
    function runs on the host array it is handed -- every level bit-exact against the oracle running the same function."""
    import waldboost_amd as wb
    img = synth_image(150, 201, 9)
    img = (img.astype(np.int32) * 100 - 9000).astype(dtype) if dtype == np.int16 else img.astype(dtype)
    opts = dict(shrink=shrink, n_per_oct=4, smooth=smooth, channels=func)
    got = list(wb.channels.channel_pyramid(img, opts))
    ref = list(orc.channel_pyramid(img, opts))
    assert len(got) == len(ref) > 4
    for (c, s), (rc, rs) in zip(got, ref):
        assert s == rs and c.dtype == rc.dtype and c.shape == rc.shape
        assert np.array_equal(c.view(np.uint8), np.ascontiguousarray(rc).view(np.uint8))
    with pytest.raises(NotImplementedError):
        list(wb.channels.channel_pyramid(img, dict(opts, channels=lambda im: im.astype(np.float64)[..., None])))


def test_model_detect_with_a_callable_without_a_kernel_vs_oracle():
    import waldboost_amd as wb
    from waldboost_amd.synth import random_tree_arrays
    from util import oracle_detect
    rng = np.random.default_rng(4)
    shape = (10, 12, 2)
    M = wb.Model(shape, dict(shrink=2, n_per_oct=4, smooth=1, channels=_two_channels))
    acc = 0.0
    for t in range(12):
        f, th, l, r, p = random_tree_arrays(rng, shape, 2, 4.0, 60.0)
        acc -= 0.3
        M.append(wb.DTree(f, th, l, r, p), float(np.float32(acc)))
    img = synth_image(160, 220, 3)
    trees = [orc.make_tree(w.feature, w.threshold, w.left, w.right, w.prediction) for w in M.classifier]
    ref = orc.detect(shape, dict(M.channel_opts), trees, list(M.theta), img)
    res = M.detect_raw(img)
    assert ref["scores"].size > 0 and np.array_equal(res["alive"], ref["alive"])
    assert np.array_equal(res["level"], ref["level"]) and np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"])
    assert np.array_equal(res["scores"].view(np.uint32), ref["scores"].view(np.uint32)) and np.array_equal(res["boxes"], ref["boxes"])
    assert len(M.detect(img)) == ref["scores"].size and M.n_loc == 2 * ref["n_loc"]
