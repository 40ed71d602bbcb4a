"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the
C ABI via the reference-shaped Python surface, against the golden fixtures generated from the
reference and against the CPU oracle on the same seeded inputs.

Bar: channel arrays, window indices and per-stage alive counts bit-exact; scores bit-exact
(the north-star tolerance is 1e-5, the kernels accumulate in the reference's order so the test
asks for equality); boxes bit-exact.
"""
import os

import numpy as np
import pytest

import waldboost_amd as wb
from oracle import wb_oracle as orc
from waldboost_amd.synth import synth_image, random_tree_arrays
from util import GOLDEN, dtype_cases, golden_meta, oracle_detect, oracle_model, small_cases

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_same_detections(res, ref):
    assert np.array_equal(res["alive"], ref["alive"])
    assert np.array_equal(res["level"], ref["level"])
    assert np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"])
    assert np.array_equal(bits(res["scores"]), bits(ref["scores"]))
    assert np.array_equal(bits(res["boxes"]), bits(ref["boxes"]))


def random_model(seed, T, depth, shape=(12, 12, 4), opts=None, survive=0.85, mixed=False):
    rng = np.random.default_rng(seed)
    opts = opts or dict(wb.default_channel_opts)
    M = wb.Model(shape, opts)
    acc = 0.0
    for t in range(T):
        d = depth
        unb = False
        if mixed:
            d = int(rng.integers(1, 4))
            unb = d == 2 and rng.random() < 0.5
        f, th, l, r, p = random_tree_arrays(rng, shape, d, 2.0, 60.0, unbalanced=unb)
        acc += -0.15 if t % 3 else -0.45
        theta = float("-inf") if t % 5 == 4 else float(np.float32(acc))
        M.append(wb.DTree(f, th, l, r, p), theta)
    return M


# ------------------------------------------------------------------------------ channels
@pytest.mark.parametrize("case", list(small_cases()), ids=lambda c: c[0])
def test_channel_pyramid_bit_exact_vs_reference_fixture(case):
    name, img, info, levels = case
    opts = dict(shrink=info["shrink"], n_per_oct=info["n_per_oct"], smooth=info["smooth"], channels=wb.channels.grad_hist)
    got = list(wb.channels.channel_pyramid(img, opts))
    assert len(got) == info["n_levels"]
    for i, ((c, s), ref, rs) in enumerate(zip(got, levels, info["scales"])):
        assert c.dtype == np.float32 and c.shape == ref.shape, (name, i)
        assert s == rs
        assert np.array_equal(bits(c), bits(ref)), (name, i, np.abs(c - ref).max())


@pytest.mark.parametrize("case", list(dtype_cases()), ids=lambda c: c[0])
def test_channel_pyramid_of_other_image_dtypes_vs_reference_fixture(case):
    """float64 and integer images keep their dtype through the octaves and the resize (reference channels.py:122,
    :132): integer octave sums wrap, the resize result is truncated back; every level bit-exact, and a detection
    on such an image equals the oracle's."""
    name, img, info, levels = case
    opts = dict(wb.default_channel_opts, shrink=info["shrink"], n_per_oct=info["n_per_oct"], smooth=info["smooth"])
    got = list(wb.channels.channel_pyramid(img, opts))
    assert len(got) == info["n_levels"]
    for (c, s), ref, rs in zip(got, levels, info["scales"]):
        assert s == rs and c.dtype == np.float32 and c.shape == ref.shape
        assert np.array_equal(bits(c), bits(ref))
    M = random_model(7, 12, 2, opts=opts)
    # thresholds where the channel values are (these images span very different ranges)
    vals = np.concatenate([c.reshape(-1) for c, _ in got])
    rng = np.random.default_rng(5)
    for w in M.classifier:
        w.threshold[:] = np.quantile(vals, rng.uniform(0.3, 0.7, w.threshold.size)).astype(np.float32)
    with np.errstate(over="ignore"):
        ref = oracle_detect(M, img)
    assert_same_detections(M.detect_raw(img), ref)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("shape", [(8, 8), (9, 23), (131, 97), (480, 640)])
def test_channel_pyramid_vs_oracle(shape, dtype):
    img = synth_image(shape[0], shape[1], 21, dtype)
    if dtype == np.uint8:
        img = np.clip(img.astype(np.int32) + 60, 0, 255).astype(np.uint8)   # bright: triggers the S2 wrap
    o = dict(shrink=2, n_per_oct=8, smooth=1)
    got = list(wb.channels.channel_pyramid(img, dict(o, channels=wb.channels.grad_hist)))
    ref = list(orc.channel_pyramid(img, dict(o, channels=orc.grad_hist)))
    assert len(got) == len(ref)
    for (c, s), (rc, rs) in zip(got, ref):
        assert s == rs and c.shape == rc.shape
        assert np.array_equal(bits(c), bits(rc))


def test_channel_pyramid_is_lazy_and_survives_interleaving():
    """reference channels.py:125-146: a level is computed when the generator is advanced to it.  Two generators
    over different images of one shape share the cached engine and still yield their own levels; a consumer that
    stops early leaves the later levels uncomputed."""
    from waldboost_amd import engine as _engine
    opts = dict(wb.default_channel_opts)
    oo = dict(opts)
    oo["channels"] = orc.CHANNEL_FUNCS["grad_hist"]
    a, b = synth_image(150, 200, 71), synth_image(150, 200, 72)
    ra, rb = list(orc.channel_pyramid(a, oo)), list(orc.channel_pyramid(b, oo))
    ga, gb = wb.channels.channel_pyramid(a, opts), wb.channels.channel_pyramid(b, opts)
    for l in range(len(ra)):
        (ca, sa), (cb, sb) = next(ga), next(gb)
        assert sa == ra[l][1] and np.array_equal(bits(ca), bits(ra[l][0]))
        assert sb == rb[l][1] and np.array_equal(bits(cb), bits(rb[l][0]))
    eng = _engine.get_engine(150, 200, np.uint8, 2, 8, 1, 1)
    eng.chn.fill_(float("nan"))
    g = wb.channels.channel_pyramid(a, opts)
    next(g), next(g)
    g.close()
    assert not np.isnan(eng.read_level(0, 1)).any()
    assert np.isnan(eng.read_level(0, 2)).all() and np.isnan(eng.read_level(0, len(ra) - 1)).all()


def test_clean_edges_hit_the_projection_leftovers():
    """Noise-free blocks, ramps and 45-degree edges: gy == 0 / gx == +-gy over whole 2x2 blocks, so the
    1e-13..1e-16-sized leftovers of the fp64 projection survive the shrink and reach the smooth
    (the kernel's exact-order path for tiles holding such values)."""
    H, W = 160, 224
    y, x = np.mgrid[0:H, 0:W]
    img = np.full((H, W), 40, np.uint8)
    img[20:70, 30:90] = 200                                   # axis-aligned square
    img[(x - y > 60) & (x - y < 110) & (y > 80)] = 120        # 45-degree band
    img[(x + y > 250) & (x + y < 290) & (y < 75)] = 90        # -45-degree band
    img[100:150, 10:60] = (np.arange(50) * 3)[None, :]        # horizontal ramp (gy == 0 everywhere)
    for o in (dict(shrink=2, n_per_oct=4, smooth=1), dict(shrink=1, n_per_oct=2, smooth=1), dict(shrink=2, n_per_oct=2, smooth=0)):
        got = list(wb.channels.channel_pyramid(img, dict(o, channels=wb.channels.grad_hist)))
        ref = list(orc.channel_pyramid(img, dict(o, channels=orc.grad_hist)))
        assert len(got) == len(ref) > 0
        tiny = 0
        for (c, s), (rc, rs) in zip(got, ref):
            assert s == rs and np.array_equal(bits(c), bits(rc))
            tiny += int(((rc > 0) & (rc < 1e-6)).sum())
        assert tiny > 0, "the image no longer produces leftover values"


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_low_entropy_images_mix_residues_and_ordinary_values(seed):
    """Two-level images made of small random blocks, stripes and diagonals: gy == 0 / gx == +-gy hold for
    a large share of the pixels, in every mixture with ordinary gradients inside the 2x2 (4x4) shrink
    blocks -- the cases the channel kernel's two-pass projection must tell apart."""
    rng = np.random.default_rng(seed)
    H, W = 96 + 8 * seed, 136
    y, x = np.mgrid[0:H, 0:W]
    bs = int(rng.integers(2, 6))
    blocks = np.kron(rng.integers(0, 2, (H // bs + 1, W // bs + 1)), np.ones((bs, bs), np.int64))[:H, :W]
    diag = ((x + (1 if seed % 2 else -1) * y) // int(rng.integers(3, 9))) % 2
    stripes = (x // int(rng.integers(2, 7))) % 2
    pick = rng.integers(0, 3, (H // 16 + 1, W // 16 + 1))
    pick = np.kron(pick, np.ones((16, 16), np.int64))[:H, :W]
    img = (np.choose(pick, [blocks, diag, stripes]) * int(rng.integers(40, 255))).astype(np.uint8)
    total_tiny = 0
    for o in (dict(shrink=2, n_per_oct=4, smooth=1), dict(shrink=4, n_per_oct=2, smooth=0), dict(shrink=1, n_per_oct=2, smooth=1)):
        got = list(wb.channels.channel_pyramid(img, dict(o, channels=wb.channels.grad_hist)))
        ref = list(orc.channel_pyramid(img, dict(o, channels=orc.grad_hist)))
        assert len(got) == len(ref) > 0
        for (c, s), (rc, rs) in zip(got, ref):
            assert s == rs and np.array_equal(bits(c), bits(rc))
            total_tiny += int(((rc > 0) & (rc < 1e-6)).sum())
    assert total_tiny > 0


def test_shrink4_extension_vs_oracle():
    img = synth_image(200, 300, 5)
    o = dict(shrink=4, n_per_oct=3, smooth=1)
    got = list(wb.channels.channel_pyramid(img, dict(o, channels=wb.channels.grad_hist)))
    ref = list(orc.channel_pyramid(img, dict(o, channels=orc.grad_hist)))
    assert len(got) == len(ref) > 0
    for (c, s), (rc, rs) in zip(got, ref):
        assert s == rs and np.array_equal(bits(c), bits(rc))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_grad_hist_vs_oracle(dtype):
    for shape in [(77, 103), (5, 40), (3, 3), (1, 9)]:       # also images smaller than the pyramid's 8-pixel floor
        img = synth_image(shape[0], shape[1], 9, dtype)
        got = wb.channels.grad_hist(img)
        ref = orc.grad_hist(img)
        assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref))


def test_image_validation_errors():
    with pytest.raises(TypeError):
        list(wb.channels.channel_pyramid([[1, 2]], wb.default_channel_opts))
    with pytest.raises(ValueError):
        list(wb.channels.channel_pyramid(np.zeros((4, 4, 1), np.uint8), wb.default_channel_opts))
    with pytest.raises(AssertionError):
        list(wb.channels.channel_pyramid(np.zeros((16, 16), np.uint8), dict(wb.default_channel_opts, shrink=3)))
    assert list(wb.channels.channel_pyramid(np.zeros((7, 40), np.uint8), wb.default_channel_opts)) == []


# ------------------------------------------------------------------------------ cascade
def test_cfg1_detect_vs_reference_fixture():
    meta = golden_meta()["cfg1"]
    g = np.load(os.path.join(GOLDEN, "cfg1_640x480.npz"))
    M = wb.load(os.path.join(GOLDEN, "cfg1_d1_T32.pb"))
    img = synth_image(480, 640, 0)
    res = M.detect_raw(img)
    det = g["det"]
    assert M.n_loc == meta["n_loc"] and M.n_weak == meta["n_weak"]
    assert np.array_equal(res["alive"], g["alive"])
    assert np.array_equal(res["level"], det["level"]) and np.array_equal(res["r"], det["r"]) and np.array_equal(res["c"], det["c"])
    assert np.array_equal(bits(res["scores"]), bits(det["score"]))
    assert np.array_equal(bits(res["boxes"]), bits(np.stack([det["x1"], det["y1"], det["x2"], det["y2"]], 1)))
    # the public surface: Boxes + 'scores', stats accumulate across calls
    bx = M.detect(img)
    assert len(bx) == det.size and np.array_equal(bx.get_field("scores"), det["score"])
    assert M.n_loc == 2 * meta["n_loc"] and abs(M.eval_cost - meta["eval_cost"]) < 1e-12
    M.reset()
    assert M.n_loc == 0 and M.eval_cost == 0


def test_mixed_depth_model_vs_reference_fixture():
    g = np.load(os.path.join(GOLDEN, "mixed_200x264.npz"))
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    res = M.detect_raw(g["image"])
    det = g["det"]
    assert M.n_loc == int(g["n_loc"]) and M.n_weak == int(g["n_weak"])
    assert np.array_equal(res["alive"], g["alive"])
    assert np.array_equal(res["level"], det["level"]) and np.array_equal(res["r"], det["r"]) and np.array_equal(res["c"], det["c"])
    assert np.array_equal(bits(res["scores"]), bits(det["score"]))


def test_all_rejecting_stage_and_empty_model():
    g = np.load(os.path.join(GOLDEN, "reject_200x264.npz"))
    img = np.load(os.path.join(GOLDEN, "mixed_200x264.npz"))["image"]
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    R = wb.Model(M.shape, M.channel_opts)
    for w, t in zip(M.classifier, g["theta"]):
        R.append(w, float(t))
    res = R.detect_raw(img)
    assert res["scores"].size == 0 and np.array_equal(res["alive"], g["alive"])
    assert R.n_weak == int(g["n_weak"]) and R.n_loc == int(g["n_loc"])

    e = np.load(os.path.join(GOLDEN, "empty_40x56.npz"))
    E = wb.Model((12, 12, 4), dict(wb.default_channel_opts))
    bx = E.detect(e["image"])
    assert np.array_equal(bx.get(), e["boxes"]) and np.array_equal(bx.get_field("scores"), e["scores"])
    assert E.n_loc == int(e["n_loc"]) and E.n_weak == 0 and not E and len(E) == 0


@pytest.mark.parametrize("depth,T", [(1, 7), (2, 40), (3, 21)])
def test_detect_vs_oracle_random_models(depth, T):
    img = synth_image(300, 420, 33 + depth)
    M = random_model(100 + depth, T, depth)
    res = M.detect_raw(img)
    ref = oracle_detect(M, img)
    assert ref["scores"].size > 0
    assert_same_detections(res, ref)
    assert M.n_loc == ref["n_loc"] and M.n_weak == ref["n_weak"]


def test_detect_mixed_depths_float_image_and_odd_window():
    img = synth_image(211, 333, 44, np.float32)
    opts = dict(shrink=2, n_per_oct=5, smooth=1, channels=wb.channels.grad_hist)
    M = random_model(7, 33, 2, shape=(9, 17, 4), opts=opts, mixed=True)
    # float32 images have small channel values: rescale thresholds into range
    for w in M.classifier:
        w.threshold *= np.float32(1.0 / 255.0)
    M._device = None
    res = M.detect_raw(img)
    ref = oracle_detect(M, img)
    assert_same_detections(res, ref)


def test_special_threshold_and_prediction_values():
    """NaN / +-inf / -0.0 thresholds (a NaN comparison is False: the window goes right) and stages whose
    rejection threshold is -inf, +inf or NaN, on float32 channels."""
    rng = np.random.default_rng(31)
    shape = (12, 12, 4)
    special = np.array([np.nan, np.inf, -np.inf, -0.0, 0.0, 1e-30, 3.0e38], np.float32)
    M = wb.Model(shape, dict(wb.default_channel_opts))
    acc = 0.0
    for t in range(30):
        f, th, l, r, p = random_tree_arrays(rng, shape, 2, 2.0, 60.0)
        th = np.where(rng.random(th.size) < 0.4, rng.choice(special, th.size), th).astype(np.float32)
        acc += -0.25
        theta = [float(np.float32(acc)), float("-inf"), float(np.float32(acc))][t % 3]
        M.append(wb.DTree(f, th, l, r, p), theta)
    img = synth_image(190, 250, 77)
    with np.errstate(invalid="ignore"):
        ref = oracle_detect(M, img)
    res = M.detect_raw(img)
    assert ref["scores"].size > 0
    assert_same_detections(res, ref)
    # a NaN stage threshold rejects everything (x >= NaN is False), +inf likewise
    for bad in (float("nan"), float("inf")):
        M2 = wb.Model(shape, dict(wb.default_channel_opts))
        for i, (w, th) in enumerate(M):
            M2.append(w, bad if i == 4 else th)
        with np.errstate(invalid="ignore"):
            ref2 = oracle_detect(M2, img)
        res2 = M2.detect_raw(img)
        assert ref2["scores"].size == 0 and res2["scores"].size == 0
        assert np.array_equal(res2["alive"], ref2["alive"])


def test_predict_on_image_and_tree_eval_vs_oracle():
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 60, (70, 150, 4)).astype(np.float32)
    M = random_model(11, 19, 2)
    shape, _, trees, thetas = oracle_model(M)
    rs, cs, hs, alive = M.predict_on_image_stats(X)
    ors, ocs, ohs, oalive = orc.cascade_predict_on_image(shape, trees, thetas, X)
    assert np.array_equal(rs, ors) and np.array_equal(cs, ocs) and np.array_equal(bits(hs), bits(ohs))
    assert np.array_equal(alive, oalive)
    assert M.n_loc == (70 - 12) * (150 - 12) and M.n_weak == int(oalive.sum())
    # window smaller than the image in neither direction -> no windows
    r2, c2, h2 = M.predict_on_image(X[:12, :40])
    assert r2.size == 0 and h2.dtype == np.float32
    with pytest.raises(AssertionError):
        M.predict_on_image(X[:, :, :3])
    # single tree on explicit positions (DTree.predict_on_image)
    prs = rng.integers(0, 58, 1000)
    pcs = rng.integers(0, 138, 1000)
    for w, t in zip(M.classifier[:4], trees[:4]):
        got = w.predict_on_image(X, prs, pcs)
        assert np.array_equal(bits(got), bits(orc.tree_predict_on_image(t, X, prs, pcs)))


@pytest.mark.parametrize("dtype", [np.float64, np.float16, np.int16, np.int32, np.int64, np.uint16, np.uint32, np.bool_])
def test_channel_arrays_of_any_dtype_compare_as_numpy_does(dtype):
    """reference training.py:92 / model.py:199 compare X with the float32 thresholds in whatever arithmetic NumPy
    promotes to (float64 for float64 / int32 / int64 arrays): the values handed to the kernels must decide every
    node the same way (waldboost_amd/compare.py).  Thresholds are placed on and next to the array's values."""
    rng = np.random.default_rng(17)
    if dtype == np.float64:
        X = rng.uniform(0, 60, (40, 90, 4))
        X[::3] = np.float32(X[::3]) + 1e-9                       # just above a float32 value
    elif dtype == np.float16:
        X = rng.uniform(0, 60, (40, 90, 4)).astype(dtype)
    elif dtype == np.bool_:
        X = rng.integers(0, 2, (40, 90, 4)).astype(dtype)
    else:
        hi = min(np.iinfo(dtype).max, 2 ** 26)
        X = rng.integers(hi - 70, hi, (40, 90, 4)).astype(dtype)     # int32/int64: beyond float32's integers
    vals = np.unique(X.astype(np.float64))
    M = wb.Model((12, 12, 4), dict(wb.default_channel_opts))
    acc = 0.0
    for t in range(12):
        f, th, l, r, p = random_tree_arrays(rng, (12, 12, 4), 2, 0.0, 1.0)
        pick = np.float32(rng.choice(vals, th.size))
        th = np.where(rng.random(th.size) < 0.5, pick, np.nextafter(pick, np.float32(np.inf))).astype(np.float32)
        acc -= 0.3
        M.append(wb.DTree(f, th, l, r, p), float(np.float32(acc)))
    shape, _, trees, thetas = oracle_model(M)
    rs, cs, hs, alive = M.predict_on_image_stats(X)
    ors, ocs, ohs, oalive = orc.cascade_predict_on_image(shape, trees, thetas, X)
    assert np.array_equal(alive, oalive) and ohs.size > 0
    assert np.array_equal(rs, ors) and np.array_equal(cs, ocs) and np.array_equal(bits(hs), bits(ohs))
    prs, pcs = rng.integers(0, 28, 500), rng.integers(0, 78, 500)
    for w, t in zip(M.classifier[:3], trees[:3]):
        assert np.array_equal(bits(w.predict_on_image(X, prs, pcs)), bits(orc.tree_predict_on_image(t, X, prs, pcs)))
    crops = np.stack([X[r:r + 12, c:c + 12] for r, c in zip(prs[:50], pcs[:50])])
    H, mask = M.predict(crops)
    oH, omask = orc.model_predict(shape, trees, thetas, crops)
    assert np.array_equal(mask, omask) and np.array_equal(bits(H), bits(oH))
    with pytest.raises(TypeError):
        M.predict_on_image(X.astype(np.complex64))


def test_theta_scalar_kinds_follow_numpy_promotion():
    rng = np.random.default_rng(8)
    X = rng.uniform(0, 60, (40, 90, 4)).astype(np.float32)
    f, th, l, r, p = random_tree_arrays(rng, (12, 12, 4), 2, 5.0, 50.0)
    for theta in (0.1, np.float64(0.1), np.float32(0.1), np.float64(p[3]) + 1e-12, float(p[4])):
        M = wb.Model((12, 12, 4), dict(wb.default_channel_opts))
        M.append(wb.DTree(f, th, l, r, p), theta)
        shape, _, trees, thetas = oracle_model(M)
        rs, cs, hs = M.predict_on_image(X)
        ors, ocs, ohs, _ = orc.cascade_predict_on_image(shape, trees, thetas, X)
        assert np.array_equal(rs, ors) and np.array_equal(cs, ocs) and np.array_equal(bits(hs), bits(ohs))


def test_1080p_depth2_128_stages_vs_oracle():
    """BASELINE config 2 at full size against the oracle (about 10 s of CPU)."""
    M = wb.load(os.path.join(GOLDEN, "models", "cfg2_d2_T128.pb"))
    img = synth_image(1080, 1920, 0)
    res = M.detect_raw(img)
    ref = oracle_detect(M, img)
    assert ref["n_loc"] == 3045278 == M.n_loc
    assert_same_detections(res, ref)
    assert M.n_weak == ref["n_weak"]


# ------------------------------------------------------------------------------ octaves
@pytest.mark.parametrize("shape,dtype", [((1080, 1920), np.float32), ((2160, 3840), np.uint8), ((1081, 1923), np.uint8),
                                         ((8, 8), np.uint8), ((517, 263), np.float32)])
def test_octaves_and_clip_range_vs_oracle(shape, dtype):
    """The fused octave kernel (block kernel + tail kernel for deep pyramids) against the
    oracle's avg_pool_2 chain, including the uint8 wrap and the per-octave min/max."""
    from waldboost_amd.engine import PyramidEngine
    img = synth_image(shape[0], shape[1], 17, dtype)
    if dtype == np.uint8:
        img = np.clip(img.astype(np.int32) + 70, 0, 255).astype(np.uint8)
    e = PyramidEngine(shape[0], shape[1], dtype, 2, 8, 1, batch=1)
    e.load_images(img)
    e.launch_octaves()
    octs = list(orc.image_octaves(img))
    assert len(octs) == e.plan.n_oct
    buf = e.oct[0].cpu().numpy()
    mm = e.minmax[0].cpu().numpy().view(np.uint32)
    for k, o in enumerate(octs):
        if k:
            got = buf[int(e.plan.oct_off[k]):int(e.plan.oct_off[k]) + o.size].reshape(o.shape)
            assert np.array_equal(got.view(np.uint8), o.view(np.uint8)), k
        lo_key, hi_key = np.uint32(~mm[k, 0]), mm[k, 1]
        if dtype == np.uint8:
            assert (int(lo_key), int(hi_key)) == (int(o.min()), int(o.max())), k
        else:
            from waldboost_amd.engine import nat_f32_key
            assert lo_key == nat_f32_key(o.min()) and hi_key == nat_f32_key(o.max()), k


# ------------------------------------------------------------------------------ long cascades
@pytest.mark.parametrize("depth,T,floor", [(2, 70, -1.0), (1, 100, -1.0), (3, 66, -1.0), (2, 50, -30.0), (2, 130, -0.2),
                                           (2, 200, -1.0)])
def test_long_cascades_wave_synchronous_and_stage_parallel_tails(depth, T, floor):
    """Past stage 16 a wave with <= 8 windows left switches to the stage-parallel tail (one stage
    per lane, serial fp32 replay); crowded waves stay wave-synchronous.  A slowly falling
    rejection floor keeps a mix of both alive to the last stage (floor=-30 keeps nearly every
    window: all wave-synchronous, and the detection buffer has to grow)."""
    img = synth_image(260, 380, 50 + T)
    M = random_model(300 + T, T, depth)
    M.theta = [float(np.float32(floor - 0.3 * i)) if i % 3 == 0 else float("-inf") for i in range(T)]
    M._device = None
    res = M.detect_raw(img)
    ref = oracle_detect(M, img)
    assert ref["alive"][:, -1].sum() > 0
    assert_same_detections(res, ref)
    assert M.n_weak == ref["n_weak"]


def test_uint8_projection_fast_path_is_exact_for_every_gradient_pair():
    """project_int (fp32) == project_f64 (the reference's fp64 formula) for all 2041^2 integer
    gradient pairs a uint8 image can produce -- checked exhaustively on the device."""
    import torch
    from waldboost_amd import _native as nat
    lib = nat.load()
    dev = nat.require_gpu()
    mism = torch.zeros(1, dtype=torch.int32, device=dev)
    nat.check(lib.wb_selftest_projection(nat.stream_ptr(), nat.ptr(mism)), "wb_selftest_projection")
    assert int(mism.item()) == 0


def test_uint8_fast_and_generic_projection_agree(monkeypatch):
    img = synth_image(333, 517, 91)
    img[40:90, 100:300] = 255          # saturated flat block and hard edges: gy == 0 / gx == gy lanes
    img[200:260, 50:120] = 0
    o = dict(shrink=2, n_per_oct=4, smooth=1, channels=wb.channels.grad_hist)
    fast = [c.copy() for c, _ in wb.channels.channel_pyramid(img, o)]
    ref = [c for c, _ in orc.channel_pyramid(img, dict(o, channels=orc.grad_hist))]
    for a, b in zip(fast, ref):
        assert np.array_equal(bits(a), bits(b))


def test_multi_model_detect_shares_one_pyramid():
    """waldboost.detect (reference __init__.py:75-130): level-major, then model, then row-major."""
    img = synth_image(300, 420, 77)
    models = [random_model(500 + k, 20 + 5 * k, 2 if k else 1) for k in range(3)]
    rs = [1.0, 2.5, 0.5]
    out = wb.detect(img, *models, response_scale=rs)
    refs = [oracle_detect(M, img) for M in models]
    boxes, scores, labels = [], [], []
    n_levels = refs[0]["alive"].shape[0]
    for lv in range(n_levels):
        for k, r in enumerate(refs):
            sel = r["level"] == lv
            if sel.any():
                boxes.append(r["boxes"][sel])
                scores.append(r["scores"][sel] * np.float32(rs[k]))
                labels.append(np.full(int(sel.sum()), k, np.int64))
    assert len(out) == sum(len(b) for b in boxes) > 0
    assert np.array_equal(out.get(), np.concatenate(boxes))
    assert np.array_equal(bits(out.get_field("scores")), bits(np.concatenate(scores)))
    assert np.array_equal(out.get_field("label"), np.concatenate(labels))
    for M, r in zip(models, refs):
        assert M.n_loc == r["n_loc"] and M.n_weak == r["n_weak"]
    with pytest.raises(ValueError):
        wb.detect(img, *models, response_scale=[1.0])


def compose_multi(models, refs, rs):
    """The reference's nested loops (__init__.py:118-128) over per-model results: level-major, then model."""
    boxes, scores, labels = [np.empty((0, 4), np.float32)], [np.empty(0, np.float32)], [np.empty(0, np.int64)]
    for lv in range(refs[0]["alive"].shape[0]):
        for k, r in enumerate(refs):
            sel = r["level"] == lv
            boxes.append(r["boxes"][sel])
            scores.append(r["scores"][sel] * np.float32(rs[k]))
            labels.append(np.full(int(sel.sum()), k, np.int64))
    return np.concatenate(boxes), np.concatenate(scores), np.concatenate(labels)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_multi_model_detect_repeated_calls_replay_one_graph(dtype):
    """waldboost.detect again and again with the same models (eager call, captured call, replays), other users of the
    same engine in between (Model.detect of one of the models and of a third one: they leave their own ranks, dirty
    octave keys and their own scan states behind): every call equals the composition of the oracle's per-model results."""
    models = [random_model(700 + k, 24 + 8 * k, 2) for k in range(2)]
    other = random_model(777, 16, 1)
    rs = [1.0, 0.75]
    total = 0
    for i in range(6):
        img = synth_image(260, 380, 880 + i).astype(dtype)
        if i == 3:
            models[1].detect(img)
            other.detect(synth_image(260, 380, 3).astype(dtype))
        before = [(M.n_loc, M.n_weak) for M in models]
        out = wb.detect(img, *models, response_scale=rs)
        refs = [oracle_detect(M, img) for M in models]
        boxes, scores, labels = compose_multi(models, refs, rs)
        assert np.array_equal(out.get(), boxes), f"call {i}"
        assert np.array_equal(bits(out.get_field("scores")), bits(scores)), f"call {i}"
        assert np.array_equal(out.get_field("label"), labels), f"call {i}"
        for M, r, (l0, w0) in zip(models, refs, before):
            assert (M.n_loc - l0, M.n_weak - w0) == (r["n_loc"], r["n_weak"])
        total += len(out)
    assert total > 0


def test_multi_model_detect_rescans_only_the_cascade_whose_results_did_not_fit(monkeypatch):
    """waldboost.detect's one-wait sequence with a detection buffer too small for ONE of the two models: that cascade
    alone is scanned again on the pyramid the sequence left resident (no second channel launch), the other model keeps
    its results -- and the composition still equals the oracle's, call after call while the buffer grows."""
    from waldboost_amd import engine as E, _native as nat
    few, many = random_model(910, 30, 2), random_model(911, 10, 2)
    many.theta = [float("-inf")] * len(many.theta)            # every window of every level is a detection
    few.theta = [t + 0.6 if np.isfinite(t) else t for t in few.theta]
    rs = [1.0, 0.5]
    E._ENGINES.clear()
    launches = []
    real = E.PyramidEngine.launch_channels
    monkeypatch.setattr(E.PyramidEngine, "launch_channels", lambda self, *a, **k: (launches.append(1), real(self, *a, **k))[1])
    for i in range(4):
        img = synth_image(200, 260, 940 + i)
        if i == 0:
            wb.detect(img, few, many, response_scale=rs)     # (engine and scan states exist from here on)
            eng = next(iter(E._ENGINES.values()))
            eng.det_capacity = 64                            # far below `many`'s detections per shard
            eng._alloc_det()
        n0 = len(launches)
        out = wb.detect(img, few, many, response_scale=rs)
        refs = [oracle_detect(M, img) for M in (few, many)]
        assert refs[1]["scores"].size > 64 * nat.WB_DET_SHARDS // 8 and refs[1]["scores"].size > refs[0]["scores"].size
        boxes, scores, labels = compose_multi((few, many), refs, rs)
        assert np.array_equal(out.get(), boxes) and np.array_equal(bits(out.get_field("scores")), bits(scores)), f"call {i}"
        assert np.array_equal(out.get_field("label"), labels), f"call {i}"
        assert len(launches) - n0 == 1, f"call {i}: the pyramid was built {len(launches) - n0} times"


# ------------------------------------------------------------------------------ batches / configs
@pytest.mark.parametrize("ordered_on_device", [True, False])
def test_detect_batch_equals_per_image_detect(ordered_on_device, monkeypatch):
    """ordered_on_device: the batch's results split by image and ordered by wb_det_order_batch_launch (one read-back);
    False: the form it falls back to (device sort of all records, wb_boxes_launch)."""
    import waldboost_amd.model as wm
    monkeypatch.setattr(wm, "_ORDER_BATCH", ordered_on_device)
    imgs = np.stack([synth_image(210, 290, 600 + b) for b in range(5)])
    M = random_model(42, 30, 2)
    per = []
    for b in range(5):
        r = M.detect_raw(imgs[b])
        per.append(r)
    n_loc1, n_weak1 = M.n_loc, M.n_weak
    M.reset()
    res = M.detect_batch_raw(imgs)
    assert (M.n_loc, M.n_weak) == (n_loc1, n_weak1)
    for b in range(5):
        sel = res["image"] == b
        assert np.array_equal(res["level"][sel], per[b]["level"]) and np.array_equal(res["r"][sel], per[b]["r"])
        assert np.array_equal(res["c"][sel], per[b]["c"]) and np.array_equal(bits(res["scores"][sel]), bits(per[b]["scores"]))
        assert np.array_equal(bits(res["boxes"][sel]), bits(per[b]["boxes"]))
        assert np.array_equal(res["alive"][b], per[b]["alive"])
    bxs = M.detect_batch(imgs)
    assert [len(b) for b in bxs] == [int((res["image"] == b).sum()) for b in range(5)]


def test_full_size_batch_properties():
    """BASELINE configs[2] shape (a batch of 1080p images, 128-stage depth-2 cascade) through the
    size-independent properties of the path: the batch result is the concatenation of the per-image
    results (bit for bit), detections come sorted by (image, level, r, c), the statistics add up
    (alive[.,.,0] = every window, non-increasing over the stages, last column >= detections), and a
    second run reproduces the first exactly."""
    M = wb.load(os.path.join(GOLDEN, "models", "cfg2_d2_T128.pb"))
    B = 16
    imgs = np.stack([synth_image(1080, 1920, 1000 + b) for b in range(B)])
    res = M.detect_batch_raw(imgs)
    n_loc, n_weak = M.n_loc, M.n_weak
    assert n_loc == B * 3045278
    key = np.stack([res["image"], res["level"], res["r"], res["c"]], 1).astype(np.int64)
    order = np.lexsort((key[:, 3], key[:, 2], key[:, 1], key[:, 0]))
    assert np.array_equal(order, np.arange(key.shape[0]))
    alive = res["alive"]                                        # [B, levels, stages]
    assert alive[:, :, 0].sum() == n_loc and alive.sum() == n_weak
    assert (np.diff(alive, axis=2) <= 0).all()
    # windows alive after the last stage = detections: the last stage's theta of this model is -inf or calibrated,
    # so count them from the records
    per_image = np.bincount(res["image"], minlength=B)
    assert (alive[:, :, -1].sum(axis=1) >= per_image).all() and per_image.min() > 0
    M.reset()
    again = M.detect_batch_raw(imgs)
    for k in ("image", "level", "r", "c"):
        assert np.array_equal(res[k], again[k])
    assert np.array_equal(bits(res["scores"]), bits(again["scores"])) and np.array_equal(res["alive"], again["alive"])
    M.reset()
    for b in (0, 7, B - 1):
        one = M.detect_raw(imgs[b])
        sel = res["image"] == b
        assert np.array_equal(res["level"][sel], one["level"]) and np.array_equal(res["r"][sel], one["r"])
        assert np.array_equal(res["c"][sel], one["c"]) and np.array_equal(bits(res["scores"][sel]), bits(one["scores"]))
        assert np.array_equal(bits(res["boxes"][sel]), bits(one["boxes"])) and np.array_equal(res["alive"][b], one["alive"])


def test_config5_4k_shrink4_256_stages_vs_oracle():
    """BASELINE configs[4]: 3840x2160, shrink=4 (extension: the reference asserts shrink in [1,2]),
    n_per_oct=12, 256-stage depth-2 cascade, survival ~1e-4 -- against the oracle at full size."""
    M = wb.load(os.path.join(GOLDEN, "models", "cfg5_d2_T256.pb"))
    assert len(M) == 256 and M.channel_opts["shrink"] == 4 and M.channel_opts["n_per_oct"] == 12
    img = synth_image(2160, 3840, 0)
    res = M.detect_raw(img)
    ref = oracle_detect(M, img)
    assert ref["n_loc"] == 4435665 == M.n_loc and ref["alive"].shape == (108, 256)
    assert_same_detections(res, ref)
    assert M.n_weak == ref["n_weak"]


def _random_deep_tree(rng, shape, max_depth, p_leaf=0.25):
    """Random binary tree in pre-order (parent index < child index, like sklearn's), depth <= max_depth."""
    m, n, C = shape
    feat, thr, left, right, pred = [], [], [], [], []

    def grow(d):
        i = len(feat)
        feat.append((int(rng.integers(0, m)), int(rng.integers(0, n)), int(rng.integers(0, C))))
        thr.append(float(rng.uniform(2.0, 60.0)))
        pred.append(float(rng.uniform(0.2, 1.0) * rng.choice([-1.0, 1.0])))
        left.append(-1)
        right.append(-1)
        if d < max_depth and (d == 0 or rng.random() > p_leaf):
            left[i] = grow(d + 1)
            right[i] = grow(d + 1)
        else:
            feat[i] = None
        return i

    grow(0)
    return feat, thr, left, right, pred


def test_deep_trees_use_the_generic_kernel():
    rng = np.random.default_rng(77)
    shape = (12, 12, 4)
    M = wb.Model(shape, dict(wb.default_channel_opts))
    for t in range(18):
        f, th, l, r, p = _random_deep_tree(rng, shape, 4 + t % 3)
        M.append(wb.DTree(f, th, l, r, p), float("-inf") if t % 4 == 3 else float(np.float32(-0.4 * (t + 1))))
    assert max(w.depth() for w in M.classifier) >= 5 and M.device_cascade().depth >= 5
    img = synth_image(230, 310, 123)
    res = M.detect_raw(img)
    ref = oracle_detect(M, img)
    assert ref["scores"].size > 0
    assert_same_detections(res, ref)
    assert M.n_loc == ref["n_loc"] and M.n_weak == ref["n_weak"]
    # and on a caller-supplied channel image
    X = np.random.default_rng(3).uniform(0, 60, (40, 100, 4)).astype(np.float32)
    shape_, _, trees, thetas = oracle_model(M)
    rs, cs, hs, alive = M.predict_on_image_stats(X)
    ors, ocs, ohs, oalive = orc.cascade_predict_on_image(shape_, trees, thetas, X)
    assert np.array_equal(rs, ors) and np.array_equal(cs, ocs) and np.array_equal(bits(hs), bits(ohs)) and np.array_equal(alive, oalive)


@pytest.mark.parametrize("shape,C", [((12, 12, 3), 3), ((9, 20, 1), 1), ((6, 6, 7), 7)])
def test_predict_on_image_other_channel_counts(shape, C):
    rng = np.random.default_rng(31 + C)
    X = rng.uniform(0, 60, (50, 120, C)).astype(np.float32)
    M = random_model(60 + C, 25, 2, shape=shape)
    sh, _, trees, thetas = oracle_model(M)
    rs, cs, hs, alive = M.predict_on_image_stats(X)
    ors, ocs, ohs, oalive = orc.cascade_predict_on_image(sh, trees, thetas, X)
    assert ors.size > 0
    assert np.array_equal(rs, ors) and np.array_equal(cs, ocs) and np.array_equal(bits(hs), bits(ohs)) and np.array_equal(alive, oalive)


def test_window_too_large_for_an_lds_tile_falls_back_to_the_generic_kernel():
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 60, (150, 260, 4)).astype(np.float32)
    M = random_model(71, 12, 2, shape=(100, 120, 4))
    assert M.device_cascade().tile_rows == 4            # generic kernel geometry
    sh, _, trees, thetas = oracle_model(M)
    rs, cs, hs, alive = M.predict_on_image_stats(X)
    ors, ocs, ohs, oalive = orc.cascade_predict_on_image(sh, trees, thetas, X)
    assert ors.size > 0
    assert np.array_equal(rs, ors) and np.array_equal(cs, ocs) and np.array_equal(bits(hs), bits(ohs)) and np.array_equal(alive, oalive)


# ------------------------------------------------------------------------------ non-finite pixels
@pytest.mark.parametrize("kind", ["inf", "nan", "huge"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_channel_pyramid_and_detection_with_non_finite_pixels_vs_oracle(dtype, kind):
    """Float images holding inf / NaN / nearly overflowing pixels (reference channels.py:132: scipy's zoom multiplies
    every tap, so an identity level is NOT a copy next to an infinite pixel -- 0 * inf = NaN -- and NumPy's clip hands a
    NaN min / max on to the whole level; :61-64 pools them into the octaves; :52 np.fmax turns NaN gradients into 0).
    Every level of the pyramid against the oracle, bit for bit where it is not NaN, NaN where it is; then a detection."""
    from test_oracle import nonfinite_image
    from waldboost_amd.channels import channel_pyramid
    import scipy.ndimage as ndi
    img = nonfinite_image((136, 200), dtype, kind)
    opts = dict(wb.default_channel_opts)
    with np.errstate(invalid="ignore", over="ignore"):
        # (the oracle's own resize on this very image against SciPy, on this machine: level 1 of octave 0)
        nh, nw = 124, 182
        z = np.clip(ndi.zoom(img, [nh / 136, nw / 200], order=1, mode="mirror", grid_mode=True), img.min(), img.max())
        o = orc.resize_bilinear(img, nh, nw)
        assert np.array_equal(np.isnan(o), np.isnan(z)) and np.array_equal(o[~np.isnan(z)], z[~np.isnan(z)])
        ref = list(orc.channel_pyramid(img, dict(opts, channels=orc.grad_hist)))
    got = list(channel_pyramid(img, opts))
    assert len(got) == len(ref) > 8
    n_special = 0
    for (c, s), (rc, rs) in zip(got, ref):
        assert s == rs and c.shape == rc.shape
        nan = np.isnan(rc)
        assert np.array_equal(np.isnan(c), nan)
        assert np.array_equal(bits(c)[~nan], bits(rc)[~nan])
        n_special += int(np.isinf(rc).sum()) + int(nan.sum())
    assert kind != "inf" or n_special > 0
    M = random_model(21, 20, 2)
    with np.errstate(invalid="ignore", over="ignore"):
        want = oracle_detect(M, img)
    assert_same_detections(M.detect_raw(img), want)
