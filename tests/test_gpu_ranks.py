"""Float32 channels as threshold ranks (WB_DTYPE_RANK8): for the cascade that scans them, the channel kernel
writes each channel value as its rank among the model's sorted distinct thresholds of that channel, and a node
test `v <= thr` (reference training.py:92) becomes `rank(v) <= index(thr)`.  The decision must be the same for
EVERY float -- values equal to a threshold, one ulp either side, -0.0 / +0.0, +-inf, NaN -- and for every
threshold set: clustered (several thresholds in one lookup cell), duplicated, special values, more than 255 per
channel (then the float32 channels and the planar float tile are used).  Ranks are checked value by value against
np.searchsorted, detections against the oracle and against the float32-channel scan of the same model, bit for bit."""
import numpy as np
import pytest

import waldboost_amd as wb
from waldboost_amd import engine as _engine
from waldboost_amd import _native as nat
from waldboost_amd.channels import channel_spec, read_opts
from waldboost_amd.synth import random_tree_arrays, synth_image
from util import oracle_detect

pytestmark = pytest.mark.gpu
SHAPE = (12, 12, 4)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def model_with_thresholds(rng, T, depth, draw, theta_step=-0.2):
    """Random trees whose thresholds come from draw(n) -> float32[n]."""
    M = wb.Model(SHAPE, dict(wb.default_channel_opts))
    acc = 0.0
    for t in range(T):
        f, th, l, r, p = random_tree_arrays(rng, SHAPE, depth, 2.0, 60.0)
        th = np.where(l >= 0, draw(th.size), th).astype(np.float32)
        acc += theta_step
        M.append(wb.DTree(f, th, l, r, p), float("-inf") if t % 4 == 3 else float(np.float32(acc)))
    return M


def sorted_thresholds(M):
    """Per channel: the distinct non-NaN thresholds of the internal nodes, ascending (-0.0 and 0.0 are one)."""
    out = [[] for _ in range(SHAPE[2])]
    for w in M.classifier:
        for i in np.nonzero(w.left >= 0)[0]:
            if not np.isnan(w.threshold[i]):
                out[int(w.feature[i, 2])].append(np.float32(w.threshold[i]) + np.float32(0.0))
    return [np.unique(np.array(s, np.float32)) for s in out]


def check_ranks(M, img):
    """Every pixel of every level: rank bytes == number of the channel's thresholds below the float32 value.
    Returns the number of non-finite channel values met."""
    dm = M.device_cascade()
    assert dm.rank_dtype is not None
    nan_rank = 65535 if dm.rank_dtype == nat.WB_DTYPE_RANK16 else 255
    shrink, n_per_oct, smooth, spec = read_opts(M.channel_opts)
    eng = _engine.get_engine(img.shape[0], img.shape[1], img.dtype, shrink, n_per_oct, smooth, 1, channels=spec)
    eng.load_images(img)
    eng.run_channels(dm, floats=True)
    S = sorted_thresholds(M)
    n_nan = 0
    for l in range(eng.plan.n_levels):
        chn, rank = eng.read_level(0, l), eng.read_rank_level(0, l)
        for c in range(4):
            want = np.searchsorted(S[c], chn[..., c], side="left").astype(np.int64)
            want[np.isnan(chn[..., c])] = nan_rank
            assert np.array_equal(rank[..., c].astype(np.int64), want), (l, c)
        n_nan += int((~np.isfinite(chn)).sum())
    return n_nan


def check_detect(M, img):
    with np.errstate(invalid="ignore", over="ignore"):
        ref = oracle_detect(M, img)
    res = M.detect_raw(img)
    assert np.array_equal(res["alive"], ref["alive"])
    assert np.array_equal(res["level"], ref["level"]) and np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"])
    assert np.array_equal(bits(res["scores"]), bits(ref["scores"])) and np.array_equal(bits(res["boxes"]), bits(ref["boxes"]))
    return ref["scores"].size


def channel_values(img, n, rng):
    """n values that occur in the image's grad_hist pyramid, with their float neighbours."""
    vals = np.concatenate([chn.reshape(-1) for chn, _ in wb.channels.channel_pyramid(img, dict(wb.default_channel_opts))])
    v = rng.choice(vals[vals > 0], n).astype(np.float32)
    k = rng.integers(-1, 2, n)
    return np.where(k < 0, np.nextafter(v, np.float32(-np.inf)), np.where(k > 0, np.nextafter(v, np.float32(np.inf)), v)).astype(np.float32)


def test_thresholds_taken_from_the_channel_values_themselves():
    """Many pixels sit exactly on a threshold or one ulp beside it."""
    rng = np.random.default_rng(1)
    img = synth_image(200, 280, 11)
    M = model_with_thresholds(rng, 40, 2, lambda n: channel_values(img, n, rng))
    check_ranks(M, img)
    assert check_detect(M, img) > 0


def test_special_thresholds_and_float_images_that_overflow():
    rng = np.random.default_rng(2)
    special = np.array([np.nan, np.inf, -np.inf, -0.0, 0.0, 1e-30, 3.0e38], np.float32)
    M = model_with_thresholds(rng, 24, 2, lambda n: np.where(rng.random(n) < 0.3, rng.choice(special, n), rng.uniform(0, 60, n)),
                              theta_step=-0.3)
    img = synth_image(150, 210, 12)
    check_ranks(M, img)
    assert check_detect(M, img) > 0
    # a float32 image whose gradients overflow: infinite channel values (a NaN gradient is rectified to 0 by the
    # reference's np.fmax, channels.py:52, so grad_hist itself never yields NaN; the kernel still ranks NaN as 255)
    imf = (synth_image(120, 170, 13).astype(np.float32) * np.float32(0.25)).astype(np.float32)
    imf[40:44, 60:64] = 3.0e38
    imf[90, 20] = -3.0e38
    with np.errstate(invalid="ignore", over="ignore"):
        assert check_ranks(M, imf) > 0                       # there are infinite pixels
    # (pixel values next to FLT_MAX overflow in the octaves and in the gradients: inf and, through the zero-weight
    # taps of scipy's zoom and convolve1d, NaN -- the kernels follow the oracle there too)
    check_detect(M, imf)
    a = M.detect_raw(imf)
    shrink, n_per_oct, smooth, spec = read_opts(M.channel_opts)
    eng = _engine.get_engine(imf.shape[0], imf.shape[1], imf.dtype, shrink, n_per_oct, smooth, 1, channels=spec)
    eng.load_images(imf)
    eng.run_channels()
    b = M.scan_engine(eng)
    for k in ("level", "r", "c", "alive"):
        assert np.array_equal(a[k], b[k])
    assert np.array_equal(bits(a["scores"]), bits(b["scores"]))


@pytest.mark.parametrize("spread", [0.0, 1e-3, 0.5])
def test_clustered_thresholds_share_lookup_cells(spread):
    """Most thresholds within `spread` of a few centres, plus outliers that stretch the grid: several
    thresholds per cell (refinement steps > 1), or, when too many share one cell, no rank tables at all."""
    rng = np.random.default_rng(3)
    centres = np.array([7.25, 7.5, 30.0], np.float32)

    def draw(n):
        v = rng.choice(centres, n) + rng.integers(-6, 7, n).astype(np.float32) * np.float32(spread / 6 if spread else 0)
        return np.where(rng.random(n) < 0.1, rng.uniform(-1e4, 1e4, n), v)
    M = model_with_thresholds(rng, 48, 2, draw)
    img = synth_image(160, 230, 14)
    if M.device_cascade().rank_ok:
        check_ranks(M, img)
    check_detect(M, img)


def test_more_than_254_thresholds_per_channel_are_ranked_in_two_bytes():
    """A long soft cascade (reference __init__.py:230-269 appends stages without bound): more distinct thresholds per channel
    than a byte ranks -> WB_DTYPE_RANK16; value by value against np.searchsorted, detections against the oracle."""
    rng = np.random.default_rng(4)
    M = model_with_thresholds(rng, 400, 2, lambda n: rng.uniform(0, 60, n), theta_step=-0.05)
    dm = M.device_cascade()
    assert min(s.size for s in sorted_thresholds(M)) > 255 and not dm.rank_ok and dm.rank16_ok
    assert dm.rank_dtype == nat.WB_DTYPE_RANK16
    img = synth_image(130, 170, 15)
    check_ranks(M, img)
    check_detect(M, img)
    # thresholds on the channel values themselves, special values, clusters: the 16-bit tables' grid is coarser (several
    # thresholds per cell is the rule there)
    img2 = synth_image(200, 280, 11)
    special = np.array([np.nan, np.inf, -np.inf, -0.0, 0.0, 1e-30, 3.0e38], np.float32)

    def draw(n):
        v = channel_values(img2, n, rng)
        return np.where(rng.random(n) < 0.05, rng.choice(special, n), v)
    M2 = model_with_thresholds(rng, 360, 2, draw, theta_step=-0.05)
    assert M2.device_cascade().rank_dtype == nat.WB_DTYPE_RANK16
    check_ranks(M2, img2)
    assert check_detect(M2, img2) >= 0
    imf = (synth_image(120, 170, 13).astype(np.float32) * np.float32(0.25)).astype(np.float32)
    imf[40:44, 60:64] = 3.0e38
    with np.errstate(invalid="ignore", over="ignore"):
        assert check_ranks(M2, imf) > 0
    check_detect(M2, imf)


def test_a_1024_stage_cascade_runs_on_16_bit_rank_tiles_with_its_specialised_kernel():
    rng = np.random.default_rng(41)
    M = model_with_thresholds(rng, 1024, 2, lambda n: rng.uniform(0, 60, n), theta_step=-0.02)
    dm = M.device_cascade()
    assert dm.rank_dtype == nat.WB_DTYPE_RANK16 and max(s.size for s in sorted_thresholds(M)) <= 1020
    img = synth_image(150, 200, 17)
    a = M.detect_raw(img)                                    # generic kernel on the 16-bit tile
    assert dm.specialize() and nat.WB_DTYPE_RANK16 in dm.specialized()
    b = M.detect_raw(img)                                    # the model-specialised kernel
    for k in ("level", "r", "c", "alive"):
        assert np.array_equal(a[k], b[k])
    assert np.array_equal(bits(a["scores"]), bits(b["scores"]))
    check_detect(M, img)


def test_depth_3_trees_on_16_bit_rank_tiles():
    rng = np.random.default_rng(42)
    M = model_with_thresholds(rng, 128, 3, lambda n: rng.uniform(0, 60, n), theta_step=-0.1)
    dm = M.device_cascade()
    if dm.rank_ok:                                           # (896 thresholds spread over four channels may fit a byte)
        dm.rank_dtype = nat.WB_DTYPE_RANK16
    assert dm.rank16_ok
    img = synth_image(170, 230, 18)
    check_ranks(M, img)
    a = M.detect_raw(img)
    assert dm.specialize(nat.WB_DTYPE_RANK16) and nat.WB_DTYPE_RANK16 in dm.specialized()
    b = M.detect_raw(img)
    for k in ("level", "r", "c", "alive"):
        assert np.array_equal(a[k], b[k])
    assert np.array_equal(bits(a["scores"]), bits(b["scores"]))
    check_detect(M, img)


def test_the_same_cascade_in_one_and_in_two_byte_ranks():
    """A model both forms apply to, forced through the 16-bit one: identical results (engine buffers re-allocated between
    the forms, captured graphs dropped)."""
    rng = np.random.default_rng(43)
    M = model_with_thresholds(rng, 40, 2, lambda n: rng.uniform(0, 60, n))
    img = synth_image(200, 264, 19)
    dm = M.device_cascade()
    assert dm.rank_dtype == nat.WB_DTYPE_RANK8 and dm.rank16_ok
    a = [M.detect_raw(img) for _ in range(3)][-1]
    dm.rank_dtype = nat.WB_DTYPE_RANK16
    check_ranks(M, img)
    b = [M.detect_raw(img) for _ in range(3)][-1]
    dm.rank_dtype = nat.WB_DTYPE_RANK8
    c = M.detect_raw(img)
    for k in ("level", "r", "c", "alive"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k])
    assert np.array_equal(bits(a["scores"]), bits(b["scores"])) and a["scores"].size > 0


def test_more_than_1020_thresholds_per_channel_use_the_float_channels():
    rng = np.random.default_rng(5)
    M = model_with_thresholds(rng, 1500, 2, lambda n: rng.uniform(0, 60, n), theta_step=-0.01)
    dm = M.device_cascade()
    assert min(s.size for s in sorted_thresholds(M)) > 1020 and not dm.rank_ok and not dm.rank16_ok and dm.rank_dtype is None
    check_detect(M, synth_image(130, 170, 15))


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_rank_scan_equals_float_scan(depth):
    """One pyramid, both forms: the fused rank path and the float32 channels through the planar float tile."""
    rng = np.random.default_rng(8 + depth)
    M = model_with_thresholds(rng, 30, depth, lambda n: rng.uniform(0, 60, n), theta_step=-0.6 if depth == 1 else -0.2)
    img = synth_image(240, 330, 16 + depth)
    a = M.detect_raw(img)
    assert M.device_cascade().rank_ok and a["scores"].size > 0
    shrink, n_per_oct, smooth, spec = read_opts(M.channel_opts)
    eng = _engine.get_engine(img.shape[0], img.shape[1], img.dtype, shrink, n_per_oct, smooth, 1, channels=spec)
    eng.load_images(img)
    eng.run_channels()
    b = M.scan_engine(eng)                                   # float32 channels
    for k in ("level", "r", "c", "alive"):
        assert np.array_equal(a[k], b[k])
    assert np.array_equal(bits(a["scores"]), bits(b["scores"])) and np.array_equal(bits(a["boxes"]), bits(b["boxes"]))
    assert check_detect(M, img) > 0
