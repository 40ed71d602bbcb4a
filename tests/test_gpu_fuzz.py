"""Randomised end-to-end parity: random image sizes and dtypes, channel functions, pyramid options, window shapes,
tree shapes and cascades -- Model.detect on the GPU against the oracle, bit for bit.  Fixed seeds."""
import numpy as np
import pytest

import waldboost_amd as wb
from waldboost_amd.synth import random_tree_arrays, synth_image
from util import oracle_detect

pytestmark = pytest.mark.gpu

FUNCS = [("grad_hist", 4, np.float32, (2.0, 60.0)), ("grad_hist_4_u1", 4, np.uint8, (0.5, 20.0)),
         ("grad_mag_u1", 1, np.uint8, (1.0, 40.0)), ("grad_mag", 1, np.float32, (0.3, 1.8))]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_configuration(seed, max_depth=3):
    """-> (model, image) of the seed's random configuration (trees of depth 1..max_depth)."""
    rng = np.random.default_rng(1000 + seed)
    key, C, _, (lo, hi) = FUNCS[seed % 4]
    func = wb.channels.SPECS[key].func
    img_dtype = np.uint8 if (key.endswith("_u1") or rng.random() < 0.6) else np.float32
    H, W = int(rng.integers(40, 400)), int(rng.integers(40, 400))
    img = synth_image(H, W, seed, img_dtype)
    if img_dtype == np.float32:
        if key == "grad_hist":
            lo, hi = lo / 255.0, hi / 255.0              # float images are in [0, 1]: scale the thresholds into range
    if rng.random() < 0.3 and img_dtype == np.uint8:
        img[: H // 3, : W // 2] = int(rng.integers(0, 256))   # a flat region (zero gradients, pooled zeros)
    opts = dict(shrink=int(rng.choice([1, 2, 2, 2])), n_per_oct=int(rng.integers(1, 9)), smooth=int(rng.integers(0, 2)), channels=func)
    m, n = int(rng.integers(5, 25)), int(rng.integers(5, 25))
    shape = (m, n, C)
    M = wb.Model(shape, opts)
    T = int(rng.integers(3, 50))
    acc, step = 0.0, float(rng.uniform(-0.5, 0.1))
    for t in range(T):
        depth = int(rng.integers(1, max_depth + 1))
        f, th, l, r, p = random_tree_arrays(rng, shape, depth, lo, hi, unbalanced=(depth == 2 and rng.random() < 0.3))
        acc += step
        M.append(wb.DTree(f, th, l, r, p), float("-inf") if rng.random() < 0.2 else float(np.float32(acc)))
    return M, img


def check_against_oracle(M, img, counters=True):
    ref = oracle_detect(M, img)
    res = M.detect_raw(img)
    assert np.array_equal(res["alive"], ref["alive"]), (img.dtype, img.shape, M.channel_opts, M.shape, len(M))
    assert np.array_equal(res["level"], ref["level"]) and np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"])
    assert np.array_equal(bits(res["scores"]), bits(ref["scores"]))
    assert np.array_equal(bits(res["boxes"]), bits(ref["boxes"]))
    if counters:                                  # (the model's counters add up over its scans, as the reference's do)
        assert M.n_loc == ref["n_loc"] and M.n_weak == ref["n_weak"]


@pytest.mark.parametrize("seed", range(64))
def test_random_configuration(seed):
    check_against_oracle(*random_configuration(seed))


# Seeds 558, 569 and 644 are the three of seeds 464..700 that failed in round 4 when the cascade kernel was specialised
# before the first scan: depth-3 trees under a permissive cascade (10k-89k detections, tiles with more than 1024
# survivors after eight stages).  The code one of the two compilers produced for them wrote a few wrong records per such
# tile; a specialised kernel is now self-tested against the generic one before use, and a model whose build fails stays on
# the generic kernel (csrc/wb_api.hip jit_selftest).  Three passes each: the damage came and went.
@pytest.mark.parametrize("seed", [558, 569, 644, 5, 18, 41])
def test_random_configuration_through_the_specialised_kernel(seed):
    M, img = random_configuration(seed)
    dm = M.device_cascade()
    took = dm.specialize()                   # (False: the build failed its self-test -- seed 558's does -- and the model stays generic)
    assert bool(dm.specialized()) == bool(took)
    for k in range(3):
        check_against_oracle(M, img, counters=k == 0)


# Nearly every model above holds a depth-3 tree somewhere, so its kernels are the depth-3 ones.  The kernels the benchmark
# runs on are the depth-2 (and depth-1) ones: the same random configurations with the depth bounded, each through the
# generic kernel and then through the specialised one.
@pytest.mark.parametrize("seed", range(700, 748))
def test_random_configuration_of_bounded_depth_on_both_kernels(seed):
    M, img = random_configuration(seed, max_depth=1 + seed % 2)
    check_against_oracle(M, img)
    dm = M.device_cascade()
    if dm.specialize():                      # (False: a model without a specialised kernel -- float32 channels without ranks)
        for k in range(2):
            check_against_oracle(M, img, counters=False)
