"""The model-specialised cascade kernel (csrc/wb_jit.hip: hiprtc, the model's stage records as compile-time constants)
against the generic kernel and the oracle: indices, alive[level, stage], score bits and boxes must be identical --
reference model.py:216-259, training.py:84-96."""
import os

import numpy as np
import pytest

import waldboost_amd as wb
from waldboost_amd import _native as nat
from waldboost_amd.synth import random_tree_arrays, synth_image
from util import GOLDEN, oracle_detect

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_same(res, ref):
    assert np.array_equal(res["alive"], ref["alive"])
    assert np.array_equal(res["level"], ref["level"]) and np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"])
    assert np.array_equal(bits(res["scores"]), bits(ref["scores"])) and np.array_equal(bits(res["boxes"]), bits(ref["boxes"]))


def both_kernels(M, img):
    """detect_raw on the generic kernel, then on the specialised one; both against the oracle.  Returns the oracle's dict."""
    ref = oracle_detect(M, img)
    dm = M.device_cascade()
    assert not dm.specialized()
    assert_same(M.detect_raw(img), ref)                     # generic kernel (first scan)
    assert dm.specialize()
    kind = dm.rank_dtype if dm.rank_dtype is not None else nat.WB_DTYPE_U8
    assert kind in dm.specialized()
    for _ in range(2):                                      # eager, then the captured graph of Model.detect
        assert_same(M.detect_raw(img), ref)
    return ref


@pytest.mark.parametrize("path,shape", [("models/cfg2_d2_T128.pb", (1080, 1920)), ("mixed_d2_T24.pb", (240, 320)),
                                        ("cfg1_d1_T32.pb", (480, 640)), ("models/cfg2_gh4u1_d2_T128.pb", (540, 960)),
                                        ("grad_hist_4_u1_d2_T24.pb", (200, 264)), ("grad_mag_u1_d2_T24.pb", (200, 264))])
def test_specialised_kernel_equals_generic_and_oracle_on_the_committed_models(path, shape):
    M = wb.load(os.path.join(GOLDEN, path))
    ref = both_kernels(M, synth_image(*shape, 11))
    assert ref["scores"].size > 0


def random_model(seed, T, depth_of, C=4, func=None, lo=2.0, hi=60.0, theta_inf=0.2, shape=(12, 12)):
    rng = np.random.default_rng(seed)
    opts = dict(wb.default_channel_opts)
    if func is not None:
        opts["channels"] = func
    M = wb.Model((*shape, C), opts)
    acc, step = 0.0, float(rng.uniform(-0.4, 0.0))
    for t in range(T):
        d = depth_of(rng)
        f, th, l, r, p = random_tree_arrays(rng, (*shape, C), d, lo, hi, unbalanced=(d == 2 and rng.random() < 0.3))
        acc += step
        M.append(wb.DTree(f, th, l, r, p), float("-inf") if rng.random() < theta_inf else float(np.float32(acc)))
    return M


@pytest.mark.parametrize("seed,T,depth", [(0, 1, 2), (1, 5, 2), (2, 8, 1), (3, 9, 2), (4, 13, 3), (5, 16, 2), (6, 17, 1), (7, 31, 2),
                                          (8, 40, 3), (9, 70, 2), (10, 130, 2), (11, 200, 1), (12, 330, 2)])
def test_specialised_kernel_on_random_cascades_of_every_length_and_depth(seed, T, depth):
    """Phase A shorter than its 8 stages, lengths off the group size, every depth the tile kernel takes (1-3), segment
    chains up to 330 stages, never-rejecting stages: rank tiles (grad_hist) and uint8 channel tiles alternate."""
    u1 = seed % 2 == 1
    M = random_model(100 + seed, T, lambda rng: depth if depth < 3 else int(rng.integers(1, 4)),
                     func=wb.channels.SPECS["grad_hist_4_u1"].func if u1 else None, lo=0.5 if u1 else 2.0, hi=20.0 if u1 else 60.0)
    img = synth_image(150 + 13 * seed, 210 + 7 * seed, seed)
    both_kernels(M, img)


@pytest.mark.parametrize("T,quiet,u1", [(40, 20, False), (24, 24, False), (12, 12, True), (30, 9, True), (48, 33, False), (7, 7, False)])
def test_tiles_whose_survivors_overflow_the_capped_queue_go_on_densely(T, quiet, u1):
    """The workgroup's survivor queue holds 64 * waves + 512 entries, not one per window of the tile: a cascade whose first
    `quiet` stages never reject (theta = -inf, model.py:253) leaves all 2048 windows of a tile alive behind phase A, so
    the tile goes on densely, eight stages at a time, until the survivors fit -- or, when the cascade ends first (quiet
    == T: every window is a detection), emits them from the dense state.  Generic and specialised kernel, rank and uint8
    tiles, against the oracle."""
    M = random_model(300 + T + quiet, T, lambda rng: 2, theta_inf=0.0,
                     func=wb.channels.SPECS["grad_hist_4_u1"].func if u1 else None, lo=0.5 if u1 else 2.0, hi=20.0 if u1 else 60.0)
    for t in range(quiet):
        M.theta[t] = float("-inf")
    for t in range(quiet, T):                                # ... then reject hard, so that the late stages are queue work
        M.theta[t] = float(np.float32(0.45 * (t - quiet + 1)))
    img = synth_image(120, 170, 60 + T)
    ref = both_kernels(M, img)
    n_windows = ref["alive"][:, 0].sum()
    assert quiet == 0 or (ref["alive"][:, quiet - 1] == ref["alive"][:, 0]).all()
    if quiet == T:
        assert ref["scores"].size == n_windows               # every window survives every stage


def test_specialised_kernel_with_special_thresholds_and_leaf_values():
    """NaN / negative / infinite thresholds (rank -1 and always-left nodes), infinite and signed-zero leaf values: the
    specialised stages take them from literals and from the LDS mirror -- same bits as the generic walk."""
    M = random_model(7, 24, lambda rng: 2)
    specials = [np.nan, -1.0, np.inf, -np.inf, 0.0, -0.0]
    for i, (w, _) in enumerate(M):
        w.threshold[i % w.threshold.size] = specials[i % len(specials)]
        if i % 5 == 0:
            w.prediction[-1] = [np.inf, -0.0, 1e-30][i // 5 % 3]
    M.theta[3] = float("-inf")
    both_kernels(M, synth_image(200, 264, 5))


def test_auto_specialisation_after_a_few_scans_and_models_it_cannot_take():
    from waldboost_amd import engine
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    img = synth_image(200, 264, 2)
    ref = oracle_detect(M, img)
    dm = M.device_cascade()
    for i in range(3):       # Model.detect: the second call builds the specialised kernel, then captures its graph with it
        assert_same(M.detect_raw(img), ref)
        assert bool(dm.specialized()) == (i >= 1 and engine._JIT_AUTO)
    M2 = wb.load(os.path.join(GOLDEN, "grad_hist_4_u1_d2_T24.pb"))      # engine-level scans: after _JIT_AFTER of them
    e = engine.PyramidEngine(200, 264, np.uint8, 2, 8, 1, channels=wb.channels.channel_spec(M2.channel_opts["channels"]))
    e.load_images(img)
    dm2 = M2.device_cascade()
    for i in range(engine._JIT_AFTER + 1):
        e.run(dm2)
        assert bool(dm2.specialized()) == (i + 1 >= engine._JIT_AFTER and engine._JIT_AUTO)
    # a cascade of trees deeper than the tile kernel walks runs on the node-walk kernel: nothing to specialise
    from test_gpu_parity import _random_deep_tree
    rng = np.random.default_rng(3)
    deep = wb.Model((12, 12, 4), dict(wb.default_channel_opts))
    for t in range(6):
        deep.append(wb.DTree(*_random_deep_tree(rng, (12, 12, 4), 5)), float(np.float32(-0.4 * (t + 1))))
    assert deep.device_cascade().depth >= 4 and deep.device_cascade().specialize() is False
    assert_same(deep.detect_raw(img), oracle_detect(deep, img))
    # float32 channel tiles have no specialised kernel
    assert dm.specialize(nat.WB_DTYPE_F32) is False


@pytest.mark.parametrize("seed", [3, 17, 26, 41, 58, 63])
def test_random_configurations_with_the_cascade_specialised_on_its_first_scan(seed, monkeypatch):
    """The randomised end-to-end test (test_gpu_fuzz) with every eligible cascade compiled before its first scan
    (tools/fuzz_more.py with WB_CASC_JIT_AFTER=1 runs further seeds the same way)."""
    import test_gpu_fuzz as F
    from waldboost_amd import engine as E
    monkeypatch.setattr(E, "_JIT_AFTER", 1)
    monkeypatch.setattr(E, "_JIT_AUTO", True)
    F.test_random_configuration(seed)


def test_the_disk_cache_of_specialised_kernels_is_bounded_and_checks_what_it_loads(tmp_path, monkeypatch):
    """A loop that changes its model every iteration must not leave one code object per model content behind for ever:
    the cache directory is pruned, oldest first, to WB_JIT_CACHE_MAX files; a cached file that is not an ELF image is
    compiled again instead of being handed to the module loader."""
    monkeypatch.setenv("WB_JIT_CACHE", str(tmp_path))
    monkeypatch.setenv("WB_JIT_CACHE_MAX", "2")
    img = synth_image(120, 160, 3)
    for k in range(4):
        M = random_model(900 + k, 10, lambda rng: 2)
        assert M.device_cascade().specialize()
        assert_same(M.detect_raw(img), oracle_detect(M, img))
        assert len([f for f in os.listdir(tmp_path) if f.endswith(".co")]) <= 2
    # a damaged entry: the same model in a fresh process state would load it -- here: overwrite every cached file, build
    # a model whose kernel is not loaded yet, and see it compile instead of failing
    for f in os.listdir(tmp_path):
        with open(tmp_path / f, "wb") as fh:
            fh.write(b"not a code object")
    M = random_model(950, 10, lambda rng: 2)
    assert M.device_cascade().specialize()
    assert_same(M.detect_raw(img), oracle_detect(M, img))


def test_a_specialised_kernel_outlives_the_other_models_that_shared_it():
    """Models with the same content share one loaded module (wb_jit.hip counts its users; an idle module stays loaded until more than
    WB_JIT_MODULES_MAX are): destroying one model must leave the other's kernel in place, and a model created
    after both are gone gets its kernel again."""
    import gc
    path = os.path.join(GOLDEN, "mixed_d2_T24.pb")
    img = synth_image(240, 320, 21)
    A, B = wb.load(path), wb.load(path)
    ref = oracle_detect(A, img)
    for M in (A, B):
        assert M.device_cascade().specialize()
    assert_same(A.detect_raw(img), ref)
    del A
    gc.collect()
    for _ in range(2):
        assert_same(B.detect_raw(img), ref)                 # B's function is still loaded
    del B
    gc.collect()
    Cm = wb.load(path)
    assert Cm.device_cascade().specialize()
    for _ in range(2):
        assert_same(Cm.detect_raw(img), ref)


def test_a_fresh_specialised_kernel_is_cross_checked_on_the_first_real_image(monkeypatch, caplog):
    """Model.detect builds the specialised kernel on its second call, before it captures its graph -- and right there runs
    the step twice more on the image at hand: specialised kernel against generic kernel, records and alive counts compared
    on the device (PyramidEngine.live_check).  Agreement: the kernel is kept.  A disagreement (forced here by corrupting
    what the check reads back from the specialised pass) switches it off for the cascade, with a warning, and every result
    stays the oracle's."""
    import logging
    from waldboost_amd import engine as E
    path = os.path.join(GOLDEN, "mixed_d2_T24.pb")
    imgs = [synth_image(240, 320, 30 + i) for i in range(3)]
    # 1. the normal case
    E._ENGINES.clear()
    M = wb.load(path)
    refs = [oracle_detect(M, im) for im in imgs]
    dm = M.device_cascade()
    kind = dm.rank_dtype if dm.rank_dtype is not None else nat.WB_DTYPE_U8
    for im, ref in zip(imgs, refs):
        assert_same(M.detect_raw(im), ref)
    assert kind in dm.specialized() and dm.live_checks_left(kind) == 0 and not dm.__dict__.get("_spec_off")
    # 2. a kernel that "disagrees"
    E._ENGINES.clear()
    M2 = wb.load(path)
    dm2 = M2.device_cascade()
    real, calls = E.sort_records, []

    def corrupted(d):
        out = real(d)
        calls.append(1)
        if len(calls) == 1 and out.shape[0]:                 # the specialised pass of the first check
            out = out.clone()
            out[0, 3] += 1                                   # one score bit
        return out
    monkeypatch.setattr(E, "sort_records", corrupted)
    with caplog.at_level(logging.WARNING, logger=E._log.name):
        for im, ref in zip(imgs, refs):
            assert_same(M2.detect_raw(im), ref)
    assert dm2.__dict__.get("_spec_off") is True and dm2.live_checks_left(kind) == 0
    assert any("switched off" in r.getMessage() for r in caplog.records)
    monkeypatch.setattr(E, "sort_records", real)
    for im, ref in zip(imgs, refs):                          # the generic kernel from here on
        assert_same(M2.detect_raw(im), ref)
