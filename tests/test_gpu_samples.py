"""GPU parity of the training-time callers of the hot path: gather_samples,
Model.predict, DTree.apply/predict and get_samples_from_image, against fixtures produced by the
reference's own code (tests/golden/make_golden_f4.py) and against the CPU oracle.  Crops and leaf
indices are bit/value-exact; responses bit-exact (same fp32 accumulation order)."""
import os

import numpy as np
import pytest

import waldboost_amd as wb
from oracle import wb_oracle as orc
from waldboost_amd.samples import (SampleLabel, SamplePool, gather_samples, get_samples_from_image, label_boxes)
from waldboost_amd.synth import synth_image
from util import GOLDEN, oracle_detect, oracle_model

pytestmark = pytest.mark.gpu

CASES = [("f32", "mixed_d2_T24.pb", "mixed_200x264.npz"), ("u8", "grad_hist_4_u1_d2_T24.pb", "grad_hist_4_u1_200x264.npz")]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("tag,pb,npz", CASES)
def test_samples_vs_reference_fixture(tag, pb, npz):
    z = np.load(os.path.join(GOLDEN, "samples_f4.npz"))
    M = wb.load(os.path.join(GOLDEN, pb))
    levels = list(M.channels(np.load(os.path.join(GOLDEN, npz))["image"]))
    for li in (0, 5):
        k = f"{tag}/L{li}"
        X = gather_samples(levels[li][0], z[f"{k}/rs"], z[f"{k}/cs"], M.shape)
        assert X.dtype == z[f"{k}/X"].dtype and X.shape == z[f"{k}/X"].shape and np.array_equal(X, z[f"{k}/X"])
        H, mask = M.predict(X)
        assert mask.dtype == bool and np.array_equal(mask, z[f"{k}/mask"])
        assert np.array_equal(bits(H), bits(z[f"{k}/H"]))
        for t in (0, 3, 10):
            assert np.array_equal(M.classifier[t].apply(X), z[f"{k}/apply{t}"])
            assert np.array_equal(bits(M.classifier[t].predict(X)), bits(z[f"{k}/predict{t}"]))


@pytest.mark.parametrize("C,dtype", [(4, np.float32), (4, np.uint8), (1, np.uint8), (1, np.float32), (3, np.float32), (5, np.uint8)])
def test_gather_samples_vs_oracle(C, dtype):
    rng = np.random.default_rng(C)
    chns = (rng.random((70, 93, C)) * 200).astype(dtype)
    for shape in [(12, 12, C), (7, 20, C), (70, 93, C)]:
        m, n, _ = shape
        N = 33
        rs, cs = rng.integers(0, 70 - m + 1, N), rng.integers(0, 93 - n + 1, N)
        got = gather_samples(chns, rs, cs, shape)
        ref = orc.gather_samples(chns, rs, cs, shape)
        assert got.dtype == ref.dtype and got.shape == ref.shape and np.array_equal(got, ref)
    assert gather_samples(chns, np.zeros(0, int), np.zeros(0, int), (12, 12, C)).shape == (0, 12, 12, C)
    with pytest.raises(ValueError):
        gather_samples(chns, np.zeros(2, int), np.zeros(3, int), (12, 12, C))
    with pytest.raises(IndexError):
        gather_samples(chns, np.array([60]), np.array([0]), (12, 12, C))


def test_model_predict_vs_oracle_many_samples():
    M = wb.load(os.path.join(GOLDEN, "models", "cfg2_d2_T128.pb"))
    shape, opts, trees, thetas = oracle_model(M)
    img = synth_image(240, 320, 3)
    chns = next(iter(M.channels(img)))[0]
    rng = np.random.default_rng(0)
    N = 5000
    rs, cs = rng.integers(0, chns.shape[0] - 12, N), rng.integers(0, chns.shape[1] - 12, N)
    X = gather_samples(chns, rs, cs, M.shape)
    H, mask = M.predict(X)
    Hr, mr = orc.model_predict(shape, trees, thetas, orc.gather_samples(chns, rs, cs, shape))
    assert np.array_equal(mask, mr) and np.array_equal(bits(H), bits(Hr))
    # the per-sample cascade agrees with the dense scan at the same windows
    r, c, h = M.predict_on_image(chns)
    keep = {(int(a), int(b)): float(s) for a, b, s in zip(r, c, h)}
    for i in np.flatnonzero(mask):
        assert keep[(int(rs[i]), int(cs[i]))] == float(H[i])
    assert M.predict(X[:0])[0].shape == (0,)
    with pytest.raises(AssertionError):
        M.predict(X[:, :11])


def test_get_samples_from_image_matches_detections_and_crops():
    g = np.load(os.path.join(GOLDEN, "mixed_200x264.npz"))
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    img, det = g["image"], g["det"]
    gt = wb.Boxes(np.array([[40, 40, 120, 120]], "f"))
    np.random.seed(0)
    got = list(get_samples_from_image(M, img, gt, max_tp_candidates=10 ** 6, max_fp_candidates=10 ** 6,
                                      min_tp_iou=0.5, max_fp_iou=0.5))
    ref_levels = list(orc.channel_pyramid(img, oracle_model(M)[1]))
    seen = 0
    for bx in got:
        lab = bx.get_field("tp_label")
        assert set(np.unique(lab)) <= {SampleLabel.TRUE_POSITIVE, SampleLabel.FALSE_POSITIVE}
        r, c = bx.get_field("row"), bx.get_field("col")
        # which level these came from: match (r, c, score) against the reference detections
        d = det[np.isin(det["r"], r) & np.isin(det["c"], c) & np.isin(det["score"], bx.get_field("scores"))]
        lv = int(np.bincount(d["level"]).argmax())
        X = orc.gather_samples(ref_levels[lv][0], r, c, M.shape)
        assert np.array_equal(bx.get_field("samples"), X)
        seen += len(bx)
    assert seen == det.size            # iou thresholds 0.5/0.5 with unlimited candidates keep every detection
    # SamplePool: collects, re-scores with Model.predict and hands the samples back
    pool = SamplePool(min_tp=5, min_fp=50, min_tp_iou=0.5, max_fp_iou=0.5)
    pool.update(M, [dict(image=img, groundtruth_boxes=gt)])
    X0, H0 = pool.get_false_positives()
    assert X0.shape[1:] == tuple(M.shape) and X0.shape[0] == H0.size > 0
    pool.update_scores(M)
    H1, mask = M.predict(X0)
    assert mask.all() and np.array_equal(bits(H1), bits(H0))    # survivors re-score to their detection scores


def test_label_boxes_without_groundtruth():
    bx = wb.Boxes(np.array([[0, 0, 10, 10], [5, 5, 20, 20]], "f"))
    label_boxes(bx, None)
    assert np.array_equal(bx.get_field("tp_label"), [SampleLabel.FALSE_POSITIVE] * 2)
    assert np.array_equal(bx.get_field("instance_id"), [-1, -1])


def test_detect_on_images_yields_the_detect_result():
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    g = np.load(os.path.join(GOLDEN, "mixed_200x264.npz"))
    gt = wb.Boxes(np.array([[10, 10, 50, 50]], "f"))
    out = list(wb.testing.detect_on_images([dict(image=g["image"], groundtruth_boxes=gt), dict(image=g["image"])], M))
    assert len(out) == 2 and out[0][0] is gt and len(out[1][0]) == 0 and out[0][2] == (200, 264)
    for _, dt, _ in out:
        assert np.array_equal(dt.get_field("scores"), g["det"]["score"]) and (dt.get_field("label") == 0).all()
    assert wb.SamplePool is SamplePool
