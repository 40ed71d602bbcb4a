"""Shared helpers for the tests (the oracle is imported here and only in tests/)."""
import json
import os

import numpy as np

from oracle import wb_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_meta():
    with open(os.path.join(GOLDEN, "golden_meta.json")) as f:
        return json.load(f)


def oracle_model(M):
    """(shape, opts, trees, thetas) of a waldboost_amd.Model in the oracle's plain-array form."""
    trees = [orc.make_tree(w.feature, w.threshold, w.left, w.right, w.prediction) for w in M.classifier]
    from waldboost_amd.channels import channel_spec
    opts = dict(M.channel_opts)
    opts["channels"] = orc.CHANNEL_FUNCS[channel_spec(M.channel_opts["channels"]).key]
    return tuple(M.shape), opts, trees, list(M.theta)


def oracle_detect(M, image):
    shape, opts, trees, thetas = oracle_model(M)
    return orc.detect(shape, opts, trees, thetas, image)


def det_table(res):
    """(level, r, c, score) rows of a detect_raw()/oracle detect() result."""
    return np.stack([res["level"].astype(np.int64), res["r"].astype(np.int64), res["c"].astype(np.int64)], 1), res["scores"]


def small_cases():
    meta = golden_meta()["cases"]
    z = np.load(os.path.join(GOLDEN, "pyramids_small.npz"))
    for name, info in meta.items():
        img = z[f"{name}/image"]
        levels = [z[f"{name}/L{i}"] for i in range(info["n_levels"])]
        yield name, img, info, levels


def f2_meta():
    with open(os.path.join(GOLDEN, "golden_meta_f2.json")) as f:
        return json.load(f)


def f2_cases():
    """Small pyramids of the other channel functions (tests/golden/make_golden_f2.py)."""
    meta = f2_meta()["cases"]
    z = np.load(os.path.join(GOLDEN, "pyramids_f2.npz"))
    for name, info in meta.items():
        img = z[f"{name}/image"]
        levels = [z[f"{name}/L{i}"] for i in range(info["n_levels"])]
        yield name, img, info, levels


def dtype_cases():
    """Small grad_hist pyramids of float64 / integer / float16 / bool images (tests/golden/make_golden_dtypes.py,
    make_golden_dtypes2.py)."""
    for stem in ("dtypes", "dtypes2"):
        with open(os.path.join(GOLDEN, f"golden_meta_{stem}.json")) as f:
            meta = json.load(f)["cases"]
        z = np.load(os.path.join(GOLDEN, f"pyramids_{stem}.npz"))
        for name, info in meta.items():
            img = z[f"{name}/image"]
            levels = [z[f"{name}/L{i}"] for i in range(info["n_levels"])]
            yield name, img, info, levels


def chanfunc_arg_cases():
    """grad_hist / grad_mag called with non-default arguments on two small images (make_golden_dtypes.py)."""
    with open(os.path.join(GOLDEN, "golden_meta_dtypes.json")) as f:
        meta = json.load(f)["chanfunc_args"]
    z = np.load(os.path.join(GOLDEN, "chanfunc_args.npz"))
    for m in meta:
        yield f"{m['func']}-{m['image']}-{m['index']}", z[f"image/{m['image']}"], m["func"], m["kwargs"], z[f"{m['image']}/{m['index']}"]
