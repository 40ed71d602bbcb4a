#!/usr/bin/env python3
"""Golden fixtures for the image dtypes the reference's channel_pyramid accepts besides uint8 and float32
(reference channels.py:122 keeps ``image.dtype`` through ``_image_octaves`` / ``resize(...).astype(dtype)``):
float64 and the integer types, every level of a few small pyramids, run through the reference's own
``channel_pyramid`` source.

Same method, stand-ins and caveats as make_golden.py (imported from there); run in the build container only:
``python tests/golden/make_golden_dtypes.py``.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def images():
    from waldboost_amd.synth import synth_image
    rng = np.random.default_rng(42)
    base = synth_image(97, 131, 21).astype(np.float64)
    yield "f64_97x131", base * 0.37 + rng.normal(0, 1e-3, base.shape), 2, 8, 1
    yield "f64_neg_60x88_s1", (synth_image(60, 88, 22).astype(np.float64) - 128.0) * 1.7, 1, 3, 1
    yield "i16_97x131", (synth_image(97, 131, 23).astype(np.int32) * 250 - 30000).astype(np.int16), 2, 8, 1      # sums wrap
    yield "u16_80x104_nosmooth", (synth_image(80, 104, 24).astype(np.uint32) * 257).astype(np.uint16), 2, 4, 0   # sums wrap
    yield "i8_64x96", (synth_image(64, 96, 25).astype(np.int16) - 128).astype(np.int8), 2, 4, 1
    yield "i32_72x90", (synth_image(72, 90, 26).astype(np.int64) * 16000000 - 2000000000).astype(np.int32), 2, 4, 1   # beyond float32 integers, sums wrap
    yield "u32_64x80_s1", (synth_image(64, 80, 27).astype(np.uint64) * 16843009).astype(np.uint32), 1, 2, 1


def main():
    mg.import_reference()
    from waldboost.channels import channel_pyramid, grad_hist
    meta = {"numpy": np.__version__, "scipy": __import__("scipy").__version__, "cases": {}}
    out = {}
    for name, img, shrink, npo, smooth in images():
        o = dict(shrink=shrink, n_per_oct=npo, smooth=smooth, channels=grad_hist)
        with np.errstate(over="ignore"):
            lv = list(channel_pyramid(img, o))
        out[f"{name}/image"] = img
        meta["cases"][name] = dict(dtype=str(img.dtype), shrink=shrink, n_per_oct=npo, smooth=smooth, n_levels=len(lv),
                                   scales=[float(s) for _, s in lv], shapes=[list(c.shape) for c, _ in lv])
        for i, (c, s) in enumerate(lv):
            assert c.dtype == np.float32
            out[f"{name}/L{i}"] = c
    # channel functions called directly with non-default arguments (reference channels.py:30-52)
    from waldboost.channels import grad_mag
    from waldboost_amd.synth import synth_image
    fimgs = {"u8": synth_image(61, 83, 31), "f32": synth_image(50, 70, 32, np.float32)}
    calls = [("grad_hist", dict(n_bins=6, full=True, bias=3)), ("grad_hist", dict(n_bins=9, full=False, bias=0.5)),
             ("grad_hist", dict(n_bins=1)), ("grad_hist", dict(n_bins=4, full=True)), ("grad_mag", dict(norm=None)),
             ("grad_mag", dict(norm=1)), ("grad_mag", dict(norm=3, eps=0.01)), ("grad_mag", dict(norm=8)),
             ("grad_mag", dict(norm=30, eps=1.0))]
    fout, fmeta = {}, []
    for k, im in fimgs.items():
        fout[f"image/{k}"] = im
        for j, (fn, kw) in enumerate(calls):
            r = (grad_hist if fn == "grad_hist" else grad_mag)(im, **kw)
            assert r.dtype == np.float32
            fout[f"{k}/{j}"] = r
            fmeta.append(dict(image=k, index=j, func=fn, kwargs=kw, shape=list(r.shape)))
    np.savez_compressed(os.path.join(HERE, "chanfunc_args.npz"), **fout)
    meta["chanfunc_args"] = fmeta
    np.savez_compressed(os.path.join(HERE, "pyramids_dtypes.npz"), **out)
    with open(os.path.join(HERE, "golden_meta_dtypes.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print({k: v["n_levels"] for k, v in meta["cases"].items()})


if __name__ == "__main__":
    main()
