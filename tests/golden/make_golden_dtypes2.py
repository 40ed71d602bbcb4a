#!/usr/bin/env python3
"""Golden fixtures for the remaining image dtypes the reference's channel_pyramid accepts (reference channels.py:122
keeps ``image.dtype`` through ``_image_octaves`` / ``resize(...).astype(dtype)``): float16, bool, int64 and uint64
images, every level of a few small pyramids, run through the reference's own ``channel_pyramid`` source.

Same method, stand-ins and caveats as make_golden.py (imported from there); run in the build container only:
``python tests/golden/make_golden_dtypes2.py``.  (A file of its own so that the round-2 fixtures stay byte-identical.)
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
    yield "f16_72x100", (synth_image(72, 100, 41).astype(np.float64) * 0.37).astype(np.float16), 2, 4, 1      # every add rounds to float16
    yield "f16_large_64x80_s1", (synth_image(64, 80, 42).astype(np.float64) * 90.0).astype(np.float16), 1, 2, 1   # pooled sums overflow to inf
    yield "bool_64x96", synth_image(64, 96, 43) > 140, 2, 4, 1
    yield "bool_70x90_nosmooth", synth_image(70, 90, 44) > 110, 2, 3, 0
    yield "i64_72x90", synth_image(72, 90, 45).astype(np.int64) * 1000000007 - 120000000000, 2, 4, 1
    yield "u64_64x80_s1", synth_image(64, 80, 46).astype(np.uint64) * np.uint64(1 << 40), 1, 2, 1


def main():
    mg.import_reference()
    from waldboost.channels import channel_pyramid, grad_hist
    meta = {"numpy": np.__version__, "scipy": __import__("scipy").__version__, "cases": {}}
    out = {}
    for name, img, shrink, npo, smooth in images():
        o = dict(shrink=shrink, n_per_oct=npo, smooth=smooth, channels=grad_hist)
        with np.errstate(over="ignore", invalid="ignore"):
            lv = list(channel_pyramid(img, o))
        out[f"{name}/image"] = img
        meta["cases"][name] = dict(dtype=str(img.dtype), shrink=shrink, n_per_oct=npo, smooth=smooth, n_levels=len(lv),
                                   scales=[float(s) for _, s in lv], shapes=[list(c.shape) for c, _ in lv])
        for i, (c, s) in enumerate(lv):
            assert c.dtype == np.float32
            out[f"{name}/L{i}"] = c
    np.savez_compressed(os.path.join(HERE, "pyramids_dtypes2.npz"), **out)
    with open(os.path.join(HERE, "golden_meta_dtypes2.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print({k: v["n_levels"] for k, v in meta["cases"].items()})


if __name__ == "__main__":
    main()
