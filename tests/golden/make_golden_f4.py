#!/usr/bin/env python3
"""Golden fixtures for the training-time callers of the hot path: the
reference's own ``samples.gather_samples``, ``Model.predict`` and ``DTree.apply/predict``
(reference samples.py:14-43, model.py:181-214, training.py:73-83) run on crops of a reference
channel pyramid.  Same method and stand-ins as make_golden.py; build container only.

Two NumPy names the reference still uses were removed in NumPy 1.24 (``np.bool`` at
model.py:205): they are aliased to ``bool`` for this run, which is what they were.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    if not hasattr(np, "bool"):
        np.bool = bool
    wb = mg.import_reference()
    from waldboost.samples import gather_samples
    out = {}
    for tag, pb, npz in (("f32", "mixed_d2_T24.pb", "mixed_200x264.npz"),
                         ("u8", "grad_hist_4_u1_d2_T24.pb", "grad_hist_4_u1_200x264.npz")):
        M = wb.Model.load(os.path.join(HERE, pb))
        g = np.load(os.path.join(HERE, npz))
        img, det = g["image"], g["det"]
        levels = list(M.channels(img))
        rng = np.random.default_rng(4)
        for li in (0, 5):
            chns = levels[li][0]
            u, v, _ = chns.shape
            N = 48 if li == 0 else 16
            rs = rng.integers(0, u - 12 + 1, N)
            cs = rng.integers(0, v - 12 + 1, N)
            rs[0], cs[0] = u - 12, v - 12                       # the last valid origin
            d = det[det["level"] == li][: N // 2]               # windows the cascade accepts at this level
            rs[1:1 + d.size], cs[1:1 + d.size] = d["r"], d["c"]
            X = gather_samples(chns, rs, cs, M.shape)
            H, mask = M.predict(X)
            k = f"{tag}/L{li}"
            out[f"{k}/rs"], out[f"{k}/cs"], out[f"{k}/X"] = rs, cs, X
            out[f"{k}/H"], out[f"{k}/mask"] = H, mask
            for t in (0, 3, 10):
                out[f"{k}/apply{t}"] = M.classifier[t].apply(X)
                out[f"{k}/predict{t}"] = M.classifier[t].predict(X)
            print(k, X.shape, X.dtype, "passed", int(mask.sum()), "of", N)
    np.savez_compressed(os.path.join(HERE, "samples_f4.npz"), **out)
    print("f4 golden fixtures written")


if __name__ == "__main__":
    main()
