#!/usr/bin/env python3
"""Golden fixtures for the other channel functions the reference ships:
``waldboost.fpga.grad_hist_4_u1`` / ``grad_mag_u1`` (reference fpga/channels.py:29-67) and
``waldboost.channels.grad_mag`` (reference channels.py:30-37), run through the reference's own
``channel_pyramid`` / ``Model.predict_on_image`` / ``Model.detect`` source.

Same method, stand-ins and caveats as make_golden.py (imported from there); run in the build
container only: ``python tests/golden/make_golden_f2.py``.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    wb = mg.import_reference()
    from waldboost_amd.synth import synth_image, random_tree_arrays
    from waldboost.channels import channel_pyramid, grad_mag
    from waldboost.fpga import grad_hist_4_u1, grad_mag_u1

    funcs = {"grad_hist_4_u1": grad_hist_4_u1, "grad_mag_u1": grad_mag_u1, "grad_mag": grad_mag}
    meta = {"numpy": np.__version__, "scipy": __import__("scipy").__version__, "cases": {},
            "names": {k: wb.model.symbol_name(f) for k, f in funcs.items()}}

    def edgy(H, W, seed):
        """High-contrast blocks: strong gradients so the uint8 channel sums wrap in avg_pool_2."""
        rng = np.random.default_rng(seed)
        img = synth_image(H, W, seed).astype(np.int32)
        blk = rng.integers(0, 2, (H // 4 + 1, W // 4 + 1)) * 255
        big = np.kron(blk, np.ones((4, 4), np.int64))[:H, :W]
        return np.where(rng.random((H, W)) < 0.5, big, img).astype(np.uint8)

    # (i) small pyramids, every level
    small = {}
    cases = [
        ("gh4u1_u8_97x131", synth_image(97, 131, 2), "grad_hist_4_u1", 2, 8, 1),
        ("gh4u1_u8_edgy_80x104", edgy(80, 104, 3), "grad_hist_4_u1", 2, 4, 1),
        ("gh4u1_u8_61x75_s1_nosmooth", synth_image(61, 75, 5), "grad_hist_4_u1", 1, 3, 0),
        ("gh4u1_u8_edgy_64x96_s2_nosmooth", edgy(64, 96, 6), "grad_hist_4_u1", 2, 2, 0),
        ("gmu1_u8_97x131", synth_image(97, 131, 2), "grad_mag_u1", 2, 8, 1),
        ("gmu1_u8_edgy_72x88_s1", edgy(72, 88, 7), "grad_mag_u1", 1, 2, 1),
        ("gmu1_u8_edgy_64x96_s2_nosmooth", edgy(64, 96, 8), "grad_mag_u1", 2, 2, 0),
        ("gm_u8_97x131", synth_image(97, 131, 2), "grad_mag", 2, 8, 1),
        ("gm_f32_64x96_s1", synth_image(64, 96, 7, np.float32), "grad_mag", 1, 2, 1),
        ("gm_u8_80x120_s2_nosmooth", synth_image(80, 120, 6), "grad_mag", 2, 4, 0),
    ]
    for name, img, fn, shrink, npo, smooth in cases:
        o = dict(shrink=shrink, n_per_oct=npo, smooth=smooth, channels=funcs[fn])
        small[f"{name}/image"] = img
        lv = list(channel_pyramid(img, o))
        meta["cases"][name] = dict(channels=fn, shrink=shrink, n_per_oct=npo, smooth=smooth, n_levels=len(lv),
                                   scales=[float(s) for _, s in lv], shapes=[list(c.shape) for c, _ in lv],
                                   dtype=str(lv[0][0].dtype))
        for i, (c, s) in enumerate(lv):
            small[f"{name}/L{i}"] = c
    np.savez_compressed(os.path.join(HERE, "pyramids_f2.npz"), **small)

    # (ii) detection with each channel function: 24-stage cascades on a 200x264 image
    img = synth_image(200, 264, 11)
    for fn, C, lo, hi, seed in (("grad_hist_4_u1", 4, 1.0, 24.0, 5), ("grad_mag_u1", 1, 2.0, 40.0, 6),
                                ("grad_mag", 1, 0.3, 1.8, 7)):
        rng = np.random.default_rng(seed)
        shape = (12, 12, C)
        o = dict(shrink=2, n_per_oct=8, smooth=1, channels=funcs[fn])
        T = 24
        trees = [random_tree_arrays(rng, shape, 2 if t % 6 else 1, lo, hi, unbalanced=(t % 5 == 3 and t % 6 != 0))
                 for t in range(T)]
        surv = [max(0.75 ** (t + 1), 5e-3) if t % 4 != 2 else None for t in range(T)]
        th = mg.calibrate_thetas(wb, shape, o, trees, img, surv)
        M = mg.build_ref_model(wb, shape, o, trees, th)
        M.save(os.path.join(HERE, f"{fn}_d2_T24.pb"))
        det, alive, scales, hashes, n_loc, n_weak = mg.ref_scan(wb, M, img)
        np.savez_compressed(os.path.join(HERE, f"{fn}_200x264.npz"), image=img, det=det, alive=alive,
                            scales=np.array(scales), n_loc=n_loc, n_weak=n_weak)
        # the stored function name resolves back through the reference's loader
        L = wb.Model.load(os.path.join(HERE, f"{fn}_d2_T24.pb"))
        assert L.channel_opts["channels"] is funcs[fn]
        meta[fn] = dict(n_loc=int(n_loc), n_weak=int(n_weak), n_det=int(det.size), chn_sha256=[h[1] for h in hashes])
        print(fn, "n_loc", n_loc, "n_weak", n_weak, "detections", det.size)

    with open(os.path.join(HERE, "golden_meta_f2.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("f2 golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
