#!/usr/bin/env python3
"""Build the committed benchmark cascades (tests/golden/models/*.pb).

Random-weight cascades of the shape BASELINE.json names, with rejection thresholds calibrated
ONCE on synth_image(seed=0) with the CPU oracle so that the fraction of windows alive after
stage t follows a fixed front-loaded schedule.  Needs no GPU and no
reference; the outputs are committed so every run measures the same model.

    cfg2_d2_T128.pb : window (12,12,4), 128 stages, depth 2, 1080p, survival ~1e-3, eval cost ~5
    cfg5_d2_T256.pb : window (12,12,4), 256 stages, depth 2, 4K shrink=4 n_per_oct=12 (extension), ~1e-4
    cfg2_gh4u1_d2_T128.pb : cfg2 with the integer channels fpga.grad_hist_4_u1 (uint8 x4)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import waldboost_amd as wb                      # host-side container + .pb writer only
from oracle import wb_oracle as orc
from waldboost_amd.synth import synth_image, random_tree_arrays


def build(name, H, W, opts, T, depth, floor, decay, seed, chan="grad_hist"):
    rng = np.random.default_rng(seed)
    shape = (12, 12, 4)
    img = synth_image(H, W, 0)
    levels = list(orc.channel_pyramid(img, dict(opts, channels=chan)))
    shape = (12, 12, levels[0][0].shape[2])
    pool = np.concatenate([c[::3, ::3].reshape(-1) for c, _ in levels[:8]]).astype(np.float64)
    q30, q70 = np.quantile(pool, [0.3, 0.7])
    if q70 - q30 < 1.0:                          # integer channels: keep the thresholds spread over a few values
        q70 = q30 + 4.0
    m, n, _ = shape
    state, total = [], 0
    for chns, _ in levels:
        u, v, _ = chns.shape
        rs, cs = np.indices((max(u - m, 0), max(v - n, 0)))
        rs, cs = rs.flatten(), cs.flatten()
        state.append([rs, cs, np.zeros(rs.size, np.float32)])
        total += rs.size
    M = wb.Model(shape, dict(opts, channels=wb.channels.SPECS[chan].func))
    alive_frac = 1.0
    for t in range(T):
        f, th, l, r, p = random_tree_arrays(rng, shape, depth, q30, q70)
        tree = orc.make_tree(f, th, l, r, p)
        for (chns, _), st in zip(levels, state):
            if st[0].size:
                st[2] = st[2] + orc.tree_predict_on_image(tree, chns, st[0], st[1])
        target = max(decay ** (t + 1), floor)
        theta = float("-inf")
        if t % 8 != 7 and target < alive_frac:
            pooled = np.concatenate([st[2] for st in state])
            keep = max(int(round(target * total)), 1)
            if keep < pooled.size:
                theta = float(np.float32(np.partition(pooled, pooled.size - keep)[pooled.size - keep]))
                for st in state:
                    mk = st[2] >= np.float32(theta)
                    st[0], st[1], st[2] = st[0][mk], st[1][mk], st[2][mk]
                alive_frac = sum(st[0].size for st in state) / total
        M.append(wb.DTree(f, th, l, r, p), theta)
    path = os.path.join(HERE, "models", name)
    M.save(path)
    print(name, "windows", total, "final survivors", sum(st[0].size for st in state), "->", path)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg2", "cfg5"]
    if "cfg2" in which:
        build("cfg2_d2_T128.pb", 1080, 1920, dict(shrink=2, n_per_oct=8, smooth=1), 128, 2, 1e-3, 0.8, 2024)
    if "gh4u1" in which:
        build("cfg2_gh4u1_d2_T128.pb", 1080, 1920, dict(shrink=2, n_per_oct=8, smooth=1), 128, 2, 1e-3, 0.8, 2026,
              chan="grad_hist_4_u1")
    if "cfg5" in which:
        build("cfg5_d2_T256.pb", 2160, 3840, dict(shrink=4, n_per_oct=12, smooth=1), 256, 2, 1e-4, 0.8, 2025)
