"""Synthetic inputs for tests and bench (NumPy only, no device code).

Image recipe follows SURVEY.md section 8(d): a smooth sinusoidal background
plus gaussian noise plus 0-3 bright squares in the style of the reference's
``utils.fake_data_generator`` (reference utils.py:81-97).  There is no network
for datasets, so every measured number in this repo is on these images.
"""
import numpy as np


def synth_image(H, W, seed=0, dtype=np.uint8):
    rng = np.random.default_rng(seed)
    # (127 + 60 sin(x / 17) cos(y / 23)) + noise, every operation in the order of the recipe; the background is an outer
    # product of a row and a column -- the same float64 values as on the full coordinate grid, at a third of the time
    xs, ys = np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64)
    img = (60.0 * np.sin(xs / 17.0))[None, :] * np.cos(ys / 23.0)[:, None]
    img = 127.0 + img
    img += rng.normal(0.0, 20.0, (H, W))
    for _ in range(int(rng.integers(0, 4))):
        w = int(rng.integers(max(min(H, W) // 10, 2), max(min(H, W) // 4, 4)))
        px = int(rng.integers(0, max(W - w, 1)))
        py = int(rng.integers(0, max(H - w, 1)))
        img[py:py + w, px:px + w] += rng.uniform(50.0, 128.0)
    np.clip(img, 0, 255, out=img)
    if np.dtype(dtype) == np.uint8:
        return img.astype(np.uint8)
    return (img / 255.0).astype(dtype)


def synth_batch(B, H, W, seed0=0, dtype=np.uint8):
    return np.stack([synth_image(H, W, seed0 + b, dtype) for b in range(B)])


def random_tree_arrays(rng, shape, depth, thr_lo, thr_hi, unbalanced=False):
    """Arrays of one decision tree in the reference DTree layout
    (reference training.py:24-31): BFS node order, ``left/right = -1`` on leaves."""
    m, n, C = shape
    if depth == 1:
        left, right = [1, -1, -1], [2, -1, -1]
    elif depth == 2 and unbalanced:          # root -> (leaf, split) : 5 nodes
        left, right = [1, -1, 3, -1, -1], [2, -1, 4, -1, -1]
    elif depth == 2:
        left, right = [1, 3, 5, -1, -1, -1, -1], [2, 4, 6, -1, -1, -1, -1]
    elif depth == 3:
        left = [1, 3, 5, 7, 9, 11, 13] + [-1] * 8
        right = [2, 4, 6, 8, 10, 12, 14] + [-1] * 8
    else:
        raise ValueError("depth must be 1, 2 or 3")
    k = len(left)
    feature = np.stack([rng.integers(0, m, k), rng.integers(0, n, k), rng.integers(0, C, k)], 1).astype(np.uint8)
    threshold = rng.uniform(thr_lo, thr_hi, k).astype(np.float32)
    pred = (rng.uniform(0.2, 1.0, k) * rng.choice([-1.0, 1.0], k)).astype(np.float32)
    left = np.array(left, np.int8)
    right = np.array(right, np.int8)
    leaf = left < 0
    feature[leaf] = 0                         # reference writes (0,0,0) on leaves (training.py:25)
    return feature, threshold, left, right, pred
