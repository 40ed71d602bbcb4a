"""CPU restatement of the waldboost detection hot path  --  TEST INFRASTRUCTURE ONLY.

This module is the *oracle* for the MI355X build.  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``; the product package (``waldboost_amd``) never imports it and has
no CPU fallback.

It restates, in plain NumPy (no numba / scikit-image / bbx), what the reference
computes under NumPy 2.x / SciPy 1.15 for

  * ``waldboost.channels.channel_pyramid``      (reference channels.py:111-146)
  * ``waldboost.channels.grad_hist``            (reference channels.py:40-52)
  * ``waldboost.channels.gradients``            (reference channels.py:16-21)
  * ``waldboost.channels.avg_pool_2``           (reference channels.py:55-64)
  * ``waldboost.channels._smooth/smooth_image_3d`` (reference channels.py:78-90)
  * ``waldboost.channels._image_octaves``       (reference channels.py:93-101)
  * ``skimage.transform.resize(order=1, anti_aliasing=False, preserve_range=True)``
    as called at channels.py:132 (third party, absent here: restated as
    scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True) + clip)
  * ``waldboost.channels.grad_mag``             (reference channels.py:11-37)
  * ``waldboost.fpga.grad_hist_4_u1 / grad_mag_u1`` (reference fpga/channels.py:5-67)
  * ``waldboost.model.Model.predict_on_image``  (reference model.py:216-259)
  * ``waldboost.training.DTree.predict_on_image`` (reference training.py:84-96)
  * ``waldboost.model.Model.get_boxes/detect``  (reference model.py:136-179)
  * ``waldboost.samples.gather_samples``        (reference samples.py:14-43)
  * ``waldboost.model.Model.predict`` / ``DTree.apply`` on samples
                                                (reference model.py:181-214, training.py:73-83)

Parity pinning: checked against (a) the reference's own source files imported
from /root/reference with its four missing leaf dependencies stubbed
(tests/golden/make_golden.py -> committed fixtures in tests/golden/), and
(b) SciPy known-answer tests for the resize and gradient steps
(tests/test_oracle.py).  The numba-typing claims (uint8 wrap in the octave
pool, fp64 accumulation in the 3x3 smooth) and scikit-image's resize path are
inferred, not executable here: upstream has no tests for them
("parity unpinned upstream" for those three points).

Semantics S1..S15 refer to SURVEY.md section 3.4.
"""
import math

import numpy as np

__all__ = [
    "level_plan", "octave_shapes", "image_octaves", "resize_bilinear", "gradients",
    "grad_hist", "grad_mag", "grad_hist_4_u1", "grad_mag_u1", "triangle_kernel", "CHANNEL_FUNCS",
    "avg_pool_2", "smooth_image_3d", "channel_pyramid",
    "tree_predict_on_image", "cascade_predict_on_image", "get_boxes", "detect",
    "gather_samples", "tree_apply", "model_predict",
]


# --------------------------------------------------------------------------- S1/S2 plan
def octave_shapes(H, W):
    """Shapes of the octave images (reference channels.py:93-101)."""
    out = []
    h, w = int(H), int(W)
    while not (w < 8 or h < 8):
        out.append((h, w))
        h, w = h // 2, w // 2
    return out


def level_plan(H, W, shrink, n_per_oct):
    """Python-float level plan, exactly as reference channels.py:124-131 (S1).

    Returns a list of dicts: oct (octave index), h, w (octave base size),
    nh, nw (resized size), scale (= real_scale/shrink, channels.py:146).
    shrink=4 is an extension (the reference asserts shrink in [1,2]).
    """
    factor = 2 ** (-1 / n_per_oct)
    levels = []
    for o, (h, w) in enumerate(octave_shapes(H, W)):
        for i in range(n_per_oct):
            s = factor ** i
            nw, nh = int((w * s) / shrink) * shrink, int((h * s) / shrink) * shrink
            real_scale = nw / W
            levels.append(dict(oct=o, h=h, w=w, nh=nh, nw=nw, scale=real_scale / shrink))
    return levels


# --------------------------------------------------------------------------- S2/S8 pooling
def avg_pool_2(arr):
    """2x2 mean, reference channels.py:55-64 under NumPy-2 semantics.

    uint8: the three adds are uint8 ufunc adds (wrap mod 256), '/4' promotes to
    fp64, astype(uint8) truncates  ->  ((a+b+c+d) & 255) >> 2            (S2).
    float32: ((a+b)+c)+d in fp32, exact /4                                (S8).
    """
    u, v = arr.shape[0], arr.shape[1]
    u_lim = u - (u % 2)
    v_lim = v - (v % 2)
    a = arr[0:u_lim:2, 0:v_lim:2, ...]
    b = arr[1:u_lim:2, 0:v_lim:2, ...]
    c = arr[0:u_lim:2, 1:v_lim:2, ...]
    d = arr[1:u_lim:2, 1:v_lim:2, ...]
    if arr.dtype == np.uint8:
        s = (a.astype(np.uint32) + b + c + d) & 255
        return (s >> 2).astype(np.uint8)
    with np.errstate(over="ignore"):
        s = ((a + b) + c) + d                     # stays in arr.dtype (fp32 / fp64)
    return (s / 4).astype(arr.dtype)


def image_octaves(image):
    """reference channels.py:93-101."""
    base = image.copy()
    while True:
        h, w = base.shape[:2]
        if w < 8 or h < 8:
            break
        yield base
        base = avg_pool_2(base)


# --------------------------------------------------------------------------- S3/S4 resize
def _axis_taps(n_in, n_out):
    """scipy NI_ZoomShift coordinates + order-1 spline weights (grid_mode=True)."""
    zoom = np.float64(n_in) / np.float64(n_out)
    k = np.arange(n_out, dtype=np.float64)
    cc = ((k + 0.5) * zoom) - 0.5
    fl = np.floor(cc)
    x = cc - fl
    w0 = 1.0 - x
    w1 = 1.0 - w0
    i0 = fl.astype(np.int64)
    i1 = i0 + 1

    def mirror(i):
        if n_in == 1:
            return np.zeros_like(i)
        p = 2 * (n_in - 1)
        i = np.mod(i, p)
        return np.where(i >= n_in, p - i, i)

    return mirror(i0), mirror(i1), w0, w1


def resize_bilinear(base, nh, nw):
    """skimage.transform.resize(base,(nh,nw),preserve_range=True,order=1,
    anti_aliasing=False).astype(base.dtype) as called at reference channels.py:132.

    uint8 (any non f/d dtype) goes through fp64 and is truncated on the cast
    back (S3, S4); float32 stays float32 (fp64 accumulate, one rounding on store).
    The clip to [min(base), max(base)] is skimage's default clip=True.
    """
    h, w = base.shape
    r0, r1, wr0, wr1 = _axis_taps(h, nh)
    c0, c1, wc0, wc1 = _axis_taps(w, nw)
    v = base.astype(np.float64)
    wr0 = wr0[:, None]; wr1 = wr1[:, None]
    wc0 = wc0[None, :]; wc1 = wc1[None, :]
    t = (v[r0][:, c0] * wr0) * wc0
    t = t + (v[r0][:, c1] * wr0) * wc1
    t = t + (v[r1][:, c0] * wr1) * wc0
    t = t + (v[r1][:, c1] * wr1) * wc1
    if base.dtype == np.float32:
        out = t.astype(np.float32)
        return np.clip(out, base.min(), base.max())
    if base.dtype == np.float64:
        return np.clip(t, base.min(), base.max())
    out = np.clip(t, np.float64(base.min()), np.float64(base.max()))
    return out.astype(base.dtype)               # C truncation toward zero


# --------------------------------------------------------------------------- S5 gradients
def _reflect_pad1(a, axis):
    """scipy.ndimage 'reflect' (edge pixel duplicated), one element each side."""
    first = np.take(a, [0], axis=axis)
    last = np.take(a, [a.shape[axis] - 1], axis=axis)
    return np.concatenate([first, a, last], axis=axis)


def _conv_H(a32, axis):
    """convolve1d(a,[1,2,1],axis): fp64 accumulate 2*x[i] + (x[i-1]+x[i+1]), fp32 store."""
    p = _reflect_pad1(a32.astype(np.float64), axis)
    n = a32.shape[axis]
    lo = np.take(p, range(0, n), axis=axis)
    mid = np.take(p, range(1, n + 1), axis=axis)
    hi = np.take(p, range(2, n + 2), axis=axis)
    return (mid * 2.0 + (lo + hi) * 1.0).astype(np.float32)


def _conv_D(a32, axis):
    """convolve1d(a,[-1,0,1],axis): fp64 x[i]*0 + (x[i-1]-x[i+1])*1, fp32 store."""
    p = _reflect_pad1(a32.astype(np.float64), axis)
    n = a32.shape[axis]
    lo = np.take(p, range(0, n), axis=axis)
    mid = np.take(p, range(1, n + 1), axis=axis)
    hi = np.take(p, range(2, n + 2), axis=axis)
    return (mid * 0.0 + (lo - hi) * 1.0).astype(np.float32)


def gradients(image32):
    """reference channels.py:16-21 (S5)."""
    gy = _conv_D(_conv_H(image32, 1), 0)
    gx = _conv_D(_conv_H(image32, 0), 1)
    return gx, gy


# --------------------------------------------------------------------------- S6/S7
def orientation_table(n_bins=4, full=False):
    """cos/sin constants exactly as reference channels.py:43-46 builds them (fp64)."""
    max_theta = 2 * np.pi if full else np.pi
    theta = np.linspace(0, max_theta, n_bins + 1)
    return np.cos(theta[:-1]), np.sin(theta[:-1])


def grad_hist(image, n_bins=4, full=False, bias=0):
    """reference channels.py:40-52 (S6, S7): fp64 projection, one fp32 rounding."""
    image = image.astype("f")
    gx, gy = gradients(image)
    cs, sn = orientation_table(n_bins, full)
    u, v = gx.shape
    chns = np.empty((u, v, n_bins), np.float32)
    gx64 = gx.astype(np.float64)
    gy64 = gy.astype(np.float64)
    for i, (c, s) in enumerate(zip(cs, sn)):
        chns[..., i] = gx64 * c - gy64 * s
    # (NumPy-2 promotion: a Python scalar or a float32 leaves this float32, a float64 / int64 NumPy scalar makes it float64)
    value = np.fmax(np.abs(chns) - (bias if isinstance(bias, np.generic) else np.float32(bias)), 0)
    return np.sign(chns) * value if full else value


# --------------------------------------------------------------------------- f2: other channel functions
def triangle_kernel(n):
    """reference channels.py:11-13."""
    H = (np.r_[:n + 1, n - 1:-1:-1] + 1).astype("f")
    return H / H.sum()


def _conv_sym(a32, w32, axis):
    """scipy.ndimage.convolve1d(a, w, axis) for an odd symmetric kernel, mode='reflect':
    NI_Correlate1D's symmetric branch -- fp64, tmp = x[l]*w[c]; then for jj = -size1..-1:
    tmp += (x[l+jj] + x[l-jj]) * w[c+jj] -- and one rounding to the array dtype on the store."""
    w = np.asarray(w32, np.float64)
    size1 = w.size // 2
    a = np.moveaxis(a32.astype(np.float64), axis, 0)
    n = a.shape[0]
    idx = np.arange(-size1, n + size1)
    # 'reflect': (d c b a | a b c d | d c b a), valid for any pad length
    period = 2 * n
    idx = np.mod(idx, period)
    idx = np.where(idx >= n, period - 1 - idx, idx)
    p = a[idx]
    tmp = p[size1:size1 + n] * w[size1]
    for jj in range(-size1, 0):
        tmp = tmp + (p[size1 + jj:size1 + jj + n] + p[size1 - jj:size1 - jj + n]) * w[size1 + jj]
    return np.moveaxis(tmp, 0, axis).astype(a32.dtype)


def grad_mag(image, norm=5, eps=1e-3):
    """reference channels.py:30-37: fp32 gradient magnitude, divided by its triangle-filtered
    self (+eps); one channel."""
    gx, gy = gradients(image.astype("f"))
    mag = np.sqrt(gx ** 2 + gy ** 2)
    if norm is not None and norm > 1:
        H = triangle_kernel(norm)
        nrm = _conv_sym(_conv_sym(mag, H, 0), H, 1)
        # (`mag /= norm + eps`: with a float64 NumPy eps the quotient is float64, cast back into the float32 array)
        mag = (mag / (nrm + (eps if isinstance(eps, np.generic) else np.float32(eps)))).astype(np.float32)
    return mag[..., None]


def _sobel_int(arr):
    """reference fpga/channels.py:5-27: the two 3x3 numba stencils.  Numba promotes the uint8
    elements to int64 for scalar arithmetic (no wrap) and leaves the 1-pixel output border 0."""
    a = arr.astype(np.int64)
    u, v = a.shape
    dx = np.zeros((u, v), np.int64)
    dy = np.zeros((u, v), np.int64)
    if u >= 3 and v >= 3:
        def sh(dr, dc):
            return a[1 + dr:u - 1 + dr, 1 + dc:v - 1 + dc]
        dx[1:-1, 1:-1] = -(sh(-1, -1) + 2 * sh(0, -1) + sh(1, -1)) + sh(-1, 1) + 2 * sh(0, 1) + sh(1, 1)
        dy[1:-1, 1:-1] = -(sh(-1, -1) + 2 * sh(-1, 0) + sh(-1, 1)) + sh(1, -1) + 2 * sh(1, 0) + sh(1, 1)
    return dx, dy


def grad_hist_4_u1(arr):
    """reference fpga/channels.py:29-53: 4 integer orientation channels, uint8.
    y1/y3 are fp64 halves stored into int32 (truncation toward zero); |y| // 4 clamped to 255."""
    dx, dy = _sobel_int(arr)
    y = np.empty(arr.shape + (4,), np.int32)
    y[..., 0] = dx
    y[..., 1] = np.trunc(0.5 * dx - 0.5 * dy)
    y[..., 2] = dy
    y[..., 3] = np.trunc(0.5 * dx + 0.5 * dy)
    return np.fmin(np.abs(y) // 4, 255).astype(np.uint8)


def grad_mag_u1(arr):
    """reference fpga/channels.py:56-67: max(|dx|, |dy|) // 4 clamped to 255, one uint8 channel."""
    dx, dy = _sobel_int(arr)
    y = np.maximum(np.abs(dx), np.abs(dy)).astype(np.int32)[..., None]
    return np.fmin(y // 4, 255).astype(np.uint8)


CHANNEL_FUNCS = {"grad_hist": grad_hist, "grad_mag": grad_mag, "grad_hist_4_u1": grad_hist_4_u1,
                 "grad_mag_u1": grad_mag_u1}


# --------------------------------------------------------------------------- S9 smooth
def smooth_image_3d(arr):
    """reference channels.py:78-90 with numba stencil semantics (S9):
    fp64 nine-term sum in source order, /16, one fp32 rounding; 1-px border = 0."""
    out = np.zeros_like(arr)
    u, v = arr.shape[:2]
    if u < 3 or v < 3:
        return out
    a = arr.astype(np.float64)

    def sh(dr, dc):
        return a[1 + dr:u - 1 + dr, 1 + dc:v - 1 + dc, ...]

    acc = sh(-1, -1) + 2 * sh(-1, 0)
    acc = acc + sh(-1, 1)
    acc = acc + 2 * sh(0, -1)
    acc = acc + 4 * sh(0, 0)
    acc = acc + 2 * sh(0, 1)
    acc = acc + sh(1, -1)
    acc = acc + 2 * sh(1, 0)
    acc = acc + sh(1, 1)
    out[1:u - 1, 1:v - 1, ...] = (acc / 16).astype(arr.dtype)
    return out


# --------------------------------------------------------------------------- a1 pyramid
def channel_pyramid(image, channel_opts):
    """reference channels.py:111-146; yields (chns[u,v,C] float32, scale)."""
    if not isinstance(image, np.ndarray):
        raise TypeError("Image must be numpy array")
    if image.ndim != 2:
        raise ValueError("Image must have 2 dimensions")
    shrink = channel_opts["shrink"]
    n_per_oct = channel_opts["n_per_oct"]
    smooth = channel_opts["smooth"]
    channels = channel_opts.get("channels", grad_hist)
    if isinstance(channels, str):
        channels = CHANNEL_FUNCS[channels]
    assert shrink in [1, 2, 4], "shrink must be 1, 2 (reference) or 4 (extension)"
    factor = 2 ** (-1 / n_per_oct)
    for base in image_octaves(image):
        h, w = base.shape[:2]
        for i in range(n_per_oct):
            s = factor ** i
            nw, nh = int((w * s) / shrink) * shrink, int((h * s) / shrink) * shrink
            real_scale = nw / image.shape[1]
            im = resize_bilinear(base, nh, nw)
            chns = channels(im)
            if shrink >= 2:
                chns = avg_pool_2(chns)
            if shrink == 4:                      # extension, not reference behaviour
                chns = avg_pool_2(chns)
            if smooth == 1:
                chns = smooth_image_3d(chns)
            yield np.atleast_3d(chns), real_scale / shrink


# --------------------------------------------------------------------------- a9 tree
def tree_predict_on_image(tree, X, rs, cs):
    """reference training.py:84-96 (S13).  ``tree`` is a dict with the arrays the
    reference DTree.__init__ builds: feature u8[n,3], threshold f32[n], left i8[n],
    right i8[n], prediction f32[n]."""
    feature, threshold = tree["feature"], tree["threshold"]
    left, right, prediction = tree["left"], tree["right"], tree["prediction"]
    node = np.zeros(rs.size, "i")
    idx_in_node = {0: np.arange(rs.size)}
    for n in np.flatnonzero(left >= 0):
        r, c, ch = (int(x) for x in feature[n])
        lnode, rnode = int(left[n]), int(right[n])
        idx = idx_in_node[int(n)]
        b = X[rs[idx] + r, cs[idx] + c, ch] <= threshold[n]
        node[idx] = np.where(b, lnode, rnode)
        idx_in_node[lnode] = idx[b]
        idx_in_node[rnode] = idx[~b]
    return prediction[node]


def make_tree(feature, threshold, left, right, prediction):
    """Array normalisation of reference DTree.__init__ (training.py:24-31)."""
    return dict(
        feature=np.array([f if f is not None else [0, 0, 0] for f in feature], np.uint8).reshape(-1, 3),
        threshold=np.array(threshold, np.float32),
        left=np.array(left, np.int8),
        right=np.array(right, np.int8),
        prediction=np.array(prediction, np.float32),
    )


# --------------------------------------------------------------------------- a8 cascade
def cascade_predict_on_image(shape, trees, thetas, X):
    """reference model.py:216-259 (S11, S12).

    Returns (rs, cs, hs, alive) where alive[t] = number of windows entering
    stage t (n_weak contribution of the level = alive.sum(), n_loc = alive[0]
    when the model is non-empty else the window count)."""
    u, v, ch_image = X.shape
    m, n, ch_cls = shape
    assert ch_image == ch_cls, f"Invalid number of channels. Expected {ch_cls} given {ch_image}."
    rs, cs = np.indices((max(u - m, 0), max(v - n, 0)))
    rs = rs.flatten()
    cs = cs.flatten()
    hs = np.zeros_like(rs, np.float32)
    alive = np.zeros(len(trees), np.int64)
    for t, (tree, theta) in enumerate(zip(trees, thetas)):
        if not rs.size:
            break
        hs += tree_predict_on_image(tree, X, rs, cs)
        alive[t] = hs.size
        if theta == -np.inf:
            continue
        mask = hs >= theta
        rs, cs, hs = rs[mask], cs[mask], hs[mask]
    return rs, cs, hs, alive


def get_boxes(shape, r, c, scale):
    """reference model.py:136-147 (S14): XYXY float32 rects times 1/scale."""
    if r.size == 0:
        return np.empty((0, 4), "f")
    m, n = shape[:2]
    x1 = c.reshape(-1, 1)
    y1 = r.reshape(-1, 1)
    rects = np.concatenate([x1, y1, x1 + n, y1 + m], axis=1).astype(np.float32)
    return (rects * np.float32(1.0 / scale)).astype(np.float32)


def detect(shape, channel_opts, trees, thetas, image):
    """reference model.py:149-179.  Returns dict(boxes, scores, level, r, c,
    alive[L,T], n_loc, n_weak, scales)."""
    boxes, scores, lev, rr, cc, alive, scales = [], [], [], [], [], [], []
    n_loc = 0
    m, n, _ = shape
    for li, (chns, scale) in enumerate(channel_pyramid(image, channel_opts)):
        r, c, h, a = cascade_predict_on_image(shape, trees, thetas, chns)
        n_loc += max(chns.shape[0] - m, 0) * max(chns.shape[1] - n, 0)
        boxes.append(get_boxes(shape, r, c, scale))
        scores.append(h)
        lev.append(np.full(r.size, li, np.int32))
        rr.append(r); cc.append(c); alive.append(a); scales.append(scale)
    alive = np.stack(alive) if alive else np.zeros((0, len(trees)), np.int64)
    return dict(
        boxes=np.concatenate(boxes) if boxes else np.empty((0, 4), "f"),
        scores=np.concatenate(scores) if scores else np.empty(0, "f"),
        level=np.concatenate(lev) if lev else np.empty(0, np.int32),
        r=np.concatenate(rr) if rr else np.empty(0, np.int64),
        c=np.concatenate(cc) if cc else np.empty(0, np.int64),
        alive=alive, n_loc=int(n_loc), n_weak=int(alive.sum()), scales=scales,
    )


# --------------------------------------------------------------------------- f4: training-time callers
def gather_samples(chns, rs, cs, shape):
    """reference samples.py:14-43."""
    if rs.size != cs.size:
        raise ValueError("Sizes of 'rs' and 'cs' must match")
    m, n, _ = shape
    if rs.size == 0:
        return np.empty((0,) + tuple(shape), dtype=chns.dtype)
    return np.array([chns[r:r + m, c:c + n, ...] for r, c in zip(rs, cs)])


def tree_apply(tree, X):
    """reference training.py:73-81: leaf index per sample X[i] of shape (m, n, C)."""
    node = np.zeros(X.shape[0], "i")
    for n in np.flatnonzero(tree["left"] >= 0):
        r, c, ch = (int(x) for x in tree["feature"][n])
        idx = np.flatnonzero(node == n)
        b = X[idx, r, c, ch] <= tree["threshold"][n]
        node[idx] = np.where(b, tree["left"][n], tree["right"][n])
    return node


def model_predict(shape, trees, thetas, X):
    """reference model.py:181-214: (H, mask) of the cascade on samples X[N, m, n, C]."""
    n = X.shape[0]
    assert tuple(X.shape[1:]) == tuple(shape)
    H = np.zeros(n, np.float32)
    mask = np.ones(n, bool)
    for tree, theta in zip(trees, thetas):
        H[mask] += tree["prediction"][tree_apply(tree, X[mask, ...])]
        if theta == -np.inf:
            continue
        mask = np.logical_and(mask, H >= theta)
    H[~mask] = -np.inf
    return H, mask
