"""Channel features / pyramid -- drop-in for ``waldboost.channels`` on the hot path.

``channel_pyramid(image, channel_opts)`` keeps the reference signature and yields the same
``(chns[u,v,C] float32, scale)`` pairs (reference channels.py:111-146), computed by the fused
HIP kernel in csrc/wb_channels.hip; ``grad_hist(image)`` is reference channels.py:40-52 for
its default arguments (n_bins=4, full=False, bias=0).
"""
import numpy as np

from . import _native as nat
from . import engine as _engine


def _validate_image(image):
    # reference channels.py:104-108
    if not isinstance(image, np.ndarray):
        raise TypeError("Image must be numpy array")
    if image.ndim != 2:
        raise ValueError("Image must have 2 dimensions")


def grad_hist(image, n_bins=4, full=False, bias=0):
    """4 unsigned oriented-gradient channels of a 2-D image -> float32 [H,W,4]."""
    if n_bins != 4 or full or bias != 0:
        raise NotImplementedError("the HIP grad_hist kernel implements the defaults n_bins=4, full=False, bias=0")
    _validate_image(image)
    img = np.ascontiguousarray(image.astype("f"))
    H, W = img.shape
    if H < 1 or W < 1:
        return np.empty((H, W, 4), np.float32)
    eng = _engine.get_engine(H, W, np.float32, 1, 1, 0, 1, exact_single=True)
    eng.load_images(img)
    eng.run_channels()
    return eng.read_level(0, 0)


# Names a stored model may use for its channel function (reference model.py:302 writes
# module.qualname; model.py:27-29 evals it on load -- replaced here by this allow-list).
CHANNEL_FUNCS = {
    "waldboost.channels.grad_hist": grad_hist,
    "waldboost_amd.channels.grad_hist": grad_hist,
}


def is_grad_hist(func):
    if func is grad_hist:
        return True
    name = getattr(func, "__module__", "") + "." + getattr(func, "__qualname__", "")
    return name == "waldboost.channels.grad_hist"     # the reference's own function object


def read_opts(channel_opts):
    shrink = channel_opts["shrink"]
    n_per_oct = channel_opts["n_per_oct"]
    smooth = channel_opts["smooth"]
    channels = channel_opts["channels"]
    assert shrink in [1, 2, 4], "Shrink factor must be integer 1 <= shrink <= 2 (4: extension of this build)"
    if not is_grad_hist(channels):
        raise NotImplementedError(f"channel function {channels!r} has no HIP kernel (grad_hist only)")
    if smooth not in (0, 1):
        smooth = 0          # the reference smooths only when smooth == 1 (channels.py:141)
    return int(shrink), int(n_per_oct), int(smooth)


def channel_pyramid(image, channel_opts):
    """Generate the channel pyramid of `image` (lazy generator, like the reference).

    The whole pyramid is computed on the GPU in one launch group when the first level is
    requested; levels are copied to the host one by one as they are consumed."""
    _validate_image(image)
    shrink, n_per_oct, smooth = read_opts(channel_opts)
    H, W = image.shape
    eng = _engine.get_engine(H, W, image.dtype, shrink, n_per_oct, smooth, 1)
    if eng.plan.n_levels == 0:
        return
    eng.load_images(image)
    eng.run_channels()
    for l in range(eng.plan.n_levels):
        yield np.atleast_3d(eng.read_level(0, l)), eng.plan.scales[l]
