"""Channel features / pyramid -- drop-in for ``waldboost.channels`` on the hot path.

``channel_pyramid(image, channel_opts)`` keeps the reference signature and yields the same
``(chns[u,v,C], scale)`` pairs (reference channels.py:111-146), computed by the fused HIP
kernels in csrc/wb_channels.hip.  Channel functions with a kernel:

  ``grad_hist(image, n_bins, full, bias)``   reference channels.py:40-52        -> float32 [H,W,n_bins]
  ``grad_mag(image, norm, eps)``             reference channels.py:30-37        -> float32 [H,W,1]
                                  (inside a pyramid they run with their default arguments, as the reference's
                                  ``channels(im)`` call does, in the fused kernels; called directly with other
                                  arguments, in the plain kernels of csrc/wb_chanfunc.hip)
  ``fpga.grad_hist_4_u1(image)``  reference fpga/channels.py:29-53        -> uint8   [H,W,4]
  ``fpga.grad_mag_u1(image)``     reference fpga/channels.py:56-67        -> uint8   [H,W,1]
"""
import ctypes as C

import numpy as np

from . import _native as nat
from . import engine as _engine
from .chanfunc import SPECS, ChannelSpec  # noqa: F401


def _validate_image(image, allow_tensor=False):
    # reference channels.py:104-108
    # allow_tensor (Model.detect / detect_stream only -- an extension: the reference takes host ndarrays): a 2-D torch
    # tensor, on the host (page-locked: uploaded asynchronously) or already on the GPU
    if not isinstance(image, np.ndarray):
        if not (allow_tensor and type(image).__module__.startswith("torch") and hasattr(image, "data_ptr")):
            raise TypeError("Image must be numpy array")
    if image.ndim != 2:
        raise ValueError("Image must have 2 dimensions")


def _on_bare_image(image, spec):
    """A channel function applied to one image (no pyramid, no resize): a single-level engine."""
    _validate_image(image)
    H, W = image.shape
    if H < 1 or W < 1:
        return np.empty((H, W, spec.n_channels), spec.dtype)
    img = np.ascontiguousarray(image)
    eng = _engine.get_engine(H, W, img.dtype, 1, 1, 0, 1, exact_single=True, channels=spec)
    eng.load_images(img)
    eng.run_channels()
    return eng.read_level(0, 0)


def _scalar_arg(x, what):
    """(value, wide) of a scalar the reference combines with a float32 array (`array - bias`, `norm + eps`) under NumPy-2
    promotion: Python int / float / bool and float32 / float16 / small-integer NumPy scalars are weak -- the value is
    rounded to float32, the arithmetic stays float32 (wide False); a float64 / int64 NumPy scalar makes it float64."""
    if type(x) in (bool, int, float) or np.result_type(np.float32, x) == np.float32:     # (np.float64 subclasses float)
        return float(np.float32(x)), False
    if np.result_type(np.float32, x) == np.float64 and np.ndim(x) == 0:
        return float(np.float64(x)), True
    raise NotImplementedError(f"{what}={x!r} ({type(x).__name__}) has no kernel: pass a real scalar")


def grad_hist(image, n_bins=4, full=False, bias=0):
    """Oriented-gradient channels of a 2-D image -> float32 [H,W,n_bins] (reference channels.py:40-52): n_bins
    orientations over pi (full=False: magnitudes) or 2*pi (full=True: signed), less `bias`, clamped at 0."""
    import torch
    _validate_image(image)
    if n_bins == 4 and not full and type(bias) in (bool, int, float) and bias == 0:
        return _on_bare_image(image.astype("f"), SPECS["grad_hist"])
    n_bins = int(n_bins)
    if not 1 <= n_bins <= 32:
        raise NotImplementedError(f"grad_hist: n_bins={n_bins} has no kernel (1..32)")
    bias_v, wide = _scalar_arg(bias, "bias")
    img = np.ascontiguousarray(image.astype("f"))
    H, W = img.shape
    if H < 1 or W < 1:
        return np.empty((H, W, n_bins), np.float64 if wide else np.float32)
    dev = nat.require_gpu()
    theta = np.linspace(0, 2 * np.pi if full else np.pi, n_bins + 1)          # reference channels.py:43-46
    cs_sn = np.concatenate([np.cos(theta[:-1]), np.sin(theta[:-1])]).astype(np.float64)
    d = torch.from_numpy(img).to(dev)
    out = torch.empty((H, W, n_bins), dtype=torch.float64 if wide else torch.float32, device=dev)
    nat.check(nat.load().wb_grad_hist_launch(nat.stream_ptr(), nat.ptr(d), H, W, n_bins, int(bool(full)), bias_v, int(wide),
                                             cs_sn.ctypes.data_as(C.POINTER(C.c_double)), nat.ptr(out)), "wb_grad_hist_launch")
    return out.cpu().numpy()


def triangle_kernel(n):
    """reference channels.py:11-13."""
    H = (np.r_[:n + 1, n - 1:-1:-1] + 1).astype("f")
    return H / H.sum()


def grad_mag(image, norm=5, eps=1e-3):
    """Gradient magnitude, divided by its triangle-filtered self + eps when norm > 1 -> float32 [H,W,1]
    (reference channels.py:30-37)."""
    import torch
    _validate_image(image)
    if norm == 5 and type(eps) is float and eps == 1e-3:
        return _on_bare_image(image.astype("f"), SPECS["grad_mag"])
    img = np.ascontiguousarray(image.astype("f"))
    H, W = img.shape
    if H < 1 or W < 1:
        return np.empty((H, W, 1), np.float32)
    taps = None
    if norm is not None and norm > 1:
        taps = np.ascontiguousarray(triangle_kernel(int(norm)), np.float32)
        if taps.size > 127:
            raise NotImplementedError(f"grad_mag: norm={norm} has no kernel (up to 63)")
        eps_v, wide = _scalar_arg(eps, "eps")
    dev = nat.require_gpu()
    d = torch.from_numpy(img).to(dev)
    out = torch.empty((H, W), dtype=torch.float32, device=dev)
    scratch = torch.empty((2, H, W), dtype=torch.float32, device=dev) if taps is not None else None
    nat.check(nat.load().wb_grad_mag_launch(nat.stream_ptr(), nat.ptr(d), H, W, 0 if taps is None else int(taps.size),
                                            None if taps is None else taps.ctypes.data_as(C.POINTER(C.c_float)),
                                            0.0 if taps is None else eps_v, 0 if taps is None else int(wide), nat.ptr(scratch),
                                            nat.ptr(out)), "wb_grad_mag_launch")
    return out.cpu().numpy()[..., None]


def grad_hist_4_u1(image):
    """8 bit image -> 4 integer orientation channels, uint8 [H,W,4] (reference fpga/channels.py:29-53)."""
    _validate_image(image)
    _require_u8(image, "grad_hist_4_u1")
    return _on_bare_image(image, SPECS["grad_hist_4_u1"])


def grad_mag_u1(image):
    """8 bit image -> max(|dx|, |dy|) // 4, uint8 [H,W,1] (reference fpga/channels.py:56-67)."""
    _validate_image(image)
    _require_u8(image, "grad_mag_u1")
    return _on_bare_image(image, SPECS["grad_mag_u1"])


def _require_u8(image, name):
    if image.dtype != np.uint8:
        raise NotImplementedError(f"{name} takes 8 bit images (uint8), got {image.dtype}; "
                                  "other dtypes have no HIP kernel")


SPECS["grad_hist"].func = grad_hist
SPECS["grad_hist_4_u1"].func = grad_hist_4_u1
SPECS["grad_mag_u1"].func = grad_mag_u1
SPECS["grad_mag"].func = grad_mag

# Names a stored model may use for its channel function (reference model.py:302 writes
# module.qualname; model.py:27-29 evals it on load -- replaced here by this allow-list).
CHANNEL_FUNCS = {}
for _spec in SPECS.values():
    CHANNEL_FUNCS[_spec.reference_name] = _spec.func
    CHANNEL_FUNCS["waldboost_amd.channels." + _spec.key] = _spec.func
CHANNEL_FUNCS["waldboost.fpga.grad_hist_4_u1"] = grad_hist_4_u1
CHANNEL_FUNCS["waldboost.fpga.grad_mag_u1"] = grad_mag_u1
CHANNEL_FUNCS["waldboost_amd.fpga.grad_hist_4_u1"] = grad_hist_4_u1
CHANNEL_FUNCS["waldboost_amd.fpga.grad_mag_u1"] = grad_mag_u1


def channel_spec(func):
    """The ChannelSpec of a channel function: ours, or the reference's own function object
    (recognised by module and qualified name); None if it has no kernel."""
    for spec in SPECS.values():
        if func is spec.func:
            return spec
    name = getattr(func, "__module__", "") + "." + getattr(func, "__qualname__", "")
    for spec in SPECS.values():
        if name == spec.reference_name:
            return spec
    return None


def is_grad_hist(func):
    return channel_spec(func) is SPECS["grad_hist"]


def read_opts(channel_opts, allow_callable=False):
    """(shrink, n_per_oct, smooth, spec) of a channel_opts dict (reference channels.py:116-120).  allow_callable: a
    channel function without a kernel gives spec None (the caller then runs the function itself between the GPU steps)
    instead of NotImplementedError."""
    shrink = channel_opts["shrink"]
    n_per_oct = channel_opts["n_per_oct"]
    smooth = channel_opts["smooth"]
    channels = channel_opts["channels"]
    assert shrink in [1, 2, 4], "Shrink factor must be integer 1 <= shrink <= 2 (4: extension of this build)"
    spec = channel_spec(channels)
    if spec is None and not (allow_callable and callable(channels)):
        raise NotImplementedError(f"channel function {channels!r} has no HIP kernel (known: {sorted(SPECS)})")
    if smooth not in (0, 1):
        smooth = 0          # the reference smooths only when smooth == 1 (channels.py:141)
    return int(shrink), int(n_per_oct), int(smooth), spec


def channel_pyramid(image, channel_opts):
    """Generate the channel pyramid of `image` -- a lazy generator, like the reference (channels.py:125-146):
    the octaves are built when the first level is requested, and each level's channels are computed on the GPU
    (one launch over that level's tiles) and copied to the host only when the generator is advanced to it, so a
    caller that stops early pays for the levels it consumed.  (Model.detect does not go through this generator:
    it computes the whole pyramid in one launch and scans it on the GPU.)"""
    _validate_image(image)
    shrink, n_per_oct, smooth, spec = read_opts(channel_opts, allow_callable=True)
    if spec is None:
        yield from _callable_pyramid(image, shrink, n_per_oct, smooth, channel_opts["channels"])
        return
    if spec.dtype == np.uint8:
        _require_u8(image, spec.key)
    H, W = image.shape
    eng = _engine.get_engine(H, W, image.dtype, shrink, n_per_oct, smooth, 1, channels=spec)
    if eng.plan.n_levels == 0:
        return
    mine = None
    for l in range(eng.plan.n_levels):
        if eng.epoch != mine:
            # first level -- or the (cached, shared) engine has served another image since the last one was
            # yielded: bring this generator's image and octaves back
            eng.load_images(image)
            eng.reset_step(None, octaves=True)
            eng.launch_octaves()
            mine = eng.epoch
        eng.launch_level(l)
        yield np.atleast_3d(eng.read_level(0, l)), eng.plan.scales[l]


def _callable_pyramid(image, shrink, n_per_oct, smooth, func):
    """channel_pyramid around a channel function this build has no kernel for (reference channels.py:119,136 calls
    whatever callable channel_opts["channels"] holds): octaves, each level's resize and cast (wb_resize_level_launch),
    avg_pool_2 and smooth_image_3d of the function's result (wb_pool_smooth_launch) run on the GPU; the caller's function
    runs where it runs -- it is handed the resized image as a host ndarray of the image's dtype and returns [H,W(,C)]
    uint8 or float32."""
    if shrink not in (1, 2):
        raise NotImplementedError("shrink must be 1 or 2 around a channel function without a kernel")
    H, W = image.shape
    eng = _engine.get_engine(H, W, image.dtype, shrink, n_per_oct, smooth, 1, channels=SPECS["grad_hist"])   # octaves, level table, taps
    mine = None
    for l in range(eng.plan.n_levels):
        if eng.epoch != mine:
            eng.load_images(image)
            eng.reset_step(None, octaves=True)
            eng.launch_octaves()
            mine = eng.epoch
        im = eng.resize_level(l)
        chns = np.asarray(func(im))
        if chns.ndim not in (2, 3) or chns.shape[:2] != im.shape:
            raise ValueError(f"the channel function returned shape {chns.shape} for a {im.shape} image")
        if chns.dtype not in (np.dtype(np.uint8), np.dtype(np.float32)):
            raise NotImplementedError(f"channel arrays of dtype {chns.dtype} have no avg_pool_2 / smooth kernel (uint8 and float32 do)")
        yield np.atleast_3d(eng.pool_smooth(chns if chns.ndim == 3 else chns[..., None])), eng.plan.scales[l]
