"""waldboost_amd -- MI355X-native drop-in for the detection hot path of RomanJuranek/waldboost.

Keeps the reference surface for that path (reference waldboost/__init__.py:50-72):
``Model``, ``DTree``, ``channels.channel_pyramid``/``grad_hist``, ``load``/``save``, ``detect``.
All compute runs in hand-written HIP kernels (csrc/) through the C ABI in
include/waldboost_hip.h; there is no CPU fallback.
"""
from . import channels, fpga, model_pb2, samples, testing
from .boxes import Boxes, concatenate
from .model import Model
from .samples import SamplePool
from .training import DTree

__version__ = "0.1.0"

load = load_model = Model.load


def save_model(model, filename):
    """Save model to file. See Model.save"""
    model.save(filename)


save = save_model

def detect(image, *models, channel_opts=None, response_scale=None):
    """Detect objects with several models sharing one channel pyramid (reference
    waldboost/__init__.py:75-130): returns Boxes with 'scores' (multiplied by response_scale[k]) and
    'label' (index of the model that fired).  Order: pyramid level, then model, then row-major
    window order -- as the reference's nested loops produce it.  The reference's `np.int` label
    dtype (removed in NumPy 1.24) is int64 here."""
    import numpy as np
    from . import engine as _engine
    if not models:
        raise ValueError("detect needs at least one model")
    channel_opts = channel_opts or models[0].channel_opts
    if response_scale is None:
        response_scale = [1] * len(models)
    response_scale = np.array(response_scale, "f")
    if response_scale.size != len(models):
        raise ValueError("Wrong response_scale parameter")
    channels._validate_image(image)
    shrink, n_per_oct, smooth, spec = channels.read_opts(channel_opts)
    H, W = image.shape
    eng = _engine.get_engine(H, W, image.dtype, shrink, n_per_oct, smooth, 1, channels=spec)
    if eng.plan.n_levels == 0:
        return concatenate([], ["scores", "label"])
    eng.load_images(image)
    # one pyramid for all models: as threshold ranks of the UNION of their thresholds (one byte per value, the byte-tile
    # cascade) when that union fits a rank table, as float32 channels otherwise
    group = None
    if spec.key == "grad_hist" and not _engine._NO_RANKS and all(m.shape[2] == spec.n_channels for m in models):
        group = _engine.rank_group([m.device_cascade() for m in models])
    views = group.views if group is not None else [None] * len(models)
    dms = [v if v is not None else m.device_cascade() for m, v in zip(models, views)]
    for m in models:
        assert m.shape[2] == eng.spec.n_channels, f"Invalid number of channels. Expected {m.shape[2]} given {eng.spec.n_channels}."
    # the whole sequence -- octaves, the pyramid, per model its scan, boxes and read-back -- as ONE hipGraph replay with
    # ONE wait from the second call on (like Model.detect); model by model where that form does not apply
    fins = eng.detect_multi_run(dms, ranks=group is not None)
    if fins is not None:
        # (None in a model's place: its results did not fit this time -- that cascade alone is scanned again, on the pyramid
        # the call left resident)
        res = [m._collect(eng, d, eng._casc_state(d), True, fin) if fin is not None else None for m, d, fin in zip(models, dms, fins)]
        res = [r if r is not None else m.scan_engine(eng, view=v) for r, m, v in zip(res, models, views)]
    else:
        if group is not None:
            eng.run_channels(rank_dm=group.views[0], floats=False)
        else:
            eng.run_channels()
        res = [m.scan_engine(eng, view=v) for m, v in zip(models, views)]
    # level-major, then model, then row-major: every model's result is ordered by (level, r, c) already, so ONE stable sort
    # by (level, model) gives the reference's nested-loop order (a Python loop over levels x models building Boxes was
    # 0.8 ms of a 1.1 ms call)
    K = len(models)
    level = np.concatenate([r["level"] for r in res]).astype(np.int64)
    label = np.concatenate([np.full(r["level"].size, k, np.int64) for k, r in enumerate(res)])
    order = np.argsort(level * K + label, kind="stable")
    boxes = np.concatenate([r["boxes"] for r in res])[order]
    scores = np.concatenate([r["scores"] * response_scale[k] for k, r in enumerate(res)])[order]
    out = Boxes(boxes.reshape(-1, 4).astype(np.float32, copy=False))
    out.set_field("scores", scores.astype(np.float32, copy=False))
    out.set_field("label", label[order])
    return out


default_channel_opts = dict(shrink=2, n_per_oct=8, smooth=1, channels=channels.grad_hist)

__all__ = ["Model", "DTree", "Boxes", "concatenate", "SamplePool", "channels", "fpga", "samples", "testing", "detect", "load", "load_model", "save", "save_model",
           "default_channel_opts"]
