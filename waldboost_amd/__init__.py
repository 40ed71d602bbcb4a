"""waldboost_amd -- MI355X-native drop-in for the detection hot path of RomanJuranek/waldboost.

Keeps the reference surface for that path (reference waldboost/__init__.py:50-72):
``Model``, ``DTree``, ``channels.channel_pyramid``/``grad_hist``, ``load``/``save``, ``detect``.
All compute runs in hand-written HIP kernels (csrc/) through the C ABI in
include/waldboost_hip.h; there is no CPU fallback.
"""
from . import channels
from .boxes import Boxes, concatenate
from .model import Model
from .training import DTree

__version__ = "0.1.0"

load = load_model = Model.load


def save_model(model, filename):
    """Save model to file. See Model.save"""
    model.save(filename)


save = save_model

default_channel_opts = dict(shrink=2, n_per_oct=8, smooth=1, channels=channels.grad_hist)

__all__ = ["Model", "DTree", "Boxes", "concatenate", "channels", "load", "load_model", "save", "save_model",
           "default_channel_opts"]
