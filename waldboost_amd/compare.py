"""Channel arrays of any dtype, as the cascade kernels read them.

The reference compares ``X[...] <= threshold`` in NumPy (training.py:92, model.py:199) with float32
thresholds, so X's dtype decides the arithmetic of the comparison (NumPy-2 promotion):

  uint8                                     -> the kernels' uint8 path (a byte compares as its exact float32 value)
  float32                                   -> as is
  bool, int8, int16, uint16, float16        -> float32 comparison; every value is exact in float32
  float64, int32, uint32, int64, uint64     -> float64 comparison (int64/uint64 are first rounded to float64,
                                               as NumPy does)

For the float64 comparisons the value is replaced by the smallest float32 that is not below it (rounding toward
+inf): for a float32 threshold t and any real v,  v <= t  <=>  up32(v) <= t  (t is itself a float32 not below v
exactly when it is not below up32(v)), NaN stays NaN, so every node decision -- and with it every score -- is the
reference's.  Elementwise dtype conversion on the device (torch); the cascade itself is in csrc/.
"""
import numpy as np

from . import _native as nat

_EXACT_F32 = {"bool", "int8", "int16", "uint16", "float16", "bfloat16"}
_VIA_F64 = {"float64", "int32", "uint32", "int64", "uint64"}


def dtype_name(X):
    return str(getattr(X, "dtype", None)).replace("torch.", "")


def channel_tensor(X, dev):
    """(contiguous device tensor, WB_DTYPE_*) holding X for the comparisons above; TypeError for other dtypes."""
    import torch
    name = dtype_name(X)
    if name not in _EXACT_F32 | _VIA_F64 | {"uint8", "float32"}:
        raise TypeError(f"channel array of dtype {name} cannot be compared with float32 thresholds here")
    t = X if isinstance(X, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(X))
    t = t.to(dev)
    if name == "uint8":
        return t.contiguous(), nat.WB_DTYPE_U8
    if name == "float32":
        return t.contiguous(), nat.WB_DTYPE_F32
    if name in _EXACT_F32:
        return t.to(torch.float32).contiguous(), nat.WB_DTYPE_F32
    v = t.to(torch.float64)
    f = v.to(torch.float32)                                   # round to nearest ...
    below = f.to(torch.float64) < v                           # ... and one step up where that fell short of v
    f = torch.where(below, torch.nextafter(f, torch.full_like(f, float("inf"))), f)
    return f.contiguous(), nat.WB_DTYPE_F32
