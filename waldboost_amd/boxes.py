"""Minimal stand-in for the third-party ``bbx.Boxes`` container the reference returns
(call sites: reference model.py:139,147,177,179 and __init__.py:126-130).

``bbx`` is not vendored upstream and is absent here; only the members those call sites (and
the docstring example at model.py:166-171) use are provided.  Parity at this boundary is
unpinned upstream.
"""
import numpy as np


class Boxes:
    def __init__(self, coords, **fields):
        self._c = np.asarray(coords).reshape(-1, 4)
        self._fields = {k: np.asarray(v) for k, v in fields.items()}

    def get(self):
        """(N,4) array of [xmin, ymin, xmax, ymax]."""
        return self._c

    def set_field(self, name, value):
        value = np.asarray(value)
        if value.shape[0] != len(self):
            raise ValueError(f"field {name!r} has {value.shape[0]} rows, boxes have {len(self)}")
        self._fields[name] = value

    def add_field(self, name, value):
        """As set_field (bbx exposes both; reference samples.py:157 adds 'regression_target' with it)."""
        self.set_field(name, value)

    def get_field(self, name):
        return self._fields[name]

    def has_field(self, name):
        return name in self._fields

    def fields(self):
        return list(self._fields)

    def normalized(self, scale=1.0):
        return Boxes((self._c * np.float32(scale)).astype(self._c.dtype), **self._fields)

    def __len__(self):
        return self._c.shape[0]

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            idx = [idx]
        return Boxes(self._c[idx], **{k: v[idx] for k, v in self._fields.items()})

    def __repr__(self):
        return f"Boxes(n={len(self)}, fields={self.fields()})"


def concatenate(boxes, fields=None):
    boxes = list(boxes)
    if not boxes:
        return Boxes(np.empty((0, 4), "f"))
    names = list(boxes[0].fields()) if fields is None else list(fields)
    out = Boxes(np.concatenate([b.get() for b in boxes]))
    for n in names:
        out._fields[n] = np.concatenate([b.get_field(n) for b in boxes])
    return out


def iou(a, b):
    """Pairwise intersection-over-union of two box lists -> (len(a), len(b)) float array
    (stand-in for ``bbx.iou``, reference samples.py:133; areas are (x2-x1)*(y2-y1))."""
    A = np.asarray(a.get(), np.float64).reshape(-1, 4)
    B = np.asarray(b.get(), np.float64).reshape(-1, 4)
    iw = np.clip(np.minimum(A[:, None, 2], B[None, :, 2]) - np.maximum(A[:, None, 0], B[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(A[:, None, 3], B[None, :, 3]) - np.maximum(A[:, None, 1], B[None, :, 1]), 0, None)
    inter = iw * ih
    area_a = ((A[:, 2] - A[:, 0]) * (A[:, 3] - A[:, 1]))[:, None]
    area_b = ((B[:, 2] - B[:, 0]) * (B[:, 3] - B[:, 1]))[None, :]
    union = area_a + area_b - inter
    return np.where(union > 0, inter / np.where(union > 0, union, 1), 0.0)
