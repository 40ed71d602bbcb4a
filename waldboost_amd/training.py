"""``DTree`` -- the weak classifier of the cascade (reference training.py:23-96).

Same constructor, attributes, proto I/O and ``predict_on_image``/``apply``/``predict``
signatures as the reference so that ``from waldboost_amd.training import DTree`` is a drop-in;
the evaluation runs in a HIP kernel (csrc/wb_cascade.hip: tree_eval_kernel).  ``DTree.fit``
(sklearn training, reference training.py:33-50) is out of scope of this build.
"""
import numpy as np

from . import _native as nat
from .compare import channel_tensor


class DTree:
    def __init__(self, feature, threshold, left, right, prediction):
        # reference training.py:24-31
        self.feature = np.array([f if f is not None else [0, 0, 0] for f in feature], np.uint8).reshape(-1, 3)
        self.threshold = np.array(threshold, np.float32)
        self.left = np.array(left, np.int8)
        self.right = np.array(right, np.int8)
        self.prediction = np.array(prediction, np.float32)
        self.node = self.left >= 0
        self.node_idx = np.flatnonzero(self.node)
        k = self.left.size
        if not (self.feature.shape[0] == self.threshold.size == self.right.size == self.prediction.size == k):
            raise ValueError("DTree arrays must have one entry per node")

    @staticmethod
    def fit(*args, **kwargs):
        raise NotImplementedError("DTree.fit (training) is outside the MI355X detection hot path; "
                                  "train with the reference and load the .pb here")

    # ---- wire format (reference training.py:51-72, model.proto DTree)
    @staticmethod
    def from_proto(proto):
        ftr = np.array(proto.feature).reshape((-1, 3))
        ftr = [tuple(x) if x[0] >= 0 else None for x in ftr]
        return DTree(ftr, np.array(proto.threshold), np.array(proto.left), np.array(proto.right),
                     np.array(proto.prediction))

    def as_proto(self, proto):
        proto.Clear()
        # the reference tests `f is not None` on rows of a uint8 array, which is always true, so
        # leaves are written as 0,0,0 (never -1,-1,-1); kept for byte-compatible files
        proto.feature.extend(int(x) for x in self.feature.reshape(-1))
        proto.threshold.extend(float(x) for x in self.threshold)
        proto.left.extend(int(x) for x in self.left)
        proto.right.extend(int(x) for x in self.right)
        proto.prediction.extend(float(x) for x in self.prediction)

    # ---- evaluation
    def _device_arrays(self, dev):
        import torch
        key = str(dev)
        cache = self.__dict__.setdefault("_dev", {})
        if key not in cache:
            cache[key] = tuple(torch.from_numpy(np.array(a)).to(dev) for a in
                               (self.feature, self.threshold, self.left, self.right, self.prediction))
        return cache[key]

    def predict_on_image(self, X, rs, cs) -> np.ndarray:
        """Leaf prediction for the windows with origins (rs[i], cs[i]) of channel image
        X[u,v,C] (reference training.py:84-96)."""
        import torch
        lib = nat.load()
        dev = nat.require_gpu()
        rs = np.asarray(rs)
        cs = np.asarray(cs)
        if rs.size == 0:
            return np.empty(0, np.float32)
        u, v, C = X.shape
        fmax = self.feature[self.node].max(axis=0) if self.node.any() else np.zeros(3, np.int64)
        if rs.min() < 0 or cs.min() < 0 or rs.max() + int(fmax[0]) >= u or cs.max() + int(fmax[1]) >= v or int(fmax[2]) >= C:
            raise IndexError("window feature outside the channel image")
        Xd, wb_dt = channel_tensor(X, dev)        # any dtype, compared as NumPy would (compare.py)
        rd = torch.from_numpy(rs.astype(np.int32)).to(dev)
        cd = torch.from_numpy(cs.astype(np.int32)).to(dev)
        out = torch.empty(rs.size, dtype=torch.float32, device=dev)
        f, t, l, r, p = self._device_arrays(dev)
        nat.check(lib.wb_tree_eval_launch(nat.stream_ptr(), nat.ptr(Xd), wb_dt, u, v, C, nat.ptr(rd), nat.ptr(cd), rs.size,
                                          nat.ptr(f), nat.ptr(t), nat.ptr(l), nat.ptr(r), nat.ptr(p),
                                          self.left.size, nat.ptr(out)), "wb_tree_eval_launch")
        return out.cpu().numpy()

    def apply(self, X):
        """Index of the leaf each sample X[i] (shape (m,n,C)) reaches (reference training.py:73-81)."""
        import torch
        lib = nat.load()
        dev = nat.require_gpu()
        N, m, n, C = X.shape
        if N == 0:
            return np.zeros(0, "i")
        fmax = self.feature[self.node].max(axis=0) if self.node.any() else np.zeros(3, np.int64)
        if int(fmax[0]) >= m or int(fmax[1]) >= n or int(fmax[2]) >= C:
            raise IndexError("tree feature outside the sample")
        Xd, wb_dt = channel_tensor(X, dev)
        out = torch.empty(N, dtype=torch.int32, device=dev)
        f, t, l, r, p = self._device_arrays(dev)
        nat.check(lib.wb_tree_apply_launch(nat.stream_ptr(), nat.ptr(Xd), wb_dt, N, m, n, C, nat.ptr(f), nat.ptr(t),
                                           nat.ptr(l), nat.ptr(r), self.left.size, nat.ptr(out)), "wb_tree_apply_launch")
        return out.cpu().numpy().astype("i")

    def predict(self, X):
        """prediction[apply(X)] (reference training.py:82-83)."""
        return self.prediction[self.apply(X)]

    def depth(self):
        def d(n):
            return 0 if self.left[n] < 0 else 1 + max(d(int(self.left[n])), d(int(self.right[n])))
        return d(0)
