"""``DTree`` -- the weak classifier of the cascade (reference training.py:23-96).

Same constructor, attributes, proto I/O and ``predict_on_image``/``apply``/``predict``
signatures as the reference so that ``from waldboost_amd.training import DTree`` is a drop-in;
the evaluation runs in a HIP kernel (csrc/wb_cascade.hip: tree_eval_kernel).  ``DTree.fit``
(sklearn training, reference training.py:33-50) is out of scope of this build.
"""
import numpy as np

from . import _native as nat
from .compare import channel_tensor


_ARRAYS = ("threshold", "prediction", "feature", "left", "right")
_REBINDS = [0]          # bumped whenever an array attribute of any DTree is rebound (Model.device_cascade's cache watches it)


class DTree:
    def __init__(self, feature, threshold, left, right, prediction):
        # reference training.py:24-31.  The five arrays are views into ONE private block of bytes per tree, so that
        # `content()` -- what Model.device_cascade compares against the copy on the GPU -- is one buffer per tree;
        # they stay writable (the reference's are) and an in-place edit shows up in that buffer.
        feature = np.array([f if f is not None else [0, 0, 0] for f in feature], np.uint8).reshape(-1, 3)
        threshold = np.array(threshold, np.float32)
        left = np.array(left, np.int8)
        right = np.array(right, np.int8)
        prediction = np.array(prediction, np.float32)
        k = left.size
        if not (feature.shape[0] == threshold.size == right.size == prediction.size == k) or threshold.ndim != 1 \
                or left.ndim != 1 or right.ndim != 1 or prediction.ndim != 1:
            raise ValueError("DTree arrays must have one entry per node")
        blob = np.empty(13 * k, np.uint8)
        views = dict(threshold=blob[:4 * k].view(np.float32), prediction=blob[4 * k:8 * k].view(np.float32),
                     feature=blob[8 * k:11 * k].reshape(k, 3), left=blob[11 * k:12 * k].view(np.int8),
                     right=blob[12 * k:].view(np.int8))
        for name, src in (("threshold", threshold), ("prediction", prediction), ("feature", feature), ("left", left),
                          ("right", right)):
            views[name][...] = src
            object.__setattr__(self, name, views[name])
        object.__setattr__(self, "_blob", blob)
        self.node = self.left >= 0
        self.node_idx = np.flatnonzero(self.node)

    def __setattr__(self, name, value):
        # rebinding one of the arrays (w.threshold = other) detaches the tree from its block: content() then reads
        # the five arrays one by one
        if name in _ARRAYS:
            object.__setattr__(self, "_blob", None)
            _REBINDS[0] += 1
        object.__setattr__(self, name, value)

    # copy.copy / copy.deepcopy / pickle: NumPy copies (or unpickles) every view on its own, which would leave a tree whose
    # `_blob` no longer shares memory with `threshold`, `feature`, ...: an in-place edit of the copy would then never reach
    # content(), and Model.device_cascade would keep scanning with the stale GPU cascade.  So the state that travels is the
    # five arrays (plus whatever else a caller hung on the tree), and the restored tree gets a block and views of its own.
    def __getstate__(self):
        state = {k: v for k, v in self.__dict__.items() if k not in ("_blob", "_dev", "node", "node_idx") and k not in _ARRAYS}
        state["_arrays"] = {a: np.array(getattr(self, a)) for a in _ARRAYS}
        return state

    def __setstate__(self, state):
        state = dict(state)
        arrays = state.pop("_arrays", None)
        if arrays is None:              # (a pickle written before this protocol existed: plain attribute dict)
            arrays = {a: state.pop(a) for a in _ARRAYS}
            for k in ("_blob", "_dev", "node", "node_idx"):
                state.pop(k, None)
        DTree.__init__(self, arrays["feature"], arrays["threshold"], arrays["left"], arrays["right"], arrays["prediction"])
        for k, v in state.items():
            object.__setattr__(self, k, v)

    def content(self):
        """The tree's current arrays as one bytes-like object (cheap: the private block itself unless an array
        was rebound)."""
        if self._blob is not None:
            return self._blob
        return b"".join([np.ascontiguousarray(getattr(self, a)).tobytes() + b"|" for a in _ARRAYS])

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
        key = (str(dev), bytes(self.content()))          # (an edited tree is uploaded again)
        cache = self.__dict__.setdefault("_dev", {})
        if key not in cache:
            cache.clear()
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
