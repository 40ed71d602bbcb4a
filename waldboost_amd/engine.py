"""Device-side execution of the hot path: buffers, launches, hipGraph capture.

``PyramidEngine`` owns the HBM-resident state of one (image shape, dtype, channel_opts,
batch) configuration and drives the three kernel groups through the C ABI:

    octaves (wb_octaves_launch) -> channels (wb_channels_launch) -> cascade (wb_cascade_launch)

torch is used only for device memory, streams and graph capture; all compute is in
csrc/*.hip.  Nothing here falls back to the CPU.
"""
import contextlib
import ctypes as C
import logging
import os
import weakref

import numpy as np

from . import _native as nat
from .chanfunc import SPECS
from .plan import PyramidPlan, N_CHANNELS

_TORCH_DT = {}
_log = logging.getLogger("waldboost_amd")


@contextlib.contextmanager
def capturing(graph, **kw):
    """``torch.cuda.graph(graph, **kw)`` with Python's cyclic collector out of the way.  An object it happens to free in
    the middle of a stream capture -- a page-locked tensor (its allocator records events), another engine's CUDAGraph,
    a stream -- makes HIP calls that invalidate the capture ("operation failed due to a previous error during capture");
    torch no longer collects on entry by itself (>= 2.9).  So: collect first, then keep the collector off until the
    capture has ended."""
    import gc
    import torch
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was:
            gc.enable()

_NO_DETECT_GRAPH = bool(int(os.environ.get("WB_NO_DETECT_GRAPH", "0")))
_NO_RANKS = bool(os.environ.get("WB_NO_RANKS"))     # diagnostic: float32 channels + planar float tile everywhere
# Model-specialised cascade kernels (csrc/wb_jit.hip): a cascade that has been scanned this many times on byte tiles is
# compiled with its stage records as constants (hiprtc, ~2 s once; cached on disk).  WB_CASC_JIT=0: never automatically
# (DeviceCascade.specialize() still works), WB_CASC_JIT_AFTER=n: after n scans (default 3).
# A freshly specialised cascade kernel is cross-checked against the generic kernel on the first real image it meets (this
# many times; 0 = never): the library's self-test (wb_model_specialize) runs on synthetic bytes, this one on the caller's.
_LIVE_CHECKS = int(os.environ.get("WB_JIT_LIVE_CHECKS", "2"))
_CHECK_KEYS = bool(os.environ.get("WB_CHECK_KEYS"))         # diagnostic: verify the clean-keys flag against the device before every fused step
_NO_FUSED_RESET = bool(os.environ.get("WB_NO_FUSED_RESET"))   # diagnostic: one memset launch per step, as before
_JIT_AUTO = os.environ.get("WB_CASC_JIT", "1") != "0"
_JIT_AFTER = int(os.environ.get("WB_CASC_JIT_AFTER", "3"))
_FORCE_RANK16 = bool(os.environ.get("WB_FORCE_RANK16"))   # diagnostic / bench: 16-bit ranks also where a byte would do


def _torch_dtype(np_dtype):
    import torch
    if not _TORCH_DT:
        _TORCH_DT.update({np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32,
                          np.dtype(np.float64): torch.float64})
    try:
        return _TORCH_DT[np.dtype(np_dtype)]
    except KeyError:
        raise NotImplementedError(f"dtype {np.dtype(np_dtype)} has no HIP kernel") from None


# Image dtypes (reference channels.py:122 keeps image.dtype through the pyramid): uint8 and float32 have their own
# kernels; float64 and the integer types whose values are exact in float64 are held as float64 on the device, with
# a code that tells the kernels how avg_pool_2 wraps and how the resize result is cast back (waldboost_hip.h).
_IMAGE_CODES = {np.dtype(np.uint8): (np.uint8, nat.WB_DTYPE_U8), np.dtype(np.float32): (np.float32, nat.WB_DTYPE_F32),
                np.dtype(np.float64): (np.float64, nat.WB_DTYPE_F64), np.dtype(np.int8): (np.float64, nat.WB_DTYPE_I8),
                np.dtype(np.int16): (np.float64, nat.WB_DTYPE_I16), np.dtype(np.uint16): (np.float64, nat.WB_DTYPE_U16),
                np.dtype(np.int32): (np.float64, nat.WB_DTYPE_I32), np.dtype(np.uint32): (np.float64, nat.WB_DTYPE_U32),
                # (int64 / uint64: only values exact in float64 -- load_images checks; bool adds are logical ors; every
                # float16 add rounds to float16)
                np.dtype(np.int64): (np.float64, nat.WB_DTYPE_I64), np.dtype(np.uint64): (np.float64, nat.WB_DTYPE_U64),
                np.dtype(np.bool_): (np.float64, nat.WB_DTYPE_BOOL), np.dtype(np.float16): (np.float64, nat.WB_DTYPE_F16)}


def image_code(np_dtype):
    """(storage dtype on the device, WB_DTYPE_* code) of an image dtype; NotImplementedError if it has no kernel."""
    try:
        return _IMAGE_CODES[np.dtype(np_dtype)]
    except KeyError:
        raise NotImplementedError(
            f"image dtype {np.dtype(np_dtype)} has no HIP kernel (uint8, float16 / 32 / 64, bool and the 8 to 64 bit "
            "integers are supported)") from None


def array_dtype(images):
    """NumPy dtype of an image batch given as an ndarray or a torch tensor, checked against the kernels' image types
    (NotImplementedError otherwise -- raised before any launch or collective)."""
    dt = images.dtype if isinstance(images, np.ndarray) else str(images.dtype).replace("torch.", "")
    try:
        dt = np.dtype(dt)
    except TypeError:
        raise NotImplementedError(f"image dtype {images.dtype} has no HIP kernel") from None
    image_code(dt)
    return dt


def orientation_constants():
    """cos/sin of the 4 unsigned orientations, built exactly as reference channels.py:43-46
    builds them (NumPy fp64; note cos(pi/2) = 6.1e-17, not 0)."""
    theta = np.linspace(0, np.pi, N_CHANNELS + 1)
    return np.concatenate([np.cos(theta[:-1]), np.sin(theta[:-1])]).astype(np.float64)


def theta_as_f32(theta):
    """The fp32 value t such that ``hs >= theta`` (NumPy-2 promotion) == ``hs >= t`` for every
    fp32 ``hs``.  Python floats are weak scalars (rounded to fp32 by NumPy); NumPy fp64/int
    scalars force an fp64 comparison, which equals comparing with the smallest fp32 >= theta."""
    if isinstance(theta, (np.float32, np.float16)):
        return np.float32(theta)
    if type(theta) in (float, int, bool):
        with np.errstate(over="ignore"):
            return np.float32(theta)
    t64 = np.float64(theta)
    if np.isnan(t64):
        return np.float32(np.nan)
    with np.errstate(over="ignore"):
        t32 = np.float32(t64)
    if np.float64(t32) < t64:
        t32 = np.nextafter(t32, np.float32(np.inf))
    return t32


class DeviceCascade:
    """A WbModel handle built from the Python-side stage list (immutable snapshot)."""

    def __init__(self, shape, classifier, theta):
        lib = nat.load()
        nat.require_gpu()
        m, n, Cc = (int(x) for x in shape)
        T = len(classifier)
        node_off = np.zeros(T + 1, np.int32)
        for i, w in enumerate(classifier):
            node_off[i + 1] = node_off[i] + w.left.size
        cat = lambda name, dt: (np.ascontiguousarray(np.concatenate([getattr(w, name).reshape(-1) for w in classifier]).astype(dt))
                                if T else np.zeros(0, dt))
        feature = cat("feature", np.uint8)
        threshold = cat("threshold", np.float32)
        left = cat("left", np.int8)
        right = cat("right", np.int8)
        pred = cat("prediction", np.float32)
        th = np.array([theta_as_f32(t) for t in theta], np.float32)
        h = C.c_void_p()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        nat.check(lib.wb_model_create(T, vp(node_off), vp(feature), vp(threshold), vp(left), vp(right), vp(pred),
                                      vp(th), m, n, Cc, C.byref(h)), "wb_model_create")
        self.handle = h
        self._lib = lib
        self._owned = True
        self._scans = {}                 # byte-tile scans so far, per channel dtype code
        self._jit_failed = set()
        # whose rank tables this cascade scans: its own (a RankGroup's for a member view).  An opaque token, compared by
        # identity -- never the cascade or the group itself: a self-reference (or view -> group -> views) is a cycle, and
        # a cycle's __del__ (wb_model_destroy / wb_rankgroup_destroy: hipFree) would run whenever the cyclic collector
        # happens to, e.g. inside somebody's stream capture or between a lane's enqueue and collect
        self.rank_key = object()
        self._read_info()

    def _read_info(self):
        info = nat.WbModelInfo()
        nat.check(self._lib.wb_model_info(self.handle, C.byref(info)), "wb_model_info")
        self.n_stages, self.depth = info.n_stages, info.depth
        self.m, self.n, self.C = info.m, info.n, info.C
        self.tile_rows, self.tile_cols, self.lds_bytes = info.tile_rows, info.tile_cols, info.lds_bytes
        # the model's thresholds fit rank tables: the channel kernel can write float32 channels as one-byte ranks
        # (WB_DTYPE_RANK8) and the cascade scan them exactly as it would the floats, from a quarter of the bytes
        self.rank_ok = bool(info.rank_ok)
        # ... or, with more thresholds per channel than a byte ranks (long soft cascades), as two-byte ranks (WB_DTYPE_RANK16)
        self.rank16_ok = bool(info.rank16_ok)
        # the rank form the detection path uses for this cascade (None: float32 channels)
        self.rank_dtype = (nat.WB_DTYPE_RANK16 if (self.rank16_ok and (_FORCE_RANK16 or not self.rank_ok)) else
                           nat.WB_DTYPE_RANK8 if self.rank_ok else None)

    @classmethod
    def _view(cls, handle, group):
        """A member view of a RankGroup: same interface, handle owned by the group."""
        self = cls.__new__(cls)
        self._lib = nat.load()
        self.handle = handle
        self._owned = False
        self._scans, self._jit_failed = {}, set()
        self.rank_key = group.token      # (a token, not the group: see __init__)
        self.group = group               # keeps the handle's owner alive; the group does NOT refer back to its views
        self._read_info()
        return self

    def specialized(self):
        """Which byte tiles have a model-specialised kernel loaded: subset of {WB_DTYPE_U8, WB_DTYPE_RANK8, WB_DTYPE_RANK16}."""
        info = nat.WbModelInfo()
        nat.check(self._lib.wb_model_info(self.handle, C.byref(info)), "wb_model_info")
        return {d for bit, d in ((1, nat.WB_DTYPE_U8), (2, nat.WB_DTYPE_RANK8), (4, nat.WB_DTYPE_RANK16)) if info.specialized & bit}

    def specialize(self, chn_dtype=None):
        """Compile and load the model-specialised tile kernel (wb_model_specialize) for a kind of byte tile: the threshold
        ranks by default when the model has rank tables, else uint8 channels.  Returns False when this model has no
        specialised kernel (node-walk models, cascades beyond the LDS mirror); raises NativeError if the compiler fails."""
        if chn_dtype is None:
            chn_dtype = self.rank_dtype if self.rank_dtype is not None else nat.WB_DTYPE_U8
        had = chn_dtype in self.specialized()
        rc = self._lib.wb_model_specialize(self.handle, chn_dtype)
        if rc == nat.WB_ERR_UNSUPPORTED:
            return False
        nat.check(rc, "wb_model_specialize")
        if not had:
            self.__dict__.setdefault("_live_left", {})[chn_dtype] = _LIVE_CHECKS      # (PyramidEngine.live_check)
        return True

    def use_specialized(self, enable):
        """The loaded specialised kernels on / off for every later scan of this cascade (off: the generic kernel)."""
        nat.check(self._lib.wb_model_use_specialized(self.handle, 1 if enable else 0), "wb_model_use_specialized")
        self.__dict__["_spec_off"] = not enable

    def live_checks_left(self, chn_dtype):
        return 0 if self.__dict__.get("_spec_off") else self.__dict__.get("_live_left", {}).get(chn_dtype, 0)

    def note_scan(self, chn_dtype, force=False):
        """Called by the engine before a scan on byte tiles: after _JIT_AFTER scans (force: now -- Model.detect is about
        to capture its graph) the cascade is worth specialising.  Never inside a stream capture (compilation and module
        loading are not capturable), never twice after a failure."""
        if not _JIT_AUTO or chn_dtype == nat.WB_DTYPE_F32 or chn_dtype in self._jit_failed:
            return
        n = self._scans.get(chn_dtype, 0) + 1
        if force and n < _JIT_AFTER:
            n = _JIT_AFTER
        self._scans[chn_dtype] = n
        if n == _JIT_AFTER:
            import torch
            if torch.cuda.is_current_stream_capturing():
                self._scans[chn_dtype] = n - 1
                return
            try:
                if not self.specialize(chn_dtype):
                    self._jit_failed.add(chn_dtype)
                    # (said once per cascade and tile kind: the generic kernel is ~10 % slower at batch 1)
                    _log.info("cascade of %d stages, depth %d: no model-specialised kernel for this model (%s); staying on "
                              "the generic tile kernel", self.n_stages, self.depth, nat.last_error())
            except nat.NativeError as exc:
                self._jit_failed.add(chn_dtype)      # (stay on the generic kernel)
                _log.warning("cascade of %d stages, depth %d: building the model-specialised kernel failed, staying on the "
                             "generic tile kernel (slower): %s", self.n_stages, self.depth, exc)

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self._owned:
                self._lib.wb_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class RankGroup:
    """Several cascades over ONE pyramid of threshold ranks (wb_rankgroup_create): the rank table of every channel is
    built from the union of the members' thresholds; ``view(i)`` is member i as a DeviceCascade whose rank tables are the
    group's.  Raises NotImplementedError when the union does not fit (more than 255 thresholds on a channel).
    Ownership runs one way -- view -> group -> member cascades -- so that a group nobody scans with any more is freed
    by reference counting, at once (its __del__ issues hipFree calls: not something to leave to the cyclic collector)."""

    def __init__(self, cascades):
        lib = nat.load()
        self._lib = lib
        self.members = list(cascades)                      # (the models must outlive the group)
        arr = (C.c_void_p * len(self.members))(*[dm.handle.value for dm in self.members])
        h = C.c_void_p()
        nat.check(lib.wb_rankgroup_create(arr, len(self.members), C.byref(h)), "wb_rankgroup_create")
        self.handle = h
        self.token = object()              # what the engine's rank buffer is tagged with (DeviceCascade.rank_key)

    def view(self, i):
        v = C.c_void_p()
        nat.check(self._lib.wb_rankgroup_model(self.handle, i, C.byref(v)), "wb_rankgroup_model")
        return DeviceCascade._view(v, self)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.wb_rankgroup_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class GroupViews:
    """What rank_group hands out: the group's member views (stable objects: the engine keys its scan states by them)."""

    def __init__(self, group):
        self.group = group
        self.members = group.members
        self.views = [group.view(i) for i in range(len(group.members))]


_GROUPS = {}


def rank_group(cascades):
    """The (cached) member views of the RankGroup of these cascades, or None when they cannot share a rank table."""
    key = tuple(id(dm) for dm in cascades)
    g = _GROUPS.get(key)
    if g is None or (g and any(a is not b for a, b in zip(g.members, cascades))):
        try:
            g = GroupViews(RankGroup(cascades))
        except NotImplementedError:
            g = False
        if len(_GROUPS) >= 8:
            _GROUPS.pop(next(iter(_GROUPS)))
        _GROUPS[key] = g
    return g or None


class DetBuffer:
    """Sharded append buffer of WbDet records (include/waldboost_hip.h: WB_DET_SHARDS regions of `cap` records,
    one counter each).  `counts` may be a view the caller provides -- PyramidEngine keeps the counters in its
    control block, so that ONE memset per step resets them together with the other accumulators."""

    def __init__(self, cap, dev, counts=None):
        import torch
        self.NS = nat.WB_DET_SHARDS
        self.cap = int(cap)
        self.counts = counts if counts is not None else torch.zeros(self.NS, dtype=torch.int32, device=dev)
        self.recs = torch.zeros((self.NS * self.cap, 4), dtype=torch.int32, device=dev)

    def zero(self):
        self.counts.zero_()

    def max_count(self):
        return int(self.counts.max().item())

    def valid_records(self, counts):
        """The valid records of all shards (shard order) as one int32 [n, 4] tensor, given the host copy of
        the shard counters: plain slices, no mask."""
        import torch
        parts = [self.recs[s * self.cap: s * self.cap + int(c)] for s, c in enumerate(counts) if c]
        if not parts:
            return self.recs[:0]
        return parts[0] if len(parts) == 1 else torch.cat(parts)

    def compact(self):
        """All valid records as one int32 [n, 4] tensor (shard order)."""
        import torch
        ar = torch.arange(self.cap, device=self.recs.device, dtype=torch.int32)
        mask = ar[None, :] < self.counts[:, None].clamp(max=self.cap)
        return self.recs.view(self.NS, self.cap, 4)[mask]


def sort_records(d):
    """Records ordered by (image, level, r, c) -- the reference's output order (row-major window
    grid, levels in pyramid order)."""
    import torch
    if d.shape[0] == 0:
        return d
    rc = d[:, 2].to(torch.int64) & 0xFFFFFFFF
    key = (d[:, 0].to(torch.int64) << 44) | (d[:, 1].to(torch.int64) << 32) | ((rc & 0xFFFF) << 16) | (rc >> 16)
    return d[torch.argsort(key)].contiguous()


class CapturedStep:
    """A hipGraph of one engine's step (PyramidEngine.capture).  The graph holds device ADDRESSES: of the control block
    and of the detection buffer, both of which the engine re-allocates when they must grow.  replay() refuses to run
    a graph captured before such a re-allocation (it would add into freed memory) -- capture again."""

    def __init__(self, engine, graph):
        # (a weak reference: an engine that keeps its own captured step -- batch_enqueue -- must not become a reference
        # cycle, to be freed by the collector at some arbitrary later moment)
        self._engine, self.graph, self.generation = weakref.ref(engine), graph, engine.generation

    @property
    def engine(self):
        return self._engine()

    def replay(self):
        if self.engine is None:
            raise RuntimeError("this captured step's engine no longer exists")
        if self.generation != self.engine.generation:
            raise RuntimeError("this captured step is stale: the engine re-allocated its control block or detection "
                               "buffer after the capture (a longer cascade, or a grown detection buffer); capture again")
        self.engine.ensure_clean_keys()          # (the graph holds no memset: see PyramidEngine.run)
        self.graph.replay()


class PyramidEngine:
    # detection records read back with the first copy of fetch() / the one copy of fetch_final() (28 bytes each; the
    # copies inside a captured graph have this fixed size).  WB_FETCH_ROWS: diagnostic override
    _FETCH_ROWS = int(os.environ.get("WB_FETCH_ROWS", "4096"))

    def __init__(self, H, W, dtype, shrink, n_per_oct, smooth, batch=1, exact_single=False, det_capacity=1 << 16,
                 channels=None):
        import torch
        self.lib = nat.load()
        self.dev = nat.require_gpu()
        self.dtype = np.dtype(dtype)
        self.spec = channels if channels is not None else SPECS["grad_hist"]
        if self.spec.dtype == np.uint8 and self.dtype != np.uint8:
            raise NotImplementedError(f"{self.spec.key} takes 8 bit images (uint8), got {self.dtype}")
        self.store_dtype, self.wb_dtype = image_code(self.dtype)       # (how the image sits in HBM, what the kernels are told)
        self.store_dtype = np.dtype(self.store_dtype)
        if self.spec.key != "grad_hist" and self.wb_dtype not in (nat.WB_DTYPE_U8, nat.WB_DTYPE_F32):
            raise NotImplementedError(f"{self.spec.key} has kernels for uint8 and float32 images, got {self.dtype}")
        self.tdtype = _torch_dtype(self.store_dtype)
        self.wide_keys = self.store_dtype == np.float64                # 64-bit (min, max) keys per octave
        self.batch = int(batch)
        self.plan = PyramidPlan(H, W, shrink, n_per_oct, smooth, exact_single=exact_single,
                                n_chn=self.spec.n_channels, chn_bytes=self.spec.dtype.itemsize, chan_func=self.spec.func_id)
        self.exact_single = exact_single
        p = self.plan
        p.batch_hint = self.batch     # (the tile lists' dispatch order depends on the images per launch: plan._short_last)
        dev = self.dev
        # flat allocations with 16 spare elements: the channel kernel fetches source rows with
        # 4-byte-aligned dword loads that may touch a few bytes past the last row
        self._img_flat = torch.zeros(self.batch * p.H * p.W + 16, dtype=self.tdtype, device=dev)
        self.img = self._img_flat[: self.batch * p.H * p.W].view(self.batch, p.H, p.W)
        self._oct_flat = torch.zeros(self.batch * p.oct_total + 16, dtype=self.tdtype, device=dev)
        self.oct = self._oct_flat[: self.batch * p.oct_total].view(self.batch, p.oct_total)
        # control block: [per-octave (min, max) keys | detection shard counters | alive[B, L, T]] -- everything the
        # kernels of one step ACCUMULATE into, contiguous, so that one memset at the start of a step resets it all
        # (three separate fills and the statistics reduction used to be four extra launches per image)
        self._mm_words = self.batch * max(p.n_oct, 1) * 2 * (2 if self.wide_keys else 1)
        self._alive_words = 0
        self.generation = 0           # bumped whenever a buffer a captured graph may address is re-allocated
        self._mm_clean = False        # the octaves' (min, max) keys are known to be zero (see run)
        self._alloc_ctrl(0)
        table, total = p.level_table()
        self.chn_stride = int(total)
        self.level_np = table
        self.levels = torch.from_numpy(table.view(np.uint8).copy()).to(dev) if p.n_levels else None
        taps, _ = p.tap_table()
        self.taps = torch.from_numpy(taps.view(np.uint8).copy()).to(dev)
        tiles = p.chan_tiles()
        self.n_chan_tiles = int(tiles.size)
        self.chan_tiles = torch.from_numpy(tiles.view(np.uint8).copy()).to(dev) if tiles.size else None
        self.chan_patches = self._tile_patches(tiles)
        # (16 spare elements: the cascade's uint8 tile load fetches 16-byte groups that may run past a level row)
        self._chn_flat = torch.zeros(self.batch * self.chn_stride + 16, dtype=_torch_dtype(self.spec.dtype), device=dev)
        self.chn = self._chn_flat[: self.batch * self.chn_stride].view(self.batch, self.chn_stride)
        self.cs_sn = orientation_constants()
        self._oct_off = (C.c_int64 * max(p.n_oct, 1))(*[int(x) for x in p.oct_off[:max(p.n_oct, 1)]])
        self.rank = self._rank_flat = self.rank_owner = None
        self._rank_wide = False
        self._level_tiles = None
        self.epoch = 0
        self.det_capacity = int(det_capacity)
        self._h_packed = self._h_alive = self._fetch_ev = None
        self._h2d_ev, self._upload_async = None, False
        self._mm_host = None
        self._inv_scales_d = self._final_dims = None
        self._order = None            # buffers of fetch_ordered_batch (scratch, out, page-locked copy), on first use
        self._alloc_det()
        if exact_single:
            # a channel function on a bare image: no resize happens, so the clip range is (-inf, +inf)
            if self.wb_dtype == nat.WB_DTYPE_U8:
                lo, hi = 0, 255
            else:
                lo, hi = int(nat_f32_key(-np.inf)), int(nat_f32_key(np.inf))
            if self.wide_keys:
                raise NotImplementedError("channel functions on a bare image take uint8 or float32 arrays")
            self.minmax[:, :, 0] = int(np.array([~np.uint32(lo)], np.uint32).view(np.int32)[0])   # word 0 stores max(~key), see csrc/wb_octaves.hip
            self.minmax[:, :, 1] = int(np.array([hi], np.uint32).view(np.int32)[0])

    def _tile_patches(self, tiles):
        """The per-tile patch table of wb_channels_launch_x for a tile list (uint8 images, the gradient-histogram kernels):
        filled on the host by the library with its own tile geometry, uploaded once.  None: the kernels compute the extents."""
        import torch
        if self.wb_dtype != nat.WB_DTYPE_U8 or not tiles.size or os.environ.get("WB_NO_TILE_PATCHES"):
            return None
        p = self.plan
        tiles = np.ascontiguousarray(tiles)
        out = np.zeros(tiles.size, nat.PATCH_DTYPE)
        rc = self.lib.wb_channels_tile_patches(self.spec.func_id, p.shrink, p.smooth, self.level_np.ctypes.data_as(C.c_void_p),
                                               p.n_levels, tiles.ctypes.data_as(C.c_void_p), int(tiles.size),
                                               out.ctypes.data_as(C.c_void_p))
        if rc == nat.WB_ERR_UNSUPPORTED:
            return None
        nat.check(rc, "wb_channels_tile_patches")
        return torch.from_numpy(out.view(np.uint8).copy()).to(self.dev)

    def _alloc_ctrl(self, alive_words):
        """(Re)allocate the control block with room for `alive_words` statistics words; the views into it follow."""
        import torch
        old = getattr(self, "ctrl", None)
        self.generation += 1
        NS = nat.WB_DET_SHARDS
        self._alive_words = int(alive_words)
        self.ctrl = torch.zeros(self._mm_words + NS + max(self._alive_words, 1), dtype=torch.int32, device=self.dev)
        mm = self.ctrl[: self._mm_words]
        if old is not None:
            mm.copy_(old[: self._mm_words])            # (exact_single keeps a preset clip range there)
        self.minmax = (mm.view(torch.int64) if self.wide_keys else mm).view(self.batch, -1, 2)
        self._counts = self.ctrl[self._mm_words: self._mm_words + NS]
        if getattr(self, "detb", None) is not None:
            self.detb.counts = self._counts
        self._casc = {}
        self._multi = {}              # detect_multi_run's captured sequences, by cascade list

    def _alive_view(self, T1):
        L = max(self.plan.n_levels, 1)
        need = self.batch * L * T1
        if need > self._alive_words:
            self._alloc_ctrl(need)
        o = self._mm_words + nat.WB_DET_SHARDS
        return self.ctrl[o: o + need].view(self.batch, L, T1)

    def reset_step(self, stt=None, octaves=True):
        """ONE memset: the accumulators the coming launches add into -- the octaves' (min, max) keys (unless they
        are preset / not recomputed), the detection counters and, with a cascade state, its alive[B, L, T]."""
        lo = 0 if (octaves and not self.exact_single) else self._mm_words
        hi = self._mm_words + (nat.WB_DET_SHARDS + stt["alive"].numel() if stt is not None else 0)
        if hi > lo:
            self.ctrl[lo:hi].zero_()

    def _alloc_det(self):
        """Sharded detection buffer; det_capacity is the total record capacity, split evenly over
        the shards."""
        cap = max(16, -(-self.det_capacity // nat.WB_DET_SHARDS))
        self.det_capacity = cap * nat.WB_DET_SHARDS
        self.generation += 1
        self.detb = DetBuffer(cap, self.dev, counts=self._counts)
        self.packed = None            # header + all valid records back to back (wb_det_pack_launch), allocated on first use
        for stt in getattr(self, "_casc", {}).values():
            stt.pop("graph", None)    # (a captured detect_run holds the old buffer's address)

    # ------------------------------------------------------------------ input
    def load_images(self, images):
        """images: ndarray / tensor [B,H,W] (or [H,W]) of the engine's dtype."""
        import torch
        self.epoch += 1                       # (lazy consumers notice that the resident images changed)
        want = (self.batch, self.plan.H, self.plan.W)
        if isinstance(images, np.ndarray):
            if images.dtype != self.dtype:
                raise TypeError(f"engine built for {self.dtype} images, got {images.dtype}")
            if images.dtype != self.store_dtype:
                if images.dtype.itemsize == 8 and images.dtype.kind in "iu" and images.size:
                    # (64-bit integers are held as float64: exact -- also their 2x2 sums -- below 2^51)
                    if max(abs(int(images.max())), abs(int(images.min()))) >= 1 << 51:
                        raise NotImplementedError("64 bit integer images are supported for values below 2**51 (they are held as float64)")
                images = images.astype(self.store_dtype)           # integer types travel as float64 (exact)
            if images.ndim == 2:
                images = images[None]
            if tuple(images.shape) != want:
                raise ValueError(f"expected images of shape {want}, got {tuple(images.shape)}")
            t = None
        else:
            if array_dtype(images) != self.dtype:
                raise TypeError(f"engine built for {self.dtype} images, got {images.dtype}")
            t = images[None] if images.dim() == 2 else images
            if t.dtype != self.tdtype:
                t = t.to(self.tdtype)                                   # (integer types travel as float64: exact)
            if self.dtype.itemsize == 8 and self.dtype.kind in "iu" and t.numel():
                if float(t.to(torch.float64).abs().max()) >= float(1 << 51):
                    raise NotImplementedError("64 bit integer images are supported for values below 2**51 (they are held as float64)")
            if tuple(t.shape) != want:
                raise ValueError(f"expected images of shape {want}, got {tuple(t.shape)}")
        self._upload_async = False
        if t is None:
            # (a page-locked staging buffer was measured: memcpy + DMA came out 10 % slower per Model.detect call
            # than torch's own pipelined upload from pageable memory; for detect_stream's lanes, where the DMA would
            # overlap other work, the host copy alone -- np.copyto, 45 us -- costs what the pageable upload blocks the
            # host for, 48 us, and torch's multi-threaded CPU copy, 26 us in a tight loop, takes milliseconds once its
            # worker threads have gone to sleep between images: tools/upload_probe.py)
            t = torch.from_numpy(np.ascontiguousarray(images))
        self._copy_in(self.img, t)

    def _copy_in(self, dst, t):
        """Host or device tensor -> resident image buffer on the current stream.  From PAGE-LOCKED host memory (a caller
        that decodes into pinned buffers: torch's pin_memory, hipHostMalloc / hipHostRegister'ed arrays -- asked of the
        driver per call, a microsecond) the copy is a true asynchronous DMA: nothing blocks the host, and the source must
        stay untouched until `wait_upload` (Model.detect returns after the whole call; detect_stream waits before it takes
        the next image from the caller's iterable).  From pageable memory torch stages the bytes itself and the call
        returns once they have left the caller's array (0.05 ms for a 1080p image)."""
        if t.device.type == "cpu" and t.is_pinned():
            import torch
            dst.copy_(t, non_blocking=True)
            if self._h2d_ev is None:
                self._h2d_ev = torch.cuda.Event()
            self._h2d_ev.record()
            self._upload_async = True
        else:
            dst.copy_(t, non_blocking=True)

    def wait_upload(self):
        """Block until the last asynchronous upload has left the caller's page-locked buffer (no-op otherwise)."""
        if self._upload_async:
            self._h2d_ev.synchronize()
            self._upload_async = False

    def load_slot(self, b, image):
        """One 2-D host image into slot b of the batch (Model.detect_stream fills a batch image by image)."""
        import torch
        self.epoch += 1
        self._upload_async = False
        if not isinstance(image, np.ndarray):
            if array_dtype(image) != self.dtype:
                raise TypeError(f"engine built for {self.dtype} images, got {image.dtype}")
            if tuple(image.shape) != (self.plan.H, self.plan.W):
                raise ValueError(f"expected an image of shape {(self.plan.H, self.plan.W)}, got {tuple(image.shape)}")
            if self.dtype.itemsize == 8 and self.dtype.kind in "iu" and image.numel():
                if float(image.to(torch.float64).abs().max()) >= float(1 << 51):
                    raise NotImplementedError("64 bit integer images are supported for values below 2**51 (they are held as float64)")
            self._copy_in(self.img[b], image)
            return
        if image.dtype != self.dtype:
            raise TypeError(f"engine built for {self.dtype} images, got {getattr(image, 'dtype', type(image))}")
        if tuple(image.shape) != (self.plan.H, self.plan.W):
            raise ValueError(f"expected an image of shape {(self.plan.H, self.plan.W)}, got {tuple(image.shape)}")
        if image.dtype != self.store_dtype:
            if image.dtype.itemsize == 8 and image.dtype.kind in "iu" and image.size:
                if max(abs(int(image.max())), abs(int(image.min()))) >= 1 << 51:
                    raise NotImplementedError("64 bit integer images are supported for values below 2**51 (they are held as float64)")
            image = image.astype(self.store_dtype)
        self._copy_in(self.img[b], torch.from_numpy(np.ascontiguousarray(image)))

    # ------------------------------------------------------------------ launches
    def launch_octaves(self, zero=None):
        """zero: a tensor view of accumulator words the launch's first workgroup resets (see run)."""
        p = self.plan
        if p.n_levels == 0 or self.exact_single:
            return
        self._mm_clean = False
        nat.check(self.lib.wb_octaves_launch_z(nat.stream_ptr(), nat.ptr(self.img), self.wb_dtype, self.batch, p.H, p.W,
                                               p.H * p.W, nat.ptr(self.oct), p.oct_total, self._oct_off, p.n_oct,
                                               nat.ptr(self.minmax), nat.ptr(zero), 0 if zero is None else zero.numel()),
                  "wb_octaves_launch")

    def ensure_clean_keys(self):
        """A captured step holds no memset (run): before it is replayed the octaves' keys must be zero -- they are, unless
        something else (a pyramid for a caller, a bare octave launch) has used this engine since the last step."""
        if not self._mm_clean and not self.exact_single:
            self.ctrl[: self._mm_words].zero_()
            self._mm_clean = True
        elif _CHECK_KEYS and not self.exact_single:
            # WB_CHECK_KEYS=1 (the test suite's interleaving test turns it on): the flag is a host-side promise every user
            # of the engine's octaves has to keep -- read the keys back and see that it was kept
            import torch
            if not torch.cuda.is_current_stream_capturing() and bool(self.ctrl[: self._mm_words].any().item()):
                raise RuntimeError("the octaves' (min, max) keys are flagged clean but are not zero: some path used the "
                                   "engine's octaves without going through launch_octaves / clearing _mm_clean")

    def ranks_for(self, dm):
        """True when the fused detection path applies: grad_hist channels written straight as threshold ranks of
        cascade `dm` (float32 channels never reach HBM)."""
        return dm is not None and dm.rank_dtype is not None and self.spec.key == "grad_hist" and not _NO_RANKS

    def launch_channels(self, rank_dm=None, floats=True):
        """The channel pyramid of every resident image.  rank_dm: also (floats=False: only) write the channels as
        WB_DTYPE_RANK8 bytes for that cascade into self.rank."""
        import torch
        p = self.plan
        if p.n_levels == 0:
            return
        wide = rank_dm is not None and rank_dm.rank_dtype == nat.WB_DTYPE_RANK16
        if rank_dm is not None and (self.rank is None or self._rank_wide != wide):
            # one byte per value, or two (WB_DTYPE_RANK16); 16 spare elements for the cascade's 16-byte group loads
            tdt = torch.int16 if wide else torch.uint8
            self._rank_flat = torch.zeros(self.batch * self.chn_stride + 16, dtype=tdt, device=self.dev)
            self.rank = self._rank_flat[: self.batch * self.chn_stride].view(self.batch, self.chn_stride)
            self._rank_wide = wide
            self.generation += 1          # (captured graphs address the old buffer)
            for stt in self._casc.values():
                stt.pop("graph", None)
                stt.pop("step", None)
        nat.check(self.lib.wb_channels_launch_x(nat.stream_ptr(), nat.ptr(self.img), p.H * p.W, nat.ptr(self.oct),
                                                p.oct_total, self.wb_dtype, self.batch, nat.ptr(self.levels),
                                                p.n_levels, nat.ptr(self.chan_tiles), self.n_chan_tiles,
                                                nat.ptr(self.minmax), max(p.n_oct, 1), nat.ptr(self.taps),
                                                self.spec.func_id, p.shrink, p.smooth,
                                                self.cs_sn.ctypes.data_as(C.POINTER(C.c_double)),
                                                nat.ptr(self.chn if floats or rank_dm is None else None), self.chn_stride,
                                                rank_dm.handle if rank_dm is not None else None,
                                                nat.ptr(self.rank if rank_dm is not None else None), self.chn_stride,
                                                nat.ptr(self.chan_patches),
                                                rank_dm.rank_dtype if rank_dm is not None else nat.WB_DTYPE_RANK8),
                  "wb_channels_launch")
        self.rank_owner = rank_dm.rank_key if rank_dm is not None else None      # whose ranks self.rank holds (None: stale)

    def launch_level(self, l):
        """The float/uint8 channels of ONE level of every resident image (after launch_octaves): what a lazy
        consumer of the pyramid asks for, level by level (reference channels.py:125-146 computes a level only
        when the generator is advanced to it)."""
        import torch
        p = self.plan
        if self._level_tiles is None:
            tiles = p.chan_tiles()
            self._level_tiles = []
            for k in range(p.n_levels):
                sel = np.ascontiguousarray(tiles[tiles["level"] == k])
                self._level_tiles.append((int(sel.size), torch.from_numpy(sel.view(np.uint8).copy()).to(self.dev) if sel.size else None,
                                          self._tile_patches(sel)))
        n, tiles_d, patches_d = self._level_tiles[l]
        if n == 0:
            return
        nat.check(self.lib.wb_channels_launch_x(nat.stream_ptr(), nat.ptr(self.img), p.H * p.W, nat.ptr(self.oct),
                                                p.oct_total, self.wb_dtype, self.batch, nat.ptr(self.levels),
                                                p.n_levels, nat.ptr(tiles_d), n,
                                                nat.ptr(self.minmax), max(p.n_oct, 1), nat.ptr(self.taps),
                                                self.spec.func_id, p.shrink, p.smooth,
                                                self.cs_sn.ctypes.data_as(C.POINTER(C.c_double)),
                                                nat.ptr(self.chn), self.chn_stride, None, None, 0, nat.ptr(patches_d),
                                                nat.WB_DTYPE_RANK8),
                  "wb_channels_launch")


    # ------------------------------------------------------------------ around a caller's channel function
    def resize_level(self, l):
        """Level l's resized image of resident image 0, cast back to the image dtype, as a host ndarray [nh, nw]
        (reference channels.py:132) -- after launch_octaves."""
        import torch
        rec = np.ascontiguousarray(self.level_np[l:l + 1])
        lv = rec[0]
        nh, nw = int(lv["nh"]), int(lv["nw"])
        if self._mm_host is None or self._mm_host[0] != self.epoch:
            self._mm_host = (self.epoch, self.minmax[0].cpu().numpy().copy())       # (one read-back per image)
        mm = np.ascontiguousarray(self._mm_host[1][int(lv["oct"])])
        out = torch.empty((nh, nw), dtype=self.tdtype, device=self.dev)
        nat.check(self.lib.wb_resize_level_launch(nat.stream_ptr(), nat.ptr(self.img), nat.ptr(self.oct), self.wb_dtype,
                                                  rec.ctypes.data_as(C.c_void_p),
                                                  mm.ctypes.data_as(C.c_void_p), nat.ptr(self.taps), nat.ptr(out)),
                  "wb_resize_level_launch")
        return out.cpu().numpy().astype(self.dtype, copy=False)

    def pool_smooth(self, chns):
        """avg_pool_2 (shrink 2) and smooth_image_3d (smooth 1) of a host array [H, W, C] uint8 / float32 -> host array
        (reference channels.py:138-142)."""
        import torch
        p = self.plan
        H, W, Cn = chns.shape
        code = nat.WB_DTYPE_U8 if chns.dtype == np.uint8 else nat.WB_DTYPE_F32
        d = torch.from_numpy(np.ascontiguousarray(chns)).to(self.dev)
        oh, ow = (H // 2, W // 2) if p.shrink == 2 else (H, W)
        out = torch.empty((oh, ow, Cn), dtype=d.dtype, device=self.dev)
        tmp = torch.empty_like(out) if (p.shrink == 2 and p.smooth) else None
        if H and W and Cn:
            nat.check(self.lib.wb_pool_smooth_launch(nat.stream_ptr(), nat.ptr(d), code, H, W, Cn, p.shrink, p.smooth,
                                                     nat.ptr(tmp), nat.ptr(out)), "wb_pool_smooth_launch")
        return out.cpu().numpy()

    def _casc_state(self, dm):
        import torch
        key = id(dm)
        stt = self._casc.get(key)
        if stt is None:
            tiles = self.plan.casc_tiles(dm.m, dm.n, dm.tile_rows, dm.tile_cols)
            T1 = max(dm.n_stages, 1)
            alive = self._alive_view(T1)                # (may re-allocate the control block and drop other states)
            stt = dict(
                dm=dm, n_tiles=int(tiles.size),
                tiles=torch.from_numpy(tiles.view(np.uint8).copy()).to(self.dev) if tiles.size else None,
                alive=alive)
            if len(self._casc) >= 4:         # a few cascades resident per engine (waldboost.detect scans several models)
                self._casc.pop(next(iter(self._casc)))
            self._casc[key] = stt
        return stt

    def launch_cascade(self, dm, ranks=False, stats=True, zero_keys=False):
        """The cascade scan, adding into the detection counters and (stats) alive[B, L, T] -- reset_step first.
        ranks=True scans self.rank (written for `dm` by launch_channels) instead of the channel buffer.
        zero_keys: the launch's first workgroup also resets the octaves' (min, max) keys for the next step (see run)."""
        stt = self._casc_state(dm)
        if stt["n_tiles"] == 0:
            return stt
        if ranks and self.rank_owner is not dm.rank_key:
            raise RuntimeError("the rank buffer does not hold this cascade's ranks (launch_channels(rank_dm=...) first)")
        dm.note_scan(dm.rank_dtype if ranks else self.spec.wb_dtype)
        zk = self.ctrl[: self._mm_words] if zero_keys else None
        nat.check(self.lib.wb_cascade_launch_z(nat.stream_ptr(), dm.handle, nat.ptr(self.rank if ranks else self.chn),
                                               dm.rank_dtype if ranks else self.spec.wb_dtype,
                                               self.chn_stride,
                                               self.batch, nat.ptr(self.levels), self.plan.n_levels,
                                               nat.ptr(stt["tiles"]), stt["n_tiles"],
                                               nat.ptr(self.detb.recs), nat.ptr(self.detb.counts), self.detb.cap,
                                               nat.ptr(stt["alive"] if stats else None), nat.ptr(zk),
                                               0 if zk is None else zk.numel()),
                  "wb_cascade_launch")
        if zero_keys:
            self._mm_clean = True
        return stt

    def run_cascade(self, dm, ranks=False):
        """Reset the counters and statistics, then scan every level of every image with cascade `dm`."""
        stt = self._casc_state(dm)
        self.reset_step(stt, octaves=False)
        stt["ranks"] = ranks                       # (a re-scan after a buffer overflow repeats the same form)
        return self.launch_cascade(dm, ranks=ranks)

    def run_channels(self, rank_dm=None, floats=True):
        """Octaves and the channel pyramid of the resident images (no cascade)."""
        self.reset_step(None, octaves=True)
        self.launch_octaves()
        self.launch_channels(rank_dm, floats)

    def run(self, dm):
        """octaves -> channels -> cascade for one model: the detection path (reference model.py:149-179): one
        memset and three kernels.  When the cascade has rank tables the channels go to HBM as ranks only."""
        fused = self.ranks_for(dm)
        stt = self._casc_state(dm)
        # No memset launch: the octave kernel's first workgroup resets what the CASCADE accumulates into (shard counters,
        # alive[B, L, T] -- nothing touches them before it), the cascade's first workgroup resets what the next step's
        # OCTAVE kernel accumulates into (the (min, max) keys -- the channel kernel has finished with them).  The keys
        # must be zero on entry: ensure_clean_keys (one memset after anything else has used the engine's octaves).
        if self.exact_single or _NO_FUSED_RESET or stt["n_tiles"] == 0 or self.plan.n_levels == 0:
            self.reset_step(stt, octaves=True)
            self.launch_octaves()
            self.launch_channels(dm if fused else None, floats=not fused)
            stt["ranks"] = fused
            return self.launch_cascade(dm, ranks=fused)
        import torch
        if not torch.cuda.is_current_stream_capturing():
            self.ensure_clean_keys()
        elif not self._mm_clean:
            raise RuntimeError("capture a step only after an eager one (PyramidEngine.capture does): the octaves' keys must be clean")
        o = self._mm_words
        self.launch_octaves(zero=self.ctrl[o: o + nat.WB_DET_SHARDS + stt["alive"].numel()])
        self.launch_channels(dm if fused else None, floats=not fused)
        stt["ranks"] = fused
        return self.launch_cascade(dm, ranks=fused, zero_keys=True)

    # ------------------------------------------------------------------ hipGraph
    def capture(self, dm):
        """Capture octaves -> channels -> cascade into one hipGraph (torch.cuda.CUDAGraph on ROCm).
        Returns a CapturedStep; ``.replay()`` re-runs the whole pipeline on the resident images (and raises once the
        engine has re-allocated a buffer the graph addresses)."""
        import torch
        self._casc_state(dm)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.run(dm)                      # warm-up outside capture (module load, attributes)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with capturing(g):
            self.run(dm)
        return CapturedStep(self, g)

    # ------------------------------------------------------------------ results
    def ensure_capacity(self, dm):
        """Re-run the cascade with a larger buffer if a shard of the detection buffer overflowed
        in the last scan.  Returns the number of detections."""
        while True:
            need = self.detb.max_count()
            if need <= self.detb.cap:
                return int(self.detb.counts.sum().item())
            self.det_capacity = (int(need * 1.5) + 16) * nat.WB_DET_SHARDS
            self._alloc_det()
            self.run_cascade(dm, ranks=self._casc_state(dm).get("ranks", False))

    def shard_counts(self, dm):
        """Host copy of the shard counters after the last scan; re-runs the cascade with a larger buffer if a
        shard overflowed."""
        while True:
            counts = self.detb.counts.cpu().numpy().astype(np.int64)
            if counts.max(initial=0) <= self.detb.cap:
                return counts
            self.det_capacity = (int(counts.max() * 1.5) + 16) * nat.WB_DET_SHARDS
            self._alloc_det()
            self.run_cascade(dm, ranks=self._casc_state(dm).get("ranks", False))

    def pack(self, out=None):
        """Pack the valid records of all shards behind a 4-word header (wb_det_pack_launch) into self.packed --
        the form a collective or a host read-back takes; no synchronisation.  out: an int32 [1 + rows, 4] device
        tensor (16-byte aligned, e.g. a rank's slot of a collective's send buffer) to pack into instead -- the header
        then says how many of the valid records fitted its `rows`."""
        import torch
        if out is not None:
            nat.check(self.lib.wb_det_pack_launch(nat.stream_ptr(), nat.ptr(self.detb.recs), nat.ptr(self.detb.counts),
                                                  self.detb.cap, nat.ptr(out), out.shape[0] - 1), "wb_det_pack_launch")
            return out
        if self.packed is None:
            self.packed = torch.empty((1 + self.detb.NS * self.detb.cap, 4), dtype=torch.int32, device=self.dev)
        nat.check(self.lib.wb_det_pack_launch(nat.stream_ptr(), nat.ptr(self.detb.recs), nat.ptr(self.detb.counts),
                                              self.detb.cap, nat.ptr(self.packed), self.packed.shape[0] - 1),
                  "wb_det_pack_launch")
        return self.packed

    def fetch(self, dm, stt, limit=None):
        """Everything the host needs from the last scan in ONE synchronisation: the shard contents packed on
        the device (wb_det_pack_launch), then asynchronous copies of the header + first records and of
        alive[B, L, T] into page-locked memory, one event wait.  Grows the buffer and scans again if a shard
        overflowed; a second copy only when there are more than _FETCH_ROWS detections.
        Returns (records int32 [n, 4] in shard order, alive int64 [B, L, T]).
        limit: with more than `limit` detections the records stay on the device (self.packed[1:1 + n]; the caller orders
        them there) and their count is returned in place of the array."""
        import torch
        T = dm.n_stages
        while True:
            if self._h_packed is None:
                self._h_packed = torch.empty((1 + self._FETCH_ROWS, 4), dtype=torch.int32).pin_memory()
                self._fetch_ev = torch.cuda.Event()
            if self._h_alive is None or self._h_alive.shape != stt["alive"].shape:
                self._h_alive = torch.empty(stt["alive"].shape, dtype=torch.int32).pin_memory()
            self.pack()
            rows = min(self._h_packed.shape[0], self.packed.shape[0])
            self._h_packed[:rows].copy_(self.packed[:rows], non_blocking=True)
            self._h_alive.copy_(stt["alive"], non_blocking=True)
            self._fetch_ev.record()
            self._fetch_ev.synchronize()
            total, worst = int(self._h_packed[0, 0]), int(self._h_packed[0, 1])
            if worst <= self.detb.cap:
                break
            self.det_capacity = (int(worst * 1.5) + 16) * nat.WB_DET_SHARDS
            self._alloc_det()
            stt = self.run_cascade(dm, ranks=stt.get("ranks", False))
        alive = self._h_alive.numpy()[:, :, :T].astype(np.int64)
        if limit is not None and total > limit:
            return total, alive
        recs = self._h_packed[1:1 + min(total, rows - 1)].numpy().copy()
        if total > rows - 1:
            recs = np.concatenate([recs, self.packed[rows:1 + total].cpu().numpy()])
        return recs, alive

    def _final_ready(self):
        """Whether Model.detect's one-copy read-back form applies to this engine (one image, a pyramid within the sort
        key's 10 / 14 / 14-bit fields); allocates its buffers on first use."""
        import torch
        p = self.plan
        if self._final_dims is None:
            ok = self.batch == 1 and 0 < p.n_levels <= 1024
            mu = max((int(lv["u"]) for lv in p.levels), default=0)
            mv = max((int(lv["v"]) for lv in p.levels), default=0)
            self._final_dims = (ok and mu <= 16384 and mv <= 16384, mu, mv)
        if not self._final_dims[0]:
            return False
        if self._inv_scales_d is None:
            self._inv_scales_d = torch.from_numpy(self.inv_scales()).to(self.dev)
            self._fetch_ev = self._fetch_ev or torch.cuda.Event()
        return True

    def _final_enqueue(self, dm, stt):
        """wb_det_finish_sorted_launch + the ONE read-back copy into page-locked memory (no synchronisation).  The buffers
        belong to the cascade's scan state: several cascades can be scanned back to back on one engine (waldboost.detect)
        and read back with ONE wait -- the shared detection buffer is free again as soon as this launch has run."""
        import torch
        n_alive = stt["alive"].numel()
        if "final" not in stt or stt["final_alive_words"] != n_alive:
            # header | keys | boxes | scores | alive[B, L, T] (the kernel copies the statistics behind the scores: one
            # read-back copy instead of two)
            P = self._FETCH_ROWS
            nbytes = 16 + P * 28 + 4 * n_alive
            stt["final"] = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
            stt["h_final"] = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
            stt["final_alive_words"] = n_alive
            h = stt["h_final"].numpy()
            stt["h_final_views"] = (h[:16].view(np.int32), h[16:16 + 8 * P].view(np.uint64),
                                    h[16 + 8 * P:16 + 24 * P].view(np.float32).reshape(P, 4), h[16 + 24 * P:16 + 28 * P].view(np.float32))
            stt["h_alive"] = h[16 + 28 * P:].view(np.int32).reshape(tuple(stt["alive"].shape))
        # (ordered on the device: one workgroup sorts the keys in LDS and writes keys, boxes and scores in the reference's
        # order whenever they number at most 4096 -- header[3] says whether it did)
        nat.check(self.lib.wb_det_finish_sorted_launch(nat.stream_ptr(), nat.ptr(self.detb.recs), nat.ptr(self.detb.counts),
                                                       self.detb.cap, nat.ptr(self._inv_scales_d), self.plan.n_levels,
                                                       self._final_dims[1], self._final_dims[2], dm.m, dm.n,
                                                       nat.ptr(stt["final"]), self._FETCH_ROWS, nat.ptr(stt["alive"]), n_alive),
                  "wb_det_finish_sorted_launch")
        stt["h_final"].copy_(stt["final"], non_blocking=True)

    def fetch_final(self, dm, stt, enqueued=False, stream=None):
        """fetch() for Model.detect on ONE image: wb_det_finish_launch leaves sort keys, boxes and scores of all
        valid records behind one header; they come back with ONE copy and ONE event wait together with alive[B, L, T].
        Returns (keys uint64 [n] (level << 54 | r << 40 | c << 26 | position), boxes float32 [rows, 4], scores
        float32 [rows], alive int64 [B, L, T], ordered) -- ordered: the device sorted them (keys ascending, boxes[i] /
        scores[i] the i-th detection: at most 4096 of them); else keys unsorted, boxes / scores indexed by a key's
        position.  The arrays are views of the page-locked read-back buffer: copy what is kept.  None when this form does not apply (a batch, a pyramid beyond the key's bit fields, more than _FETCH_ROWS
        detections): use fetch() then.  Grows the detection buffer and scans again if a shard overflowed.
        enqueued: the launch and the copies are already in the stream (detect_run's graph replay).
        stream: that stream, when it is not the current one -- then only the wait happens here, and False is returned
        if more than a wait is needed (the caller comes back on that stream)."""
        if not self._final_ready():
            return None
        P, T = self._FETCH_ROWS, dm.n_stages
        while True:
            if not enqueued:
                self._final_enqueue(dm, stt)
            enqueued = False
            if stream is None:
                self._fetch_ev.record()
            else:
                self._fetch_ev.record(stream)
            self._fetch_ev.synchronize()
            hdr, keys, boxes, scores = stt["h_final_views"]
            total, worst = int(hdr[0]), int(hdr[1])
            if worst <= self.detb.cap:
                break
            if stream is not None:
                return False
            self.det_capacity = (int(worst * 1.5) + 16) * nat.WB_DET_SHARDS
            self._alloc_det()
            stt = self.run_cascade(dm, ranks=stt.get("ranks", False))
        if total > P:
            return None
        alive = stt["h_alive"][:, :, :T].astype(np.int64)
        return keys[:total], boxes, scores, alive, bool(hdr[3])

    _ORDER_ROWS = 4096               # per image: what wb_det_order_batch_launch orders (more: the caller's other path)

    def _order_buffers(self):
        """Buffers of the batch's ordered read-back (scratch, output, its page-locked copy), on first use; whether the
        form applies to this pyramid (the sort key's 10 / 14 / 14-bit fields)."""
        import torch
        p = self.plan
        if self._order is None:
            mu = max((int(lv["u"]) for lv in p.levels), default=0)
            mv = max((int(lv["v"]) for lv in p.levels), default=0)
            fits = 0 < p.n_levels <= 1024 and mu <= 16384 and mv <= 16384
            od = self._order = dict(fits=fits, mu=mu, mv=mv)
            if fits:
                P, B = self._ORDER_ROWS, self.batch
                blk = 16 + 28 * P
                od["scratch"] = torch.empty(B * (256 + 16 * P), dtype=torch.uint8, device=self.dev)
                od["out"] = torch.empty(16 + B * blk, dtype=torch.uint8, device=self.dev)
                od["h_out"] = torch.empty(16 + B * blk, dtype=torch.uint8).pin_memory()
                od["ev"] = torch.cuda.Event()
                h = od["h_out"].numpy()
                od["info"] = h[:16].view(np.int32)
                od["views"] = []
                for b in range(B):
                    o = 16 + b * blk
                    od["views"].append((h[o:o + 16].view(np.int32), h[o + 16:o + 16 + 8 * P].view(np.uint64),
                                        h[o + 16 + 8 * P:o + 16 + 24 * P].view(np.float32).reshape(P, 4), h[o + 16 + 24 * P:o + blk].view(np.float32)))
                if self._inv_scales_d is None:
                    self._inv_scales_d = torch.from_numpy(self.inv_scales()).to(self.dev)
        return self._order["fits"]

    def order_batch_enqueue(self, dm, stt):
        """wb_det_order_batch_launch on the last scan's detections + the read-back copies (ordered results, alive[B, L, T])
        into page-locked memory + an event, all in the current stream, no synchronisation: what fetch_ordered_batch
        waits for.  Enqueued right behind the step, the results are on the host by the time they are asked for.
        Returns False when the form does not apply to this pyramid."""
        import torch
        if not self._order_buffers():
            return False
        od, p = self._order, self.plan
        if self._h_alive is None or self._h_alive.shape != stt["alive"].shape:
            self._h_alive = torch.empty(stt["alive"].shape, dtype=torch.int32).pin_memory()
        nat.check(self.lib.wb_det_order_batch_launch(nat.stream_ptr(), nat.ptr(self.detb.recs), nat.ptr(self.detb.counts),
                                                     self.detb.cap, self.batch, nat.ptr(self._inv_scales_d), p.n_levels,
                                                     od["mu"], od["mv"], dm.m, dm.n, nat.ptr(od["scratch"]),
                                                     od["scratch"].numel(), nat.ptr(od["out"]), self._ORDER_ROWS),
                  "wb_det_order_batch_launch")
        od["h_out"].copy_(od["out"], non_blocking=True)
        self._h_alive.copy_(stt["alive"], non_blocking=True)
        od["ev"].record()
        return True

    def fetch_ordered_batch(self, dm, stt, enqueued=False):
        """fetch() for a batch whose results are wanted image by image in the reference's order (Model.detect_stream's
        batches): wb_det_order_batch_launch splits the shards' records by image and orders every image's keys, boxes
        and scores on the device; they come back with ONE copy and ONE event wait together with alive[B, L, T].
        enqueued: order_batch_enqueue has run behind the scan already (only the wait happens here).
        Grows the detection buffer and scans again if a shard overflowed.
        Returns ([(keys uint64 [n_b], boxes float32 [n_b, 4], scores float32 [n_b]) per image], alive int64 [B, L, T])
        -- views of the page-locked read-back buffer: copy what is kept -- or None when this form does not apply (a
        pyramid beyond the sort key's bit fields, an image with more than _ORDER_ROWS detections): use fetch() then."""
        T = dm.n_stages
        while True:
            if not enqueued and not self.order_batch_enqueue(dm, stt):
                return None
            enqueued = False
            od = self._order
            od["ev"].synchronize()
            worst = int(od["info"][1])
            if worst <= self.detb.cap:
                break
            self.det_capacity = (int(worst * 1.5) + 16) * nat.WB_DET_SHARDS
            self._alloc_det()
            stt = self.run_cascade(dm, ranks=stt.get("ranks", False))
        out = []
        for hdr, keys, boxes, scores in od["views"]:
            n_b = int(hdr[0])
            if int(hdr[1]) > self._ORDER_ROWS or int(hdr[3]) != 1:
                return None
            out.append((keys[:n_b], boxes[:n_b], scores[:n_b]))
        return out, self._h_alive.numpy()[:, :, :T].astype(np.int64)

    def live_check(self, dm):
        """A cascade whose specialised kernel has just been built, on the image(s) resident in this engine: the whole step
        with the specialised kernel, then the cascade alone again with the generic kernel on the same channels; per-stage
        alive counts and the ordered detection records must be identical.  If they are not, the specialised kernel is
        switched off for this cascade for good (a warning says so) -- the step's buffers hold the generic kernel's results
        either way.  Runs _LIVE_CHECKS times per cascade and tile kind, outside captures; costs one extra step each."""
        import torch
        dtype = dm.rank_dtype if self.ranks_for(dm) else self.spec.wb_dtype
        if dm.live_checks_left(dtype) <= 0 or dtype not in dm.specialized() or torch.cuda.is_current_stream_capturing():
            return
        left = dm.__dict__["_live_left"]
        while left.get(dtype, 0) > 0:
            left[dtype] -= 1
            stt = self.run(dm)                                       # specialised kernel
            if int(self.detb.counts.max().item()) > self.detb.cap:
                return                                                # (overflowing buffer: the caller grows it; check another time)
            spec = (sort_records(self.detb.compact()).clone(), stt["alive"].clone())
            dm.use_specialized(False)
            self.run_cascade(dm, ranks=stt["ranks"])                  # generic kernel, same channels
            gen = (sort_records(self.detb.compact()), stt["alive"])
            same = spec[0].shape == gen[0].shape and bool(torch.equal(spec[0], gen[0])) and bool(torch.equal(spec[1], gen[1]))
            if not same:
                left[dtype] = 0
                _log.warning("cascade of %d stages, depth %d: its model-specialised kernel disagrees with the generic kernel on "
                             "this image (%d against %d detections); the specialised kernel is switched off for this cascade",
                             dm.n_stages, dm.depth, int(spec[0].shape[0]), int(gen[0].shape[0]))
                return                                                # (stays off)
            dm.use_specialized(True)

    def detect_run(self, dm):
        """Model.detect's whole device sequence for the resident image -- one memset, octaves, channels, cascade,
        wb_det_finish_sorted_launch (which also carries alive[] behind the scores), the ONE read-back copy -- and its one synchronisation; from the second call with the
        same cascade on it is replayed as ONE hipGraph (one enqueue instead of seven, no gaps between the kernels).
        Returns what fetch_final returns, or None (then: run(dm) has happened, use fetch())."""
        return self.detect_collect(dm, self.detect_enqueue(dm))

    def detect_enqueue(self, dm):
        """detect_run up to, not including, its wait: everything is in the current stream when this returns (the image
        must stay resident until detect_collect, on the same stream, has run).  Returns a token for detect_collect."""
        import torch
        stt = self._casc_state(dm)
        if not self._final_ready():
            self.run(dm)
            return None
        g = stt.get("graph")
        if g is None and stt.get("detect_calls", 0) >= 1 and not _NO_DETECT_GRAPH:
            # (the first call ran eagerly: every lazily allocated buffer exists, the kernels are loaded)
            # a cascade that is scanned again is worth its specialised kernel -- built now, so that the graph holds it
            dm.note_scan(dm.rank_dtype if self.ranks_for(dm) else self.spec.wb_dtype, force=True)
            self.live_check(dm)               # (a kernel built just now meets its first real image: cross-checked before the capture)
            g = torch.cuda.CUDAGraph()
            self.ensure_clean_keys()          # (the captured step holds no memset: see run)
            torch.cuda.synchronize()
            with capturing(g):
                self.run(dm)
                self._final_enqueue(dm, stt)
            stt["graph"] = g
        stt["detect_calls"] = stt.get("detect_calls", 0) + 1
        if g is None:
            self.run(dm)
            self._final_enqueue(dm, stt)
            return stt
        self.ensure_clean_keys()
        g.replay()
        return stt

    def batch_enqueue(self, dm):
        """The step -- octaves, channels, cascade -- over the resident batch, without a wait: eager on an engine's first
        call, a replayed hipGraph afterwards (captured again when the engine re-allocated a buffer the graph addresses).
        Returns the scan state fetch() takes."""
        stt = self._casc_state(dm)
        step = stt.get("step")
        if step is not None and step.generation != self.generation:
            step = None
        if step is None and stt.get("batch_calls", 0) >= 1 and not _NO_DETECT_GRAPH:
            dm.note_scan(dm.rank_dtype if self.ranks_for(dm) else self.spec.wb_dtype, force=True)
            self.live_check(dm)
            step = stt["step"] = self.capture(dm)
        stt["batch_calls"] = stt.get("batch_calls", 0) + 1
        if step is None:
            self.run(dm)
        else:
            step.replay()
        return stt

    def detect_multi_run(self, dms, ranks):
        """waldboost.detect's whole device sequence for the resident image -- octaves, ONE channel pyramid (as ranks of
        dms[0]'s rank tables -- a rank group's union tables -- or as float32 channels), then per cascade of `dms` its
        scan, wb_det_finish_sorted_launch and the read-back copy -- with ONE wait at its end; from the second call with the
        same cascades on it is one hipGraph replay.  Returns [what fetch_final returns, per cascade] -- None in the place of
        a cascade whose results did not fit (an overflowing detection buffer, more detections than one read-back holds): the
        pyramid stays resident and the caller scans that cascade alone again --, or None when the form does not apply at all
        (more cascades than an engine keeps states for, a pyramid beyond the sort key's fields, a sequence that kept missing
        lately): the caller then builds the pyramid and scans model by model."""
        import torch
        if not self._final_ready() or not 0 < len(dms) <= 4 or len({id(d) for d in dms}) != len(dms):
            return None
        key = tuple(id(d) for d in dms) + (bool(ranks),)
        stts = [self._casc_state(d) for d in dms]            # (may grow the control block: before the generation is read)
        if any(self._casc.get(id(d)) is not t for d, t in zip(dms, stts)):
            return None
        st = self._multi.get(key)
        if st is None or st["generation"] != self.generation or any(a is not b for a, b in zip(st["stts"], stts)):
            if len(self._multi) >= 2:
                self._multi.pop(next(iter(self._multi)))
            st = self._multi[key] = dict(generation=self.generation, stts=stts, dms=list(dms), calls=0, graph=None, skip=0, fails=0)
        if st["skip"] > 0:                                    # (its results did not fit lately: do not scan twice per call)
            st["skip"] -= 1
            return None

        def enqueue():
            self.run_channels(rank_dm=dms[0] if ranks else None, floats=not ranks)
            for d in dms:
                self._final_enqueue(d, self.run_cascade(d, ranks=ranks))

        if st["graph"] is None and st["calls"] >= 1 and st["fails"] == 0 and not _NO_DETECT_GRAPH:
            # (captured after an eager call whose results fitted: a sequence that keeps missing is not worth a capture)
            for d in dms:
                d.note_scan(d.rank_dtype if ranks else self.spec.wb_dtype, force=True)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with capturing(g):
                enqueue()
            st["graph"] = g
        st["calls"] += 1
        if st["graph"] is None:
            enqueue()
        else:
            self._mm_clean = False                            # (the replay's octave launch leaves its keys behind)
            st["graph"].replay()
            self.rank_owner = dms[0].rank_key if ranks else None
        self._fetch_ev.record()
        self._fetch_ev.synchronize()
        out, missed = [], False
        for d, stt in zip(dms, stts):
            hdr, keys, boxes, scores = stt["h_final_views"]
            total, worst = int(hdr[0]), int(hdr[1])
            if worst > self.detb.cap or total > self._FETCH_ROWS:
                # this cascade's results did not fit (a shard overflowed, or more detections than one read-back holds):
                # None in its place -- the caller scans THAT cascade again on the pyramid this call left resident
                # (Model.scan_engine: run_cascade + fetch, which grows the buffer); the others keep what they have
                missed = True
                out.append(None)
            else:
                out.append((keys[:total], boxes, scores, stt["h_alive"][:, :, :d.n_stages].astype(np.int64), bool(hdr[3])))
        if missed:
            st["fails"] += 1                                  # (the whole sequence is tried again after 16, 32, 64 ... calls)
            st["skip"] = min(8 << st["fails"], 4096)
            # (the re-scans may reuse read-back buffers before the caller has collected the fitted results: hand out copies)
            out = [None if f is None else (f[0].copy(), np.array(f[1][:f[0].size]), np.array(f[2][:f[0].size]), f[3], f[4]) for f in out]
        else:
            st["fails"] = 0
        return out

    def detect_collect(self, dm, token, stream=None):
        """The wait and the read-back that end detect_run, for a token of detect_enqueue.
        stream: the stream detect_enqueue ran on, when that is not the current one."""
        import torch
        if stream is None:
            return None if token is None else self.fetch_final(dm, token, enqueued=True)
        if token is not None:
            fin = self.fetch_final(dm, token, enqueued=True, stream=stream)
            if fin is not False:
                return fin
        with torch.cuda.stream(stream):                       # (the rare ways out: launches and copies of their own)
            return None if token is None else self.fetch_final(dm, token, enqueued=False)

    def sorted_detections(self, n=None):
        """Detections ordered by (image, level, r, c) as an int32 [n, 4] tensor of WbDet records."""
        return sort_records(self.detb.compact())

    def boxes(self, det_sorted, dm):
        import torch
        n = det_sorted.shape[0]
        boxes = torch.empty((n, 4), dtype=torch.float32, device=self.dev)
        scores = torch.empty(n, dtype=torch.float32, device=self.dev)
        if n:
            inv = np.array([np.float32(1.0 / s) for s in self.plan.scales], np.float32)
            inv_d = torch.from_numpy(inv).to(self.dev)
            nat.check(self.lib.wb_boxes_launch(nat.stream_ptr(), nat.ptr(det_sorted), n, nat.ptr(inv_d), dm.m, dm.n,
                                               nat.ptr(boxes), nat.ptr(scores)), "wb_boxes_launch")
        return boxes, scores

    def inv_scales(self):
        """float32(1.0 / scale) per level: the factor get_boxes multiplies with (reference model.py:147)."""
        if getattr(self, "_inv_scales", None) is None:
            self._inv_scales = np.array([np.float32(1.0 / s) for s in self.plan.scales], np.float32)
        return self._inv_scales

    def level_tensor(self, b, l):
        """Channels of level l of image b as a device view [u,v,C] into the pyramid buffer."""
        lv = self.plan.levels[l]
        off = int(self.level_np[l]["chn_off"])
        u, v, C = lv["u"], lv["v"], self.spec.n_channels
        return self.chn[b, off:off + u * v * C].view(u, v, C)

    def read_rank_level(self, b, l):
        """Ranks of level l of image b as a uint8 ndarray [u,v,4] (after launch_channels(rank_dm=...))."""
        lv = self.plan.levels[l]
        off = int(self.level_np[l]["chn_off"])
        u, v = lv["u"], lv["v"]
        r = self.rank[b, off:off + u * v * 4].reshape(u, v, 4).cpu().numpy()
        return r.view(np.uint16) if r.dtype == np.int16 else r

    def read_level(self, b, l):
        """Channels of level l of image b as a fresh HWC ndarray [u,v,C] of the channel function's dtype."""
        lv = self.plan.levels[l]
        off = int(self.level_np[l]["chn_off"])
        u, v, C = lv["u"], lv["v"], self.spec.n_channels
        return self.chn[b, off:off + u * v * C].reshape(u, v, C).cpu().numpy()


def nat_f32_key(f):
    b = np.array([f], np.float32).view(np.uint32)[0]
    return np.uint32(~b & 0xFFFFFFFF) if (b & 0x80000000) else np.uint32(b | 0x80000000)


_ENGINES = {}


def get_engine(H, W, dtype, shrink, n_per_oct, smooth, batch=1, exact_single=False, channels=None):
    """Small cache of engines keyed by configuration (buffers are reused across calls)."""
    import torch
    channels = channels if channels is not None else SPECS["grad_hist"]
    key = (int(H), int(W), np.dtype(dtype).str, int(shrink), int(n_per_oct), int(smooth), int(batch),
           bool(exact_single), channels.key, torch.cuda.current_device() if torch.cuda.is_available() else -1)
    e = _ENGINES.get(key)
    if e is None:
        if len(_ENGINES) >= 4:
            _ENGINES.pop(next(iter(_ENGINES)))
        e = PyramidEngine(H, W, dtype, shrink, n_per_oct, smooth, batch, exact_single, channels=channels)
        _ENGINES[key] = e
    return e
