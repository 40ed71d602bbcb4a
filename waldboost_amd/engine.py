"""Device-side execution of the hot path: buffers, launches, hipGraph capture.

``PyramidEngine`` owns the HBM-resident state of one (image shape, dtype, channel_opts,
batch) configuration and drives the three kernel groups through the C ABI:

    octaves (wb_octaves_launch) -> channels (wb_channels_launch) -> cascade (wb_cascade_launch)

torch is used only for device memory, streams and graph capture; all compute is in
csrc/*.hip.  Nothing here falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _native as nat
from .chanfunc import SPECS
from .plan import PyramidPlan, N_CHANNELS

_TORCH_DT = {}


def _torch_dtype(np_dtype):
    import torch
    if not _TORCH_DT:
        _TORCH_DT.update({np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32})
    try:
        return _TORCH_DT[np.dtype(np_dtype)]
    except KeyError:
        raise NotImplementedError(
            f"image dtype {np.dtype(np_dtype)} has no HIP kernel (uint8 and float32 are supported)") from None


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
        info = nat.WbModelInfo()
        nat.check(lib.wb_model_info(h, C.byref(info)), "wb_model_info")
        self.n_stages, self.depth = info.n_stages, info.depth
        self.m, self.n, self.C = info.m, info.n, info.C
        self.tile_rows, self.tile_cols, self.lds_bytes = info.tile_rows, info.tile_cols, info.lds_bytes

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.wb_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DetBuffer:
    """Sharded append buffer of WbDet records (include/waldboost_hip.h: WB_DET_SHARDS regions of
    `cap` records, one counter each).  One contiguous int32 [NS/4 + NS*cap, 4] block: the first
    NS/4 rows are the NS counters, the records follow -- so the whole thing can be handed to a
    collective as is."""

    def __init__(self, cap, dev):
        import torch
        self.NS = nat.WB_DET_SHARDS
        self.cap = int(cap)
        self.buf = torch.zeros((self.NS // 4 + self.NS * self.cap, 4), dtype=torch.int32, device=dev)
        self.counts = self.buf[: self.NS // 4].view(-1)
        self.recs = self.buf[self.NS // 4:]

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
        ar = torch.arange(self.cap, device=self.buf.device, dtype=torch.int32)
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


class PyramidEngine:
    def __init__(self, H, W, dtype, shrink, n_per_oct, smooth, batch=1, exact_single=False, det_capacity=1 << 16,
                 channels=None):
        import torch
        self.lib = nat.load()
        self.dev = nat.require_gpu()
        self.dtype = np.dtype(dtype)
        self.spec = channels if channels is not None else SPECS["grad_hist"]
        if self.spec.dtype == np.uint8 and self.dtype != np.uint8:
            raise NotImplementedError(f"{self.spec.key} takes 8 bit images (uint8), got {self.dtype}")
        self.tdtype = _torch_dtype(dtype)
        self.wb_dtype = nat.WB_DTYPE_U8 if self.dtype == np.uint8 else nat.WB_DTYPE_F32
        self.batch = int(batch)
        self.plan = PyramidPlan(H, W, shrink, n_per_oct, smooth, exact_single=exact_single,
                                n_chn=self.spec.n_channels, chn_bytes=self.spec.dtype.itemsize)
        self.exact_single = exact_single
        p = self.plan
        dev = self.dev
        # flat allocations with 16 spare elements: the channel kernel fetches source rows with
        # 4-byte-aligned dword loads that may touch a few bytes past the last row
        self._img_flat = torch.zeros(self.batch * p.H * p.W + 16, dtype=self.tdtype, device=dev)
        self.img = self._img_flat[: self.batch * p.H * p.W].view(self.batch, p.H, p.W)
        self._oct_flat = torch.zeros(self.batch * p.oct_total + 16, dtype=self.tdtype, device=dev)
        self.oct = self._oct_flat[: self.batch * p.oct_total].view(self.batch, p.oct_total)
        self.minmax = torch.zeros((self.batch, max(p.n_oct, 1), 2), dtype=torch.int32, device=dev)
        table, total = p.level_table()
        self.chn_stride = int(total)
        self.level_np = table
        self.levels = torch.from_numpy(table.view(np.uint8).copy()).to(dev) if p.n_levels else None
        taps, _ = p.tap_table()
        self.taps = torch.from_numpy(taps.view(np.uint8).copy()).to(dev)
        tiles = p.chan_tiles()
        self.n_chan_tiles = int(tiles.size)
        self.chan_tiles = torch.from_numpy(tiles.view(np.uint8).copy()).to(dev) if tiles.size else None
        # (16 spare elements: the cascade's uint8 tile load fetches 16-byte groups that may run past a level row)
        self._chn_flat = torch.zeros(self.batch * self.chn_stride + 16, dtype=_torch_dtype(self.spec.dtype), device=dev)
        self.chn = self._chn_flat[: self.batch * self.chn_stride].view(self.batch, self.chn_stride)
        self.cs_sn = orientation_constants()
        self._oct_off = (C.c_int64 * max(p.n_oct, 1))(*[int(x) for x in p.oct_off[:max(p.n_oct, 1)]])
        self.det_capacity = int(det_capacity)
        self._alloc_det()
        self.alive = None
        self._casc = {}
        if exact_single:
            # grad_hist on a bare image: no resize happens, so the clip range is (-inf, +inf)
            lo = np.array([nat_f32_key(-np.inf)], np.uint32).view(np.int32)[0]
            hi = np.array([nat_f32_key(np.inf)], np.uint32).view(np.int32)[0]
            if self.wb_dtype == nat.WB_DTYPE_U8:
                lo, hi = 0, 255
            lo = np.array([~np.uint32(np.array([lo]).astype(np.int64)[0] & 0xFFFFFFFF)], np.uint32).view(np.int32)[0]
            self.minmax[:, :, 0] = int(lo)   # word 0 stores max(~key), see csrc/wb_octaves.hip
            self.minmax[:, :, 1] = int(hi)

    def _alloc_det(self):
        """Sharded detection buffer; det_capacity is the total record capacity, split evenly over
        the shards."""
        cap = max(16, -(-self.det_capacity // nat.WB_DET_SHARDS))
        self.det_capacity = cap * nat.WB_DET_SHARDS
        self.detb = DetBuffer(cap, self.dev)
        self.det_buf = self.detb.buf

    # ------------------------------------------------------------------ input
    def load_images(self, images):
        """images: ndarray / tensor [B,H,W] (or [H,W]) of the engine's dtype."""
        import torch
        if isinstance(images, np.ndarray):
            if images.dtype != self.dtype:
                raise TypeError(f"engine built for {self.dtype} images, got {images.dtype}")
            t = torch.from_numpy(np.ascontiguousarray(images))
        else:
            t = images
        if t.dim() == 2:
            t = t[None]
        if tuple(t.shape) != (self.batch, self.plan.H, self.plan.W):
            raise ValueError(f"expected images of shape {(self.batch, self.plan.H, self.plan.W)}, got {tuple(t.shape)}")
        self.img.copy_(t, non_blocking=True)

    # ------------------------------------------------------------------ launches
    def launch_octaves(self):
        p = self.plan
        if p.n_levels == 0 or self.exact_single:
            return
        nat.check(self.lib.wb_octaves_launch(nat.stream_ptr(), nat.ptr(self.img), self.wb_dtype, self.batch, p.H, p.W,
                                             p.H * p.W, nat.ptr(self.oct), p.oct_total, self._oct_off, p.n_oct,
                                             nat.ptr(self.minmax)), "wb_octaves_launch")

    def launch_channels(self):
        p = self.plan
        if p.n_levels == 0:
            return
        nat.check(self.lib.wb_channels_launch(nat.stream_ptr(), nat.ptr(self.img), p.H * p.W, nat.ptr(self.oct),
                                              p.oct_total, self.wb_dtype, self.batch, nat.ptr(self.levels),
                                              p.n_levels, nat.ptr(self.chan_tiles), self.n_chan_tiles,
                                              nat.ptr(self.minmax), max(p.n_oct, 1), nat.ptr(self.taps),
                                              self.spec.func_id, p.shrink, p.smooth,
                                              self.cs_sn.ctypes.data_as(C.POINTER(C.c_double)), nat.ptr(self.chn),
                                              self.chn_stride), "wb_channels_launch")

    def run_channels(self):
        self.launch_octaves()
        self.launch_channels()

    def _casc_state(self, dm):
        import torch
        key = id(dm)
        stt = self._casc.get(key)
        if stt is None:
            tiles = self.plan.casc_tiles(dm.m, dm.n, dm.tile_rows, dm.tile_cols)
            csr = self.plan.tile_csr(tiles, max(self.plan.n_levels, 1))
            T1 = max(dm.n_stages, 1)
            stt = dict(
                dm=dm, n_tiles=int(tiles.size),
                tiles=torch.from_numpy(tiles.view(np.uint8).copy()).to(self.dev) if tiles.size else None,
                csr=torch.from_numpy(csr).to(self.dev),
                tile_hist=torch.empty((self.batch, max(int(tiles.size), 1), T1), dtype=torch.int32, device=self.dev),
                alive=torch.zeros((self.batch, max(self.plan.n_levels, 1), T1), dtype=torch.int32, device=self.dev))
            self._casc = {key: stt}          # one cascade resident per engine
        return stt

    def launch_cascade(self, dm, reduce=True):
        """The cascade scan; reduce=False launches the tile kernel alone (no per-level statistics), which
        is what bench.py times for the roofline of that kernel."""
        stt = self._casc_state(dm)
        if stt["n_tiles"] == 0:
            return stt
        nat.check(self.lib.wb_cascade_launch(nat.stream_ptr(), dm.handle, nat.ptr(self.chn), self.spec.wb_dtype,
                                             self.chn_stride,
                                             self.batch, nat.ptr(self.levels), self.plan.n_levels,
                                             nat.ptr(stt["tiles"]), nat.ptr(stt["csr"]), stt["n_tiles"],
                                             nat.ptr(self.detb.recs), nat.ptr(self.detb.counts), self.detb.cap,
                                             nat.ptr(stt["tile_hist"]), nat.ptr(stt["alive"] if reduce else None)),
                  "wb_cascade_launch")
        return stt

    def run_cascade(self, dm):
        """Zero the counters and scan every level of every image with cascade `dm`."""
        stt = self._casc_state(dm)
        self.detb.zero()
        if stt["n_tiles"] == 0:
            stt["alive"].zero_()
        return self.launch_cascade(dm)

    def run(self, dm):
        self.run_channels()
        return self.run_cascade(dm)

    # ------------------------------------------------------------------ hipGraph
    def capture(self, dm):
        """Capture octaves -> channels -> cascade into one hipGraph (torch.cuda.CUDAGraph on ROCm).
        Returns the graph; ``graph.replay()`` re-runs the whole pipeline on the resident images."""
        import torch
        self._casc_state(dm)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.run(dm)                      # warm-up outside capture (module load, attributes)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.run(dm)
        return g

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
            self.run_cascade(dm)

    def shard_counts(self, dm):
        """Host copy of the shard counters after the last scan; re-runs the cascade with a larger buffer if a
        shard overflowed."""
        while True:
            counts = self.detb.counts.cpu().numpy().astype(np.int64)
            if counts.max(initial=0) <= self.detb.cap:
                return counts
            self.det_capacity = (int(counts.max() * 1.5) + 16) * nat.WB_DET_SHARDS
            self._alloc_det()
            self.run_cascade(dm)

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

    def level_tensor(self, b, l):
        """Channels of level l of image b as a device view [u,v,C] into the pyramid buffer."""
        lv = self.plan.levels[l]
        off = int(self.level_np[l]["chn_off"])
        u, v, C = lv["u"], lv["v"], self.spec.n_channels
        return self.chn[b, off:off + u * v * C].view(u, v, C)

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
