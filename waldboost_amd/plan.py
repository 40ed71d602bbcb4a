"""Host-side pyramid plan: octaves, level table, tile tables, HBM layout.

The level plan is Python-float arithmetic in the reference (channels.py:124-131) and is
reproduced here expression by expression -- it is never re-derived in C floats.
Everything in this module is NumPy/pure Python and runs without a GPU.

HBM layout per image (one contiguous buffer each, images of a batch at a fixed stride):
  octaves   : octave k>=1 at element offset oct_off[k] (octave 0 is the image itself)
  channels  : level l at float offset chn_off[l], [u][v][4]: one aligned float4 per pixel -- the
              layout channel_pyramid hands to callers, 16-byte stores for the channel kernel and
              one contiguous run per tile row for the cascade.  Other channel functions give
              [u][v][C] in their own dtype (uint8 x4 = one dword per pixel, or one channel);
              offsets count elements and every level starts on a multiple of 4 elements
"""
import math

import os

import numpy as np

from ._native import LEVEL_DTYPE, TAP_DTYPE, TILE_DTYPE

N_CHANNELS = 4          # grad_hist with n_bins=4 (reference channels.py:40): the default channel function


def chan_tile(chan_func, shrink):
    """(tile_u, tile_v) outputs per workgroup of the channel kernel for a channel function (WB_CHN_*) and shrink:
    asked of the library (wb_channels_tile) -- the kernel's compile-time geometry is the only authority."""
    import ctypes as C
    from . import _native as nat
    tu, tv = C.c_int(), C.c_int()
    nat.check(nat.load().wb_channels_tile(int(chan_func), int(shrink), C.byref(tu), C.byref(tv)), "wb_channels_tile")
    return tu.value, tv.value


def octave_shapes(H, W):
    """reference channels.py:93-101: halve (dropping odd tails) until w<8 or h<8."""
    out = []
    h, w = int(H), int(W)
    while not (w < 8 or h < 8):
        out.append((h, w))
        h, w = h // 2, w // 2
    return out


def xcd_order(n):
    """Permutation p so that workgroups b, b+8, b+16, ... (which share an XCD and its L2 under
    round-robin dispatch) receive consecutive entries of the natural tile order."""
    q, r = divmod(n, 8)
    b = np.arange(n)
    x = b % 8
    start = np.where(x < r, x * (q + 1), r * (q + 1) + (x - r) * q)
    return start + b // 8


class PyramidPlan:
    def __init__(self, H, W, shrink, n_per_oct, smooth=1, exact_single=False, n_chn=N_CHANNELS, chn_bytes=4, chan_func=0):
        assert shrink in (1, 2, 4), "Shrink factor must be 1 or 2 (4 is a documented extension)"
        self.H, self.W = int(H), int(W)
        self.shrink, self.n_per_oct, self.smooth = int(shrink), int(n_per_oct), int(smooth)
        self.n_chn, self.chn_bytes = int(n_chn), int(chn_bytes)     # channels per pixel, bytes per channel value
        self.chan_func = int(chan_func)                              # WB_CHN_* (0 = grad_hist): selects the channel kernel's tile
        # exact_single: one level at the image's own size, whatever that size is
        self.octaves = [(self.H, self.W)] if exact_single else octave_shapes(H, W)
        self.n_oct = len(self.octaves)
        off, acc = [], 0
        for k, (h, w) in enumerate(self.octaves):
            off.append(acc if k else 0)
            if k:
                acc += h * w
        self.oct_off = np.array(off, np.int64)
        self.oct_total = max(acc, 1)

        levels = []
        if exact_single:
            # one level at the image's own size (grad_hist on a bare image: no resize rounding)
            levels.append(dict(oct=0, h=self.H, w=self.W, nh=self.H, nw=self.W, scale=1.0))
        else:
            factor = 2 ** (-1 / n_per_oct)
            for o, (h, w) in enumerate(self.octaves):
                for i in range(n_per_oct):
                    s = factor ** i
                    nw, nh = int((w * s) / shrink) * shrink, int((h * s) / shrink) * shrink
                    real_scale = nw / self.W
                    levels.append(dict(oct=o, h=h, w=w, nh=nh, nw=nw, scale=real_scale / shrink))
        for lv in levels:
            lv["u"], lv["v"] = lv["nh"] // shrink, lv["nw"] // shrink
            if lv["u"] >= 65536 or lv["v"] >= 65536:
                raise ValueError("channel image larger than 65535 pixels per side")
        self.levels = levels
        self.n_levels = len(levels)
        self.scales = [lv["scale"] for lv in levels]

        self._table = None
        self._chan_tiles = None
        # tile lists with the workgroups known to be short dispatched last on every XCD (_tiles, _short_last); the engine
        # says how many images a launch holds
        self.batch_hint = 1

    def _short_last(self, n_tiles):
        """Whether a list of n_tiles tiles is dealt with the short workgroups last on every XCD -- which also hands every XCD
        an even share of the long ones.  Contiguous eighths of the level-major list give one XCD the largest level's full
        tiles and another the small levels' cut ones; with several images per launch the dispatch phase of image y is
        (y * n_tiles) mod 8, so the shares rotate over the XCDs -- unless n_tiles is a multiple of 8 or 4 (4K, shrink 4:
        19224 channel tiles, the same XCD takes the same share of every image: 252 against 240 us per image at any
        batch size).  Measured at 1080p (3421 / 1670 tiles, full rotation): +3 % at batch 2, +1 % at 4, nothing at 8 and
        16, -1 % at 64 (the plain order keeps neighbouring tiles on one XCD's L2).  WB_TILE_ORDER=natural / short: A/B runs."""
        mode = os.environ.get("WB_TILE_ORDER", "")
        if mode in ("natural", "short"):
            return mode == "short"
        return self.batch_hint <= 4 or n_tiles % 8 in (0, 4)

    # ------------------------------------------------------------------ layout
    def chn_offsets(self):
        offs, acc = [], 0
        for lv in self.levels:
            offs.append(acc)
            acc += (self.n_chn * lv["u"] * lv["v"] + 3) & ~3
        return offs, max(acc, 4)

    @staticmethod
    def axis_taps(n_in, n_out):
        """Resampling taps of one axis (WbTap): scipy NI_ZoomShift with grid_mode=True, order 1,
        mode='mirror', in fp64 with scipy's operation order."""
        t = np.zeros(n_out, TAP_DTYPE)
        step = np.float64(n_in) / np.float64(n_out)
        cc = ((np.arange(n_out, dtype=np.float64) + 0.5) * step) - 0.5
        fl = np.floor(cc)
        x = cc - fl
        t["w0"] = 1.0 - x
        t["w1"] = 1.0 - t["w0"]
        i0 = fl.astype(np.int64)

        def mirror(i):
            i = np.abs(i)
            i = np.where(i >= n_in, 2 * (n_in - 1) - i, i)
            return np.maximum(i, 0)

        t["i0"], t["i1"] = mirror(i0), mirror(i0 + 1)
        return t

    def tap_table(self):
        """All levels' taps back to back (rows then columns of each level) and the level offsets."""
        parts, offs, acc = [], [], 0
        for lv in self.levels:
            offs.append(acc)
            parts += [self.axis_taps(lv["h"], lv["nh"]), self.axis_taps(lv["w"], lv["nw"])]
            acc += lv["nh"] + lv["nw"]
        return (np.concatenate(parts) if parts else np.zeros(1, TAP_DTYPE)), offs

    def level_table(self):
        if self._table is None:
            offs, total = self.chn_offsets()
            _, tap_offs = self.tap_table()
            t = np.zeros(self.n_levels, LEVEL_DTYPE)
            for i, lv in enumerate(self.levels):
                t[i]["tap_off"] = tap_offs[i]
                t[i]["oct"] = lv["oct"]
                t[i]["src_h"], t[i]["src_w"] = lv["h"], lv["w"]
                t[i]["nh"], t[i]["nw"] = lv["nh"], lv["nw"]
                t[i]["u"], t[i]["v"] = lv["u"], lv["v"]
                t[i]["src_off"] = self.oct_off[lv["oct"]]
                t[i]["chn_off"] = offs[i]
                # scipy zoom recomputes the step from the integer shapes in fp64
                t[i]["sy"] = np.float64(lv["h"]) / np.float64(lv["nh"])
                t[i]["sx"] = np.float64(lv["w"]) / np.float64(lv["nw"])
            self._table = (t, total)
        return self._table

    # ------------------------------------------------------------------ tiles
    @staticmethod
    def _tiles(dims, tr, tc, cheap=None, short_last=lambda n: True):
        """dims: per level (rows, cols) to cover with tr x tc tiles; natural order, XCD-permuted.
        cheap: callable(tiles in natural order) -> (bool mask of the tiles known to be SHORT workgroups -- the channel
        kernel's identity levels (a copy instead of a resample), tiles cut by the edge of their level --, their relative
        cost).  Every XCD is handed its share of them, costliest first, behind its share of the others: the workgroups
        that start last are the short ones (a kernel ends one workgroup lifetime after its last dispatch, and in a
        pipeline of several streams the next kernel's workgroups move into slots that free up one by one)."""
        parts = []
        for l, (r, c) in enumerate(dims):
            if r <= 0 or c <= 0:
                continue
            ny, nx = -(-r // tr), -(-c // tc)
            ty, tx = np.divmod(np.arange(ny * nx), nx)
            a = np.zeros(ny * nx, TILE_DTYPE)
            a["level"], a["ty"], a["tx"] = l, ty, tx
            parts.append(a)
        if not parts:
            return np.zeros(0, TILE_DTYPE)
        nat = np.concatenate(parts)
        if cheap is not None and short_last(nat.size):
            mask, cost = cheap(nat)
            a, b = nat[~mask], nat[mask]
            b = b[np.argsort(-cost[mask], kind="stable")]
            n = nat.size
            slots = [len(range(x, n, 8)) for x in range(8)]
            ia = np.array_split(np.arange(a.size), 8)
            if a.size and b.size and all(len(ia[x]) <= slots[x] for x in range(8)):
                # XCD x runs dispatch slots x, x + 8, ...: its share of `a` (a contiguous run of the natural order, as
                # xcd_order deals them), then every eighth tile of `b` (costliest first on every XCD)
                out = np.zeros(n, TILE_DTYPE)
                need = [slots[x] - len(ia[x]) for x in range(8)]
                ib = [[] for _ in range(8)]
                x = 0
                for j in range(b.size):                      # deal b round-robin over the XCDs that still have room
                    while len(ib[x]) >= need[x]:
                        x = (x + 1) % 8
                    ib[x].append(j)
                    x = (x + 1) % 8
                for x in range(8):
                    out[x::8] = np.concatenate([a[ia[x]], b[np.asarray(ib[x], dtype=np.intp)]])
                return out
        return nat[xcd_order(nat.size)]

    @staticmethod
    def tile_csr(tiles, n_levels):
        """int32 [n_levels + 1 + n_tiles]: per level the range [start, end) into the trailing list
        of (launch-order) tile indices belonging to it -- what the cascade's statistics reduction
        walks (include/waldboost_hip.h: wb_cascade_launch)."""
        order = np.argsort(tiles["level"], kind="stable").astype(np.int32)
        counts = np.bincount(tiles["level"], minlength=n_levels)[:n_levels]
        start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        return np.concatenate([start, order]).astype(np.int32)

    def chan_tiles(self):
        if self._chan_tiles is None:
            tu, tv = chan_tile(self.chan_func, self.shrink)
            # the channel kernel's short workgroups: tiles of identity levels (level = its octave's own size: a copy
            # instead of a resample, 8 against 12.5 ns per tile) and tiles cut by the bottom edge of their level (the
            # kernel computes only the rows they hold)
            ident = np.array([lv["h"] == lv["nh"] and lv["w"] == lv["nw"] for lv in self.levels], bool)
            us = np.array([lv["u"] for lv in self.levels], np.int64)

            def cheap(nat):
                rows = np.minimum(tu, us[nat["level"]] - nat["ty"].astype(np.int64) * tu)   # output rows the tile holds
                cost = np.where(ident[nat["level"]], 0.65, 1.0) * (0.35 + 0.65 * rows / tu)
                return ident[nat["level"]] | (rows < tu), cost
            self._chan_tiles = self._tiles([(lv["u"], lv["v"]) for lv in self.levels], tu, tv, cheap=cheap, short_last=self._short_last)
        return self._chan_tiles

    def window_grid(self, m, n):
        """(u-m) x (v-n) windows per level -- one row and one column fewer than geometrically
        valid, exactly like reference model.py:243."""
        return [(max(lv["u"] - m, 0), max(lv["v"] - n, 0)) for lv in self.levels]

    def casc_tiles(self, m, n, tile_rows, tile_cols):
        grid = self.window_grid(m, n)
        # the cascade's short workgroups: tiles cut by an edge of their level
        nr = np.array([g[0] for g in grid], np.int64)
        nc = np.array([g[1] for g in grid], np.int64)

        def cheap(nat):
            rows = np.minimum(tile_rows, nr[nat["level"]] - nat["ty"].astype(np.int64) * tile_rows)
            cols = np.minimum(tile_cols, nc[nat["level"]] - nat["tx"].astype(np.int64) * tile_cols)
            cost = rows * cols / float(tile_rows * tile_cols)
            return cost < 1.0, cost
        return self._tiles(grid, tile_rows, tile_cols, cheap=cheap, short_last=self._short_last)

    def n_loc(self, m, n):
        return int(sum(r * c for r, c in self.window_grid(m, n)))

    # ------------------------------------------------------------------ roofline arithmetic
    def algorithmic_bytes(self, px_bytes):
        """Per image, SURVEY section 8(d): image read + octave writes + per-level source reads
        + channel write (+ the same bytes again for the cascade's read); channels in the dtype
        and count of the plan's channel function."""
        img = self.H * self.W * px_bytes
        octw = sum(h * w for h, w in self.octaves[1:]) * px_bytes
        src = sum(lv["h"] * lv["w"] for lv in self.levels) * px_bytes
        chn = sum(lv["u"] * lv["v"] for lv in self.levels) * self.n_chn * self.chn_bytes
        return dict(image=img, octaves=octw, level_src=src, chn_write=chn, chn_read=chn,
                    channels_kernel=src + chn, cascade_kernel=chn, total=img + octw + src + 2 * chn)
