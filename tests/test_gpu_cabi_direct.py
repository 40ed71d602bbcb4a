"""The C ABI bound directly with ctypes, the way INTEGRATION.md shows a maintainer of the reference would do
it: raw device pointers from torch, host tables from the plan, no waldboost_amd.engine / Model in the call
path.  One 1-level scan (Model.predict_on_image) and one whole pyramid + cascade (Model.detect), checked against
the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import wb_oracle as orc
from waldboost_amd import _native as nat                       # only for the library path and the struct dtypes
from waldboost_amd.plan import PyramidPlan
from waldboost_amd.synth import random_tree_arrays, synth_image

pytestmark = pytest.mark.gpu
P = C.c_void_p


def _lib():
    import torch  # noqa: F401  (the HIP runtime the library binds to)
    lib = C.CDLL(nat.LIB_PATH)
    lib.wb_last_error.restype = C.c_char_p
    return lib


def _check(lib, rc):
    assert rc == 0, lib.wb_last_error().decode()


def _model_arrays(seed, T, shape):
    rng = np.random.default_rng(seed)
    trees, thetas, acc = [], [], 0.0
    for t in range(T):
        trees.append(random_tree_arrays(rng, shape, 2, 2.0, 60.0))
        acc += -0.15 if t % 3 else -0.45
        thetas.append(float("-inf") if t % 5 == 4 else float(np.float32(acc)))
    node_off = np.cumsum([0] + [tr[0].shape[0] for tr in trees]).astype(np.int32)
    cat = [np.ascontiguousarray(np.concatenate([tr[k] for tr in trees])) for k in range(5)]
    return trees, thetas, node_off, cat


def _create(lib, node_off, cat, thetas, shape):
    f, thr, l, r, p = cat
    th = np.array(thetas, np.float32)
    h = P()
    _check(lib, lib.wb_model_create(C.c_int(len(thetas)), node_off.ctypes.data_as(P), f.ctypes.data_as(P), thr.ctypes.data_as(P),
                                    l.ctypes.data_as(P), r.ctypes.data_as(P), p.ctypes.data_as(P), th.ctypes.data_as(P),
                                    C.c_int(shape[0]), C.c_int(shape[1]), C.c_int(shape[2]), C.byref(h)))
    info = nat.WbModelInfo()
    _check(lib, lib.wb_model_info(h, C.byref(info)))
    return h, info


def _records(det, cap):
    counts = det[:16].reshape(-1)[:64].astype(np.int64)
    recs = det[16:].reshape(64, cap, 4)
    out = np.concatenate([recs[s, :min(counts[s], cap)] for s in range(64)]) if counts.sum() else np.zeros((0, 4), np.int32)
    d = np.ascontiguousarray(out).view(nat.DET_DTYPE).reshape(-1)
    return d[np.lexsort((d["c"], d["r"], d["level"], d["image"]))]


def test_pyramid_and_cascade_through_the_raw_abi():
    import torch
    lib = _lib()
    dev = torch.device("cuda")
    st = P(torch.cuda.current_stream().cuda_stream)
    shape = (12, 12, 4)
    trees, thetas, node_off, cat = _model_arrays(3, 40, shape)
    h, info = _create(lib, node_off, cat, thetas, shape)
    img = synth_image(300, 420, 8)
    H, W = img.shape
    plan = PyramidPlan(H, W, 2, 8, 1)
    table, chn_total = plan.level_table()
    taps, _ = plan.tap_table()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    img_d = torch.zeros(H * W + 16, dtype=torch.uint8, device=dev)
    img_d[:H * W] = torch.from_numpy(img.reshape(-1)).to(dev)
    oct_d = torch.zeros(plan.oct_total + 16, dtype=torch.uint8, device=dev)
    minmax = torch.zeros((plan.n_oct, 2), dtype=torch.int32, device=dev)
    oct_off = (C.c_int64 * plan.n_oct)(*[int(x) for x in plan.oct_off])
    _check(lib, lib.wb_octaves_launch(st, P(img_d.data_ptr()), C.c_int(0), C.c_int(1), C.c_int(H), C.c_int(W), C.c_int64(H * W),
                                      P(oct_d.data_ptr()), C.c_int64(plan.oct_total), oct_off, C.c_int(plan.n_oct), P(minmax.data_ptr())))
    levels_d, taps_d, ctiles = up(table), up(taps), plan.chan_tiles()
    ctiles_d = up(ctiles)
    chn = torch.zeros(chn_total, dtype=torch.float32, device=dev)
    theta = np.linspace(0, np.pi, 5)
    cs_sn = np.concatenate([np.cos(theta[:-1]), np.sin(theta[:-1])])
    _check(lib, lib.wb_channels_launch(st, P(img_d.data_ptr()), C.c_int64(H * W), P(oct_d.data_ptr()), C.c_int64(plan.oct_total),
                                       C.c_int(0), C.c_int(1), P(levels_d.data_ptr()), C.c_int(plan.n_levels), P(ctiles_d.data_ptr()),
                                       C.c_int(ctiles.size), P(minmax.data_ptr()), C.c_int(plan.n_oct), P(taps_d.data_ptr()),
                                       C.c_int(0), C.c_int(2), C.c_int(1), cs_sn.ctypes.data_as(C.POINTER(C.c_double)),
                                       P(chn.data_ptr()), C.c_int64(chn_total), None, None, C.c_int64(0)))
    tiles = plan.casc_tiles(12, 12, info.tile_rows, info.tile_cols)
    tiles_d = up(tiles)
    cap, T = 4096, len(thetas)
    det = torch.zeros((16 + 64 * cap, 4), dtype=torch.int32, device=dev)          # the caller zeroes the counters ...
    alive = torch.zeros((1, plan.n_levels, T), dtype=torch.int32, device=dev)    # ... and the statistics (accumulated into)
    _check(lib, lib.wb_cascade_launch(st, h, P(chn.data_ptr()), C.c_int(1), C.c_int64(chn_total), C.c_int(1), P(levels_d.data_ptr()),
                                      C.c_int(plan.n_levels), P(tiles_d.data_ptr()), C.c_int(tiles.size),
                                      P(det[16:].data_ptr()), P(det.data_ptr()), C.c_uint32(cap), P(alive.data_ptr())))
    torch.cuda.synchronize()
    d = _records(det.cpu().numpy(), cap)
    otrees = [orc.make_tree(*tr) for tr in trees]
    ref = orc.detect(shape, dict(shrink=2, n_per_oct=8, smooth=1, channels=orc.grad_hist), otrees, thetas, img)
    assert ref["scores"].size > 0
    assert np.array_equal(d["level"], ref["level"]) and np.array_equal(d["r"], ref["r"]) and np.array_equal(d["c"], ref["c"])
    assert np.array_equal(d["score"].view(np.uint32), ref["scores"].view(np.uint32))
    assert np.array_equal(alive.cpu().numpy()[0].astype(np.int64), ref["alive"])
    # the same scan on WB_DTYPE_RANK8 channels: the channel kernel writes ranks only (chn = NULL), the cascade reads
    # them with chn_dtype = 2; then the packed read-back form (wb_det_pack_launch)
    assert info.rank_ok == 1
    rank = torch.zeros(chn_total + 16, dtype=torch.uint8, device=dev)
    _check(lib, lib.wb_channels_launch(st, P(img_d.data_ptr()), C.c_int64(H * W), P(oct_d.data_ptr()), C.c_int64(plan.oct_total),
                                       C.c_int(0), C.c_int(1), P(levels_d.data_ptr()), C.c_int(plan.n_levels), P(ctiles_d.data_ptr()),
                                       C.c_int(ctiles.size), P(minmax.data_ptr()), C.c_int(plan.n_oct), P(taps_d.data_ptr()),
                                       C.c_int(0), C.c_int(2), C.c_int(1), cs_sn.ctypes.data_as(C.POINTER(C.c_double)),
                                       None, C.c_int64(chn_total), h, P(rank.data_ptr()), C.c_int64(chn_total)))
    det2 = torch.zeros((16 + 64 * cap, 4), dtype=torch.int32, device=dev)
    alive2 = torch.zeros((1, plan.n_levels, T), dtype=torch.int32, device=dev)
    _check(lib, lib.wb_cascade_launch(st, h, P(rank.data_ptr()), C.c_int(2), C.c_int64(chn_total), C.c_int(1), P(levels_d.data_ptr()),
                                      C.c_int(plan.n_levels), P(tiles_d.data_ptr()), C.c_int(tiles.size),
                                      P(det2[16:].data_ptr()), P(det2.data_ptr()), C.c_uint32(cap), P(alive2.data_ptr())))
    packed = torch.zeros((1 + 64 * cap, 4), dtype=torch.int32, device=dev)
    _check(lib, lib.wb_det_pack_launch(st, P(det2[16:].data_ptr()), P(det2.data_ptr()), C.c_uint32(cap), P(packed.data_ptr()),
                                       C.c_uint32(64 * cap)))
    torch.cuda.synchronize()
    pk = packed.cpu().numpy()
    assert pk[0].tolist() == [d.size, int(det2[:16].max()), d.size, cap]
    d2 = pk[1:1 + d.size].copy().view(nat.DET_DTYPE).reshape(-1)
    d2 = d2[np.lexsort((d2["c"], d2["r"], d2["level"]))]
    assert np.array_equal(d2, d) and np.array_equal(alive2.cpu().numpy(), alive.cpu().numpy())
    # boxes
    inv = torch.from_numpy(np.array([np.float32(1.0 / s) for s in plan.scales], np.float32)).to(dev)
    recs = torch.from_numpy(d.view(np.int32).reshape(-1, 4).copy()).to(dev)
    boxes = torch.empty((d.size, 4), dtype=torch.float32, device=dev)
    scores = torch.empty(d.size, dtype=torch.float32, device=dev)
    _check(lib, lib.wb_boxes_launch(st, P(recs.data_ptr()), C.c_int64(d.size), P(inv.data_ptr()), C.c_int(12), C.c_int(12),
                                    P(boxes.data_ptr()), P(scores.data_ptr())))
    assert np.array_equal(boxes.cpu().numpy().view(np.uint32), ref["boxes"].view(np.uint32))
    # the last step of Model.detect in one launch (wb_det_finish_launch): header | sort keys | boxes | scores
    Pn = 2 * ((d.size + 8) // 2)
    fin = torch.zeros(16 + 28 * Pn, dtype=torch.uint8, device=dev)
    _check(lib, lib.wb_det_finish_launch(st, P(det2[16:].data_ptr()), P(det2.data_ptr()), C.c_uint32(cap), P(inv.data_ptr()),
                                         C.c_int(plan.n_levels), C.c_int(max(int(lv["u"]) for lv in plan.levels)),
                                         C.c_int(max(int(lv["v"]) for lv in plan.levels)), C.c_int(12), C.c_int(12),
                                         P(fin.data_ptr()), C.c_uint32(Pn)))
    torch.cuda.synchronize()
    fh = fin.cpu().numpy()
    assert fh[:16].view(np.int32).tolist() == [d.size, int(det2[:16].max()), d.size, cap]
    keys = np.sort(fh[16:16 + 8 * Pn].view(np.uint64)[:d.size])
    at = (keys & np.uint64((1 << 26) - 1)).astype(np.intp)
    assert np.array_equal((keys >> np.uint64(54)).astype(np.int64), ref["level"])
    assert np.array_equal(((keys >> np.uint64(40)) & np.uint64(0x3fff)).astype(np.int64), ref["r"])
    assert np.array_equal(((keys >> np.uint64(26)) & np.uint64(0x3fff)).astype(np.int64), ref["c"])
    fb = fh[16 + 8 * Pn:16 + 24 * Pn].view(np.float32).reshape(Pn, 4)[at]
    fs = fh[16 + 24 * Pn:].view(np.float32)[at]
    assert np.array_equal(fb.view(np.uint32), ref["boxes"].view(np.uint32))
    assert np.array_equal(fs.view(np.uint32), ref["scores"].view(np.uint32))
    # ... and with the ordering done on the device too (wb_det_finish_sorted_launch; tests/test_gpu_finish.py has its
    # own cases): this scan's detections are more than the 4096 it orders, so the sections are what
    # wb_det_finish_launch wrote and header[3] says 0; the strongest 3000 of them, re-packed, arrive in order
    assert d.size > 4096
    fin2 = torch.full((16 + 28 * Pn,), 0xAB, dtype=torch.uint8, device=dev)
    dims = (C.c_int(plan.n_levels), C.c_int(max(int(lv["u"]) for lv in plan.levels)), C.c_int(max(int(lv["v"]) for lv in plan.levels)),
            C.c_int(12), C.c_int(12))
    _check(lib, lib.wb_det_finish_sorted_launch(st, P(det2[16:].data_ptr()), P(det2.data_ptr()), C.c_uint32(cap), P(inv.data_ptr()),
                                                *dims, P(fin2.data_ptr()), C.c_uint32(Pn), None, C.c_uint32(0)))
    torch.cuda.synchronize()
    f2 = fin2.cpu().numpy()
    assert f2[:16].view(np.int32).tolist() == [d.size, int(det2[:16].max()), d.size, 0]
    for lo, hi in ((16, 16 + 8 * d.size), (16 + 8 * Pn, 16 + 8 * Pn + 16 * d.size), (16 + 24 * Pn, 16 + 24 * Pn + 4 * d.size)):
        assert np.array_equal(f2[lo:hi], fh[lo:hi])
    keep = np.sort(np.argsort(-d["score"], kind="stable")[:3000])          # (d is in the reference's order: so is d[keep])
    few = torch.zeros((16 + 64 * 64, 4), dtype=torch.int32, device=dev)
    shard = np.arange(keep.size) % 61                                       # dealt over 61 of the 64 shards
    cnt = np.bincount(shard, minlength=64).astype(np.int32)
    body = np.zeros((64, 64, 4), np.int32)
    for s_ in range(61):
        body[s_, :cnt[s_]] = np.ascontiguousarray(d[keep[shard == s_]][::-1]).view(np.int32).reshape(-1, 4)
    few[:16] = torch.from_numpy(cnt.reshape(16, 4)).to(dev)
    few[16:] = torch.from_numpy(body.reshape(-1, 4)).to(dev)
    fin3 = torch.zeros(16 + 28 * 4096, dtype=torch.uint8, device=dev)
    _check(lib, lib.wb_det_finish_sorted_launch(st, P(few[16:].data_ptr()), P(few.data_ptr()), C.c_uint32(64), P(inv.data_ptr()),
                                                *dims, P(fin3.data_ptr()), C.c_uint32(4096), None, C.c_uint32(0)))
    torch.cuda.synchronize()
    f3 = fin3.cpu().numpy()
    assert f3[:16].view(np.int32).tolist() == [keep.size, int(cnt.max()), keep.size, 1]
    k3 = f3[16:16 + 8 * keep.size].view(np.uint64)
    assert np.array_equal(k3 >> np.uint64(26), keys[keep] >> np.uint64(26))
    assert np.array_equal(f3[16 + 8 * 4096:16 + 24 * 4096].view(np.uint32).reshape(-1, 4)[:keep.size], ref["boxes"].view(np.uint32)[keep])
    assert np.array_equal(f3[16 + 24 * 4096:].view(np.uint32)[:keep.size], ref["scores"].view(np.uint32)[keep])
    # a pyramid beyond the key's bit fields is refused, not truncated
    assert lib.wb_det_finish_launch(st, P(det2[16:].data_ptr()), P(det2.data_ptr()), C.c_uint32(cap), P(inv.data_ptr()),
                                    C.c_int(2000), C.c_int(100), C.c_int(100), C.c_int(12), C.c_int(12), P(fin.data_ptr()),
                                    C.c_uint32(Pn)) != 0
    assert b"do not fit" in lib.wb_last_error()
    _check(lib, lib.wb_model_destroy(h))
    # error reporting stays on the C side of the boundary
    assert lib.wb_cascade_launch(st, None, None, 1, 0, 1, None, 1, None, 1, None, None, 0, None) != 0
    assert b"null pointer" in lib.wb_last_error()
