"""wb_det_finish_sorted_launch on synthetic detection buffers: the ordering Model.detect used to do on the host
(reference model.py:173-179: levels in pyramid order, windows row-major inside a level; get_boxes model.py:136-147),
done on the device: every record is ranked by the number of smaller keys.  Checked against NumPy for record counts
around the sizes the kernel's loops change shape at (a workgroup's 32 records, a pass's 64 keys, the 4096-key limit
behind which the kernel leaves the ordering to the host), and wb_det_order_batch_launch -- the same per image of a
batch -- on records mixed over the shards as a batched scan leaves them."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = C.c_void_p
SHARDS = 64


def _finish(lib, fn, recs_by_shard, cap, inv, m, n, out_cap, n_levels=40, mr=16384, mc=16384, tail=None):
    import torch
    from waldboost_amd import _native as nat
    dev = "cuda:0"
    det = np.zeros((SHARDS * cap, 4), np.int32)
    counts = np.zeros(SHARDS, np.uint32)
    for s, r in enumerate(recs_by_shard):
        det[s * cap:s * cap + min(len(r), cap)] = r[:cap].view(np.int32).reshape(-1, 4)
        counts[s] = len(r)                                   # (a count above cap: records were dropped by the scan)
    det_d = torch.from_numpy(det).to(dev)
    cnt_d = torch.from_numpy(counts.view(np.int32)).to(dev)
    inv_d = torch.from_numpy(inv).to(dev)
    n_tail = 0 if tail is None else tail.size
    out = torch.full((16 + 28 * out_cap + 4 * n_tail,), 0xCD, dtype=torch.uint8, device=dev)
    more = ()
    if fn is lib.wb_det_finish_sorted_launch:                # (takes the caller's tail words: the scan's statistics)
        tail_d = None if tail is None else torch.from_numpy(tail).to(dev)
        more = (None if tail is None else P(tail_d.data_ptr()), C.c_uint32(n_tail))
    rc = fn(None, P(det_d.data_ptr()), P(cnt_d.data_ptr()), C.c_uint32(cap), P(inv_d.data_ptr()), C.c_int(n_levels), C.c_int(mr),
            C.c_int(mc), C.c_int(m), C.c_int(n), P(out.data_ptr()), C.c_uint32(out_cap), *more)
    assert rc == 0, lib.wb_last_error()
    torch.cuda.synchronize()
    h = out.cpu().numpy()
    if tail is not None:
        assert np.array_equal(h[16 + 28 * out_cap:].view(np.int32), tail)
    return (h[:16].view(np.int32), h[16:16 + 8 * out_cap].view(np.uint64), h[16 + 8 * out_cap:16 + 24 * out_cap].view(np.float32).reshape(-1, 4),
            h[16 + 24 * out_cap:16 + 28 * out_cap].view(np.float32))


def _records(rng, total, n_levels=40):
    from waldboost_amd import _native as nat
    # unique (level, r, c) triples, as a scan produces them
    flat = rng.choice(n_levels * 300 * 500, size=total, replace=False)
    d = np.zeros(total, nat.DET_DTYPE)
    d["level"], d["r"], d["c"] = flat // (300 * 500), (flat // 500) % 300, flat % 500
    d["score"] = rng.standard_normal(total).astype(np.float32)
    return d


@pytest.mark.parametrize("total", [0, 1, 2, 63, 127, 128, 129, 255, 256, 257, 1000, 2047, 2048, 2049, 3098, 4095, 4096])
def test_finish_sorted_orders_keys_boxes_and_scores(total):
    from waldboost_amd import _native as nat
    lib = nat.load()
    rng = np.random.default_rng(total)
    d = _records(rng, total)
    # uneven shards, some empty
    shard = rng.integers(0, SHARDS, total) if total % 2 else rng.choice([0, 5, 63], total)
    by = [d[shard == s] for s in range(SHARDS)]
    cap = max(16, max(len(b) for b in by))
    inv = (1.0 / (1.0 + 0.09 * np.arange(40))).astype(np.float32)
    m, n = 12, 14
    out_cap = 4096
    tail = rng.integers(-5, 1 << 30, size=(total * 7) % 9001, dtype=np.int32) if total % 3 else None   # (alive[] in Model.detect)
    hdr, keys, boxes, scores = _finish(lib, lib.wb_det_finish_sorted_launch, by, cap, inv, m, n, out_cap, tail=tail)
    assert hdr.tolist() == [total, max(len(b) for b in by) if total else 0, total, 1]
    order = np.lexsort((d["c"], d["r"], d["level"]))
    e = d[order]
    k = keys[:total]
    assert np.array_equal((k >> np.uint64(54)).astype(np.int64), e["level"])
    assert np.array_equal(((k >> np.uint64(40)) & np.uint64(0x3fff)).astype(np.int64), e["r"])
    assert np.array_equal(((k >> np.uint64(26)) & np.uint64(0x3fff)).astype(np.int64), e["c"])
    assert np.array_equal(np.sort(k & np.uint64((1 << 26) - 1)), np.arange(total))       # every packed position once
    assert np.array_equal(scores[:total].view(np.uint32), e["score"].view(np.uint32))
    sc = inv[e["level"]]
    c, r = e["c"].astype(np.int64), e["r"].astype(np.int64)
    want = np.stack([c.astype(np.float32) * sc, r.astype(np.float32) * sc, (c + n).astype(np.float32) * sc, (r + m).astype(np.float32) * sc], 1)
    assert np.array_equal(boxes[:total].view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("total,out_cap", [(4097, 8192), (6000, 8192), (3000, 2048), (5000, 4096)])
def test_finish_sorted_leaves_large_results_to_the_host(total, out_cap):
    """More valid records than the kernel sorts (4096) or than the buffer holds: header[3] = 0 and the sections are what
    wb_det_finish_launch writes."""
    from waldboost_amd import _native as nat
    lib = nat.load()
    rng = np.random.default_rng(total)
    d = _records(rng, total)
    shard = rng.integers(0, SHARDS, total)
    by = [d[shard == s] for s in range(SHARDS)]
    cap = max(len(b) for b in by)
    inv = (1.0 / (1.0 + 0.09 * np.arange(40))).astype(np.float32)
    a = _finish(lib, lib.wb_det_finish_sorted_launch, by, cap, inv, 12, 12, out_cap, tail=np.arange(777, dtype=np.int32))
    b = _finish(lib, lib.wb_det_finish_launch, by, cap, inv, 12, 12, out_cap)
    present = min(total, out_cap)
    assert a[0].tolist() == [total, cap, present, 0] and b[0].tolist() == [total, cap, present, cap]
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x[:present].view(np.uint8), y[:present].view(np.uint8))


def test_finish_sorted_with_an_overflowed_shard():
    """A shard whose counter ran past its capacity: its first `cap` records are valid, header[1] reports the overflow
    (the caller grows the buffer and scans again), the valid records still arrive in order."""
    from waldboost_amd import _native as nat
    lib = nat.load()
    rng = np.random.default_rng(7)
    d = _records(rng, 900)
    shard = rng.integers(0, 8, 900)
    by = [d[shard == s] for s in range(SHARDS)]
    cap = min(len(b) for b in by[:8]) - 5                    # every one of the eight used shards overflows by a few
    inv = np.ones(40, np.float32)
    hdr, keys, boxes, scores = _finish(lib, lib.wb_det_finish_sorted_launch, by, cap, inv, 12, 12, 4096)
    valid = np.concatenate([b[:cap] for b in by])
    assert hdr.tolist() == [valid.size, max(len(b) for b in by), valid.size, 1]
    e = valid[np.lexsort((valid["c"], valid["r"], valid["level"]))]
    assert np.array_equal(scores[:valid.size].view(np.uint32), e["score"].view(np.uint32))
    assert np.array_equal(boxes[:valid.size, 0], e["c"].astype(np.float32))


def _order_batch(lib, recs_by_shard, cap, n_images, inv, m, n, P_rows, n_levels=40):
    import torch
    dev = "cuda:0"
    det = np.zeros((SHARDS * cap, 4), np.int32)
    counts = np.zeros(SHARDS, np.uint32)
    for s, r in enumerate(recs_by_shard):
        det[s * cap:s * cap + min(len(r), cap)] = r[:cap].view(np.int32).reshape(-1, 4)
        counts[s] = len(r)
    det_d = torch.from_numpy(det).to(dev)
    cnt_d = torch.from_numpy(counts.view(np.int32)).to(dev)
    inv_d = torch.from_numpy(inv).to(dev)
    blk = 16 + 28 * P_rows
    scratch = torch.full((n_images * (256 + 16 * P_rows),), 0xEE, dtype=torch.uint8, device=dev)
    out = torch.full((16 + n_images * blk,), 0xCD, dtype=torch.uint8, device=dev)
    rc = lib.wb_det_order_batch_launch(None, P(det_d.data_ptr()), P(cnt_d.data_ptr()), C.c_uint32(cap), C.c_int(n_images), P(inv_d.data_ptr()),
                                       C.c_int(n_levels), C.c_int(16384), C.c_int(16384), C.c_int(m), C.c_int(n), P(scratch.data_ptr()),
                                       C.c_size_t(scratch.numel()), P(out.data_ptr()), C.c_uint32(P_rows))
    assert rc == 0, lib.wb_last_error()
    torch.cuda.synchronize()
    h = out.cpu().numpy()
    blocks = []
    for b in range(n_images):
        o = 16 + b * blk
        blocks.append((h[o:o + 16].view(np.int32), h[o + 16:o + 16 + 8 * P_rows].view(np.uint64),
                       h[o + 16 + 8 * P_rows:o + 16 + 24 * P_rows].view(np.float32).reshape(-1, 4), h[o + 16 + 24 * P_rows:o + blk].view(np.float32)))
    return h[:16].view(np.int32), blocks


@pytest.mark.parametrize("per_image", [[0, 1, 3000, 4096, 100], [700] * 16, [5000, 10, 0], [33]])
def test_order_batch_splits_by_image_and_orders_every_image(per_image):
    """wb_det_order_batch_launch: a batch's records, mixed over the shards as a batched scan leaves them, come back image
    by image in the reference's order; an image with more detections than a block holds is reported, not truncated
    silently."""
    from waldboost_amd import _native as nat
    lib = nat.load()
    rng = np.random.default_rng(sum(per_image))
    B, P_rows = len(per_image), 4096
    parts = []
    for b, k in enumerate(per_image):
        d = _records(rng, k)
        d["image"] = b
        parts.append(d)
    allr = np.concatenate(parts)
    allr = allr[rng.permutation(allr.size)]
    shard = rng.integers(0, SHARDS, allr.size)
    by = [allr[shard == s] for s in range(SHARDS)]
    cap = max(16, max(len(x) for x in by))
    inv = (1.0 / (1.0 + 0.09 * np.arange(40))).astype(np.float32)
    m, n = 12, 12
    info, blocks = _order_batch(lib, by, cap, B, inv, m, n, P_rows)
    assert info.tolist() == [allr.size, max(len(x) for x in by), B, P_rows]
    for b, (hdr, keys, boxes, scores) in enumerate(blocks):
        k = per_image[b]
        if k > P_rows:
            assert hdr[0] == P_rows and hdr[1] == k                        # the overflow is visible: the caller takes its other path
            continue
        assert hdr.tolist() == [k, k, k, 1]
        e = parts[b][np.lexsort((parts[b]["c"], parts[b]["r"], parts[b]["level"]))]
        kk = keys[:k]
        assert np.array_equal((kk >> np.uint64(54)).astype(np.int64), e["level"])
        assert np.array_equal(((kk >> np.uint64(40)) & np.uint64(0x3fff)).astype(np.int64), e["r"])
        assert np.array_equal(((kk >> np.uint64(26)) & np.uint64(0x3fff)).astype(np.int64), e["c"])
        assert np.array_equal(scores[:k].view(np.uint32), e["score"].view(np.uint32))
        sc = inv[e["level"]]
        c, r = e["c"].astype(np.int64), e["r"].astype(np.int64)
        want = np.stack([c.astype(np.float32) * sc, r.astype(np.float32) * sc, (c + n).astype(np.float32) * sc, (r + m).astype(np.float32) * sc], 1)
        assert np.array_equal(boxes[:k].view(np.uint32), want.view(np.uint32))
