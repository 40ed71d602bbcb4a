"""world_size-2 test of the multi-GPU path on CPU (gloo): batch sharding and the fixed-size
all-gather + merge of the sharded detection buffers (waldboost_amd/distributed.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from waldboost_amd import _native as nat
from waldboost_amd.distributed import DetectionGatherer, shard_range
from waldboost_amd.engine import DetBuffer


def test_shard_range_partitions_contiguously():
    for n in (0, 1, 7, 8, 64, 513):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1


def _fake_detections(rank, n_img):
    rng = np.random.default_rng(100 + rank)
    n = 150 + 37 * rank
    d = np.zeros(n, nat.DET_DTYPE)
    d["image"] = rng.integers(0, n_img, n)
    d["level"] = rng.integers(0, 20, n)
    d["r"] = rng.integers(0, 500, n)
    d["c"] = rng.integers(0, 900, n)
    d["score"] = rng.normal(size=n).astype(np.float32)
    return d


def _fill(detb, recs, rng):
    """Scatter records over the shards the way concurrent workgroups would."""
    shard = rng.integers(0, detb.NS, recs.size)
    counts = np.zeros(detb.NS, np.int32)
    buf = detb.recs.view(detb.NS, detb.cap, 4)
    raw = torch.from_numpy(recs.view(np.int32).reshape(-1, 4).copy())
    for i, s in enumerate(shard):
        buf[s, counts[s]] = raw[i]
        counts[s] += 1
    detb.counts.copy_(torch.from_numpy(counts))


def _worker(rank, world, port, images_per_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        detb = DetBuffer(32, torch.device("cpu"))
        recs = _fake_detections(rank, images_per_rank[rank])
        _fill(detb, recs, np.random.default_rng(rank))
        g = DetectionGatherer(detb)
        g.gather(detb)
        merged = g.merged(images_per_rank)
        q.put((rank, merged.tobytes()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_gather_and_merge_world_size_2():
    world, images_per_rank = 2, [3, 2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, images_per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # expected: both ranks' records, rank 1's images shifted by rank 0's image count, reference order
    parts = []
    for r in range(world):
        d = _fake_detections(r, images_per_rank[r])
        d["image"] += sum(images_per_rank[:r])
        parts.append(d)
    want = np.concatenate(parts)
    want = want[np.lexsort((want["c"], want["r"], want["level"], want["image"]))]
    for r in range(world):
        got = np.frombuffer(out[r], nat.DET_DTYPE)
        key = lambda a: np.stack([a["image"], a["level"], a["r"], a["c"]], 1)
        assert np.array_equal(key(got), key(want))
        assert sorted(got["score"].tolist()) == sorted(want["score"].tolist())


def test_overflowing_shard_is_reported():
    detb = DetBuffer(4, torch.device("cpu"))
    detb.counts[3] = 9
    class G(DetectionGatherer):
        def __init__(self, detb):
            self.world, self.NS, self.cap, self.rows = 1, detb.NS, detb.cap, detb.buf.shape[0]
            self.recv = detb.buf.clone()
    with pytest.raises(OverflowError):
        G(detb).merged([1])
