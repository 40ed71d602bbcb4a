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
from waldboost_amd.distributed import DetectionGatherer, agree_capacity, gather_records, reduce_alive, shard_range


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


def _packed(recs, cap_rows, shard_cap=32):
    """What wb_det_pack_launch leaves on a rank: a 4-word header, then the valid records back to back."""
    out = torch.zeros((1 + cap_rows, 4), dtype=torch.int32)
    out[0] = torch.tensor([recs.size, min(recs.size, shard_cap), min(recs.size, cap_rows), shard_cap], dtype=torch.int32)
    out[1:1 + recs.size] = torch.from_numpy(recs.view(np.int32).reshape(-1, 4).copy())
    return out


def _worker(rank, world, port, images_per_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        recs = _fake_detections(rank, images_per_rank[rank])
        g = DetectionGatherer(256, torch.device("cpu"))           # prefix: 256 records per rank, not the buffer
        g.gather(_packed(recs, 1024))
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


def test_overflowing_shard_or_short_prefix_is_reported():
    class G(DetectionGatherer):
        def __init__(self, packed, rows):
            self.world, self.rows, self.recv = 1, rows, packed[: 1 + rows].clone()
    recs = _fake_detections(0, 1)
    over = _packed(recs, 1024)
    over[0, 1] = 99                                      # fullest shard beyond its capacity of 32
    with pytest.raises(OverflowError):
        G(over, 1024).merged([1])
    with pytest.raises(OverflowError):
        G(_packed(recs, 1024), 64).merged([1])           # 150 detections, prefix of 64
    assert G(_packed(recs, 1024), 256).merged([1]).size == recs.size


# ------------------------------------------------------------------------------ the end-of-batch exchange
class _FakeScan:
    """A rank's scan state as agree_capacity sees it: `true_need` records would land in the fullest shard."""

    def __init__(self, cap, true_need):
        self.cap, self.true_need, self.grown = cap, true_need, []

    def need(self):
        return self.true_need

    def grow(self, cap):
        self.grown.append(cap)
        self.cap = cap


def _exchange_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (1) only rank 1 overflows: both ranks must end with the same capacity, in the same number of rounds
        scan = _FakeScan(cap=32, true_need=[20, 75][rank])
        rounds = agree_capacity(scan)
        # (2) unequal starting capacities, nobody overflows: the smaller one grows to the larger
        scan2 = _FakeScan(cap=[64, 16][rank], true_need=[10, 12][rank])
        rounds2 = agree_capacity(scan2)
        # (3) nothing to do
        scan3 = _FakeScan(cap=32, true_need=[0, 31][rank])
        rounds3 = agree_capacity(scan3)
        # (4) valid prefixes of different lengths (one rank may have none) to rank 0, alive summed everywhere
        recs = _fake_detections(rank, [3, 2][rank])
        if rank == 0:
            recs = recs[:0]
        det = gather_records(torch.from_numpy(recs.view(np.int32).reshape(-1, 4).copy()), [3, 2][rank], dst=0)
        alive = np.arange(12, dtype=np.int64).reshape(3, 4) * (rank + 1)
        tot = reduce_alive(alive)
        q.put((rank, dict(cap=scan.cap, grown=scan.grown, rounds=rounds, cap2=scan2.cap, rounds2=rounds2,
                          cap3=scan3.cap, rounds3=rounds3, grown3=scan3.grown,
                          det=None if det is None else det.tobytes(), tot=tot.tolist())))
    finally:
        dist.destroy_process_group()


def test_capacity_agreement_prefix_gather_and_alive_reduce_world_size_2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = out[0], out[1]
    # (1) rank 0 did not overflow but grows with rank 1, to the same size, after the same single round
    assert a["cap"] == b["cap"] == int(75 * 1.5) + 16 and a["rounds"] == b["rounds"] == 1
    assert a["grown"] == b["grown"] == [a["cap"]]
    # (2) capacities equalised without a rescan on the rank that was large enough
    assert a["cap2"] == b["cap2"] == 64 and a["rounds2"] == b["rounds2"] == 1
    # (3) no growth at all
    assert a["cap3"] == b["cap3"] == 32 and a["rounds3"] == b["rounds3"] == 0 and a["grown3"] == b["grown3"] == []
    # (4) rank 0 holds rank 1's records with global image indices (rank 0 scanned 3 images, found nothing)
    assert b["det"] is None
    got = np.frombuffer(a["det"], nat.DET_DTYPE)
    want = _fake_detections(1, 2)
    want["image"] += 3
    want = want[np.lexsort((want["c"], want["r"], want["level"], want["image"]))]
    assert np.array_equal(np.stack([got[k].astype(np.int64) for k in ("image", "level", "r", "c")]),
                          np.stack([want[k].astype(np.int64) for k in ("image", "level", "r", "c")]))
    assert sorted(got["score"].tolist()) == sorted(want["score"].tolist())
    assert a["tot"] == b["tot"] == (np.arange(12).reshape(3, 4) * 3).tolist()


def _round_gatherer_worker(rank, world, port, q):
    """RoundGatherer on gloo/CPU: two buffer sets, three slots per rank, ranks with different detection counts."""
    import torch
    import torch.distributed as dist
    from waldboost_amd.distributed import RoundGatherer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows, slots = 16, 3
        g = RoundGatherer(rows, slots, "cpu")
        want = {}
        for which in range(g.SETS):
            for slot in range(slots):
                for r in range(world):
                    n = (3 * r + 5 * slot + 7 * which) % 11            # 0 .. 10 records
                    d = np.zeros(n, nat.DET_DTYPE)
                    d["image"], d["level"] = 0, np.arange(n) % 3
                    d["r"], d["c"] = 100 * r + np.arange(n), 10 * slot + which
                    d["score"] = np.arange(n, dtype=np.float32) + 0.25 * r
                    want[(which, slot, r)] = d
                    if r == rank:
                        g.send[which][slot].copy_(_packed(d, rows))
            g.gather(which)
        for which in range(g.SETS):
            for slot in range(slots):
                got = g.merged(which, slot, [1] * world)
                exp = np.concatenate([want[(which, slot, r)] for r in range(world)])
                exp["image"] = np.concatenate([np.full(want[(which, slot, r)].size, r) for r in range(world)])
                exp = exp[np.lexsort((exp["c"], exp["r"], exp["level"], exp["image"]))]
                assert np.array_equal(got, exp), (which, slot)
        q.put((rank, "ok"))
    except Exception as e:                                   # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_round_gatherer_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    procs = [ctx.Process(target=_round_gatherer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == {0: "ok", 1: "ok"}, res


def _run_bench(*flags):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=600)


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus N` with a bare environment (how the driver calls it at N=1) must start N ranks itself,
    rendezvous, and let rank 0 print ONE JSON line; --dry-launch stops there, before anything needs a GPU."""
    import json
    r = _run_bench("--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    # stdout is that line and nothing else (gloo's connection banner, which it writes to stdout, goes to stderr)
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout
    out = json.loads(r.stdout)
    assert out["dry_launch"] and out["n_gpus"] == 2 and out["world_size_reported"] == 2
    assert out["ranks"] == [0, 1] and out["local_ranks"] == [0, 1] and out["steps"] == 2


def test_bench_dry_launch_reaches_rank_0_with_eight_ranks():
    """BASELINE configs[3]'s launch shape, `--gpus 8`: eight ranks rendezvous (gloo, CPU) and rank 0 prints the one line."""
    import json
    r = _run_bench("--gpus", "8", "--dry-launch", "--steps", "20", "--warmup", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout
    out = json.loads(r.stdout)
    assert out["dry_launch"] and out["n_gpus"] == 8 and out["world_size_reported"] == 8
    assert out["ranks"] == list(range(8)) and out["local_ranks"] == list(range(8))


def test_bench_propagates_a_failing_rank():
    """Without a GPU the real run cannot start: every rank exits non-zero with a clear message and the parent
    returns that failure instead of hanging or printing a line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--no-cpu-baseline", "--steps", "2")
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())
