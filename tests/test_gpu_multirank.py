"""Two ranks on ONE GPU (gloo, host-staged gather): the end-to-end sharded detection of
waldboost_amd.distributed.detect_sharded against a single-process run.  The real multi-GPU path
uses backend "nccl" (RCCL) with one GPU per rank; this rehearses everything but the transport."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import waldboost_amd as wb
        from waldboost_amd import engine as _engine
        from waldboost_amd.distributed import detect_sharded, shard_range
        from waldboost_amd.synth import synth_image
        here = os.path.dirname(os.path.abspath(__file__))
        M = wb.load(os.path.join(here, "golden", "mixed_d2_T24.pb"))
        lo, hi = shard_range(5, rank, world)
        imgs = np.stack([synth_image(200, 264, 900 + b) for b in range(lo, hi)])      # this rank's shard only
        if rank == 1:
            # rank 1 starts with a detection buffer that is too small: both ranks must grow together
            eng = _engine.get_engine(200, 264, np.uint8, 2, 8, 1, hi - lo)
            eng.det_capacity = 16
            eng._alloc_det()
        det, alive, total = detect_sharded(M, imgs)
        caps = [e.detb.cap for e in _engine._ENGINES.values()]
        q.put((rank, None if det is None else det.tobytes(), alive.shape, total.tolist(), (M.n_loc, M.n_weak), caps))
    finally:
        dist.destroy_process_group()


def test_detect_sharded_two_ranks_one_gpu():
    import waldboost_amd as wb
    from waldboost_amd import _native as nat
    from waldboost_amd.synth import synth_image
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict((r, rest) for r, *rest in (q.get(timeout=300) for _ in range(2)))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    M = wb.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mixed_d2_T24.pb"))
    imgs = np.stack([synth_image(200, 264, 900 + b) for b in range(5)])
    ref = M.detect_batch_raw(imgs)
    assert out[1][0] is None                                # the records go to rank 0 only
    got = np.frombuffer(out[0][0], nat.DET_DTYPE)
    assert np.array_equal(got["image"], ref["image"]) and np.array_equal(got["level"], ref["level"])
    assert np.array_equal(got["r"], ref["r"].astype(np.uint16)) and np.array_equal(got["c"], ref["c"].astype(np.uint16))
    assert np.array_equal(got["score"].view(np.uint32), ref["scores"].view(np.uint32))
    assert out[0][1][0] == 3 and out[1][1][0] == 2          # shard sizes 3 + 2
    for r in range(2):                                      # the summed statistics, on every rank
        assert np.array_equal(np.array(out[r][2]), ref["alive"].sum(axis=0))
        assert out[r][3] == (M.n_loc, M.n_weak)
    assert out[0][4] == out[1][4] and out[1][4][0] > 16     # rank 1 overflowed; both ranks grew to the same capacity


def _chunk_worker(rank, world, port, q):
    """Shards scanned in chunks (batch=2): rank 0 holds 4 images = two full chunks on two engines and two streams, rank 1
    three = a full chunk and a remainder; device tensors; the call repeated (graph replays); no per-image statistics."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import waldboost_amd as wb
        from waldboost_amd.distributed import detect_sharded, shard_range
        from waldboost_amd.synth import synth_image
        here = os.path.dirname(os.path.abspath(__file__))
        M = wb.load(os.path.join(here, "golden", "mixed_d2_T24.pb"))
        lo, hi = shard_range(7, rank, world)
        imgs = torch.from_numpy(np.stack([synth_image(200, 264, 900 + b) for b in range(lo, hi)])).cuda()
        outs = []
        for rep in range(3):
            M.reset()
            det, alive, total = detect_sharded(M, imgs, batch=2, per_image_alive=(rep == 0))
            outs.append((None if det is None else det.tobytes(), None if alive is None else alive.tolist(), total.tolist(),
                         (M.n_loc, M.n_weak)))
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


def test_detect_sharded_in_chunks_on_two_streams():
    import waldboost_amd as wb
    from waldboost_amd import _native as nat
    from waldboost_amd.synth import synth_image
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    M = wb.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mixed_d2_T24.pb"))
    imgs = np.stack([synth_image(200, 264, 900 + b) for b in range(7)])
    ref = M.detect_batch_raw(imgs)
    for rep in range(3):
        det0, alive0, total0, cnt0 = out[0][rep]
        det1, alive1, total1, cnt1 = out[1][rep]
        assert det1 is None
        got = np.frombuffer(det0, nat.DET_DTYPE)
        assert np.array_equal(got["image"], ref["image"]) and np.array_equal(got["level"], ref["level"])
        assert np.array_equal(got["r"], ref["r"].astype(np.uint16)) and np.array_equal(got["c"], ref["c"].astype(np.uint16))
        assert np.array_equal(got["score"].view(np.uint32), ref["scores"].view(np.uint32))
        assert np.array_equal(np.array(total0), ref["alive"].sum(axis=0)) and total0 == total1
        assert cnt0 == cnt1 == (M.n_loc, M.n_weak)
        if rep == 0:
            assert np.array_equal(np.concatenate([np.array(alive0), np.array(alive1)]), ref["alive"])
        else:
            assert alive0 is None and alive1 is None


def _edge_worker(rank, world, port, q):
    """float64 DEVICE tensors (held as float64 with a dtype code), an empty shard on rank 1, then images too small for
    any pyramid level: no rank may raise alone or block the other in a collective."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import waldboost_amd as wb
        from waldboost_amd.distributed import detect_sharded, shard_range
        from waldboost_amd.synth import synth_image
        here = os.path.dirname(os.path.abspath(__file__))
        M = wb.load(os.path.join(here, "golden", "mixed_d2_T24.pb"))
        lo, hi = shard_range(1, rank, world)                                       # rank 0: one image, rank 1: none
        imgs = np.stack([synth_image(200, 264, 900 + b) for b in range(lo, hi)] or [np.zeros((200, 264), np.uint8)])[: hi - lo]
        t = torch.from_numpy(imgs.astype(np.float64)).cuda()
        det, alive, total = detect_sharded(M, t)
        tiny = torch.zeros((hi - lo, 6, 40), dtype=torch.float64, device="cuda")  # h < 8: no octave, no level
        det0, alive0, total0 = detect_sharded(M, tiny)
        try:
            detect_sharded(M, torch.zeros((1, 200, 264), dtype=torch.complex64, device="cuda"))
            unsupported = "no error"
        except NotImplementedError:
            unsupported = "NotImplementedError"
        q.put((rank, None if det is None else det.tobytes(), alive.shape, total.tolist(), (M.n_loc, M.n_weak),
               (None if det0 is None else det0.size, alive0.shape, total0.shape, unsupported)))
    finally:
        dist.destroy_process_group()


def test_detect_sharded_float64_tensors_empty_shard_and_no_levels():
    import waldboost_amd as wb
    from waldboost_amd import _native as nat
    from waldboost_amd.synth import synth_image
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_edge_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict((r, rest) for r, *rest in (q.get(timeout=300) for _ in range(2)))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    M = wb.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mixed_d2_T24.pb"))
    ref = M.detect_raw(synth_image(200, 264, 900).astype(np.float64))
    got = np.frombuffer(out[0][0], nat.DET_DTYPE)
    assert out[1][0] is None and got.size == ref["scores"].size > 0
    assert np.array_equal(got["level"], ref["level"]) and np.array_equal(got["r"], ref["r"].astype(np.uint16))
    assert np.array_equal(got["score"].view(np.uint32), ref["scores"].view(np.uint32))
    assert out[0][1][0] == 1 and out[1][1][0] == 0
    for r in range(2):
        assert np.array_equal(np.array(out[r][2]), ref["alive"]) and out[r][3] == (M.n_loc, M.n_weak)
        assert out[r][4][1:] == ((1 - r, 0, len(M)), (0, len(M)), "NotImplementedError")
    assert out[0][4][0] == 0 and out[1][4][0] is None


def test_bench_multi_rank_region_with_its_exchange():
    """bench.py's multi-rank form on the one GPU of the box (a single-rank RCCL group, --force-collective): per-stream
    region graphs whose steps pack their detections, one all_gather at the region's end; the line must say so, carry the
    per-rank gate results and the check of what the collective delivered."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-collective", "--steps", "8", "--warmup", "2",
                        "--no-cpu-baseline", "--no-through-api", "--repeats", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout
    out = json.loads(r.stdout)
    assert out["n_gpus"] == 1 and out["steps"] == 8 and out["value"] > 0
    assert out["ranks"]["world_size_reported"] == 1 and out["ranks"]["parity_gate_passed"] == [True]
    assert "end of every timed region" in out["config"]["collective"]
    assert out["parity"]["bit_exact"] and out["parity"]["gathered"]["bit_exact"] and out["parity"]["gathered"]["step"] == 7
