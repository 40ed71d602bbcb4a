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
        from waldboost_amd.distributed import detect_sharded
        from waldboost_amd.synth import synth_image
        here = os.path.dirname(os.path.abspath(__file__))
        M = wb.load(os.path.join(here, "golden", "mixed_d2_T24.pb"))
        imgs = np.stack([synth_image(200, 264, 900 + b) for b in range(5)])
        det, alive = detect_sharded(M, imgs)
        q.put((rank, det.tobytes(), alive.shape))
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
    out = dict((r, (d, sh)) for r, d, sh in (q.get(timeout=300) for _ in range(2)))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    M = wb.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mixed_d2_T24.pb"))
    imgs = np.stack([synth_image(200, 264, 900 + b) for b in range(5)])
    ref = M.detect_batch_raw(imgs)
    for r in range(2):
        got = np.frombuffer(out[r][0], nat.DET_DTYPE)
        assert np.array_equal(got["image"], ref["image"]) and np.array_equal(got["level"], ref["level"])
        assert np.array_equal(got["r"], ref["r"].astype(np.uint16)) and np.array_equal(got["c"], ref["c"].astype(np.uint16))
        assert np.array_equal(got["score"].view(np.uint32), ref["scores"].view(np.uint32))
    assert out[0][1][0] == 3 and out[1][1][0] == 2          # shard sizes 3 + 2
