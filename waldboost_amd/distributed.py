"""Multi-GPU sharding of image batches and the gather of detections.

The path shards embarrassingly (images are independent): one process per GPU,
each scans its own contiguous chunk of the batch with its own resident cascade; no collective on
the data path.  The exchange at the end is small and latency-bound (16 B per detection):

  * one all-gather of three integers per rank (largest shard fill, valid records, images) -- every
    rank learns whether ANY rank overflowed its detection buffer, and all of them then grow to the
    same capacity and scan again together (a rank never re-allocates on its own);
  * one gather of the valid record prefixes, padded to the longest, to the destination rank;
  * one all-reduce (sum) of the per-level alive[level, stage] counters -- the reference's additive
    n_loc / n_weak statistics (model.py:248,252).

RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests.  `DetectionGatherer` is the sync-free
variant bench.py overlaps with the next step: a fixed-size all-gather with no host read-back.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous chunk [lo, hi) of n_items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DetectionGatherer:
    """The sync-free gather bench.py overlaps with the next step: every rank's PACKED detections (a 4-word header
    + the valid records back to back, wb_det_pack_launch / PyramidEngine.pack) are all-gathered as a fixed-size
    prefix of `rows` records -- sized once from what the workload produces, not from the buffer's capacity -- with
    no host read-back of counts.  `merged` tells from the headers whether a prefix was too short."""

    def __init__(self, rows, device, group=None):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.rows = int(rows)                                     # records per rank in the collective (+ 1 header row)
        self.recv = torch.zeros((self.world * (1 + self.rows), 4), dtype=torch.int32, device=device)
        # gloo has no all_gather for device tensors: a rehearsal of the multi-rank loop on one GPU
        # (bench.py --backend gloo) stages through the host; the real path is RCCL ("nccl")
        self._host_staged = torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"

    def gather(self, packed, async_op=False):
        """packed: int32 [>= 1 + rows, 4] as wb_det_pack_launch writes it."""
        send = packed[: 1 + self.rows]
        if self._host_staged:
            import torch
            recv = torch.empty(self.recv.shape, dtype=torch.int32)
            self.dist.all_gather_into_tensor(recv, send.cpu(), group=self.group)
            self.recv.copy_(recv)
            return None
        return self.dist.all_gather_into_tensor(self.recv, send, group=self.group, async_op=async_op)

    def merged(self, images_per_rank):
        """Host-side merge on any rank: records of all ranks with image indices made global
        (rank r's local image i -> sum(images_per_rank[:r]) + i), in reference order.  Raises if a rank's
        detection buffer overflowed or its detections did not fit the gathered prefix."""
        from ._native import DET_DTYPE
        recv = self.recv.cpu().numpy().reshape(self.world, 1 + self.rows, 4)
        parts, base = [], 0
        for r in range(self.world):
            total, worst, present, cap = (int(x) for x in recv[r, 0])
            if worst > cap:
                raise OverflowError(f"rank {r}: a detection shard holds {worst} records, capacity {cap}")
            if total > self.rows:
                raise OverflowError(f"rank {r}: {total} detections, the gathered prefix holds {self.rows}")
            d = recv[r, 1: 1 + total].copy().view(DET_DTYPE).reshape(-1)
            d["image"] += base
            parts.append(d)
            base += int(images_per_rank[r])
        out = np.concatenate(parts) if parts else np.zeros(0, DET_DTYPE)
        order = np.lexsort((out["c"], out["r"], out["level"], out["image"]))
        return out[order]


class RoundGatherer:
    """DetectionGatherer for a pipeline of `slots` engines per rank: the engines pack straight into their slot of ONE
    send buffer (PyramidEngine.pack(out=gatherer.send[set][slot])) and ONE all_gather moves all slots of a round --
    one collective launch per `slots` steps instead of one per step (the per-step launch, ~50 us of host time with its
    events, made the multi-rank step host-bound).  Two buffer sets alternate between rounds, so the packs of round
    r + 1 never wait for the collective of round r."""

    SETS = 2

    def __init__(self, rows, slots, device, group=None):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.rows, self.slots = int(rows), int(slots)
        self.send = [torch.zeros((self.slots, 1 + self.rows, 4), dtype=torch.int32, device=device) for _ in range(self.SETS)]
        self.recv = [torch.zeros((self.world, self.slots, 1 + self.rows, 4), dtype=torch.int32, device=device)
                     for _ in range(self.SETS)]
        self._host_staged = torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"

    def gather(self, which):
        """All ranks' send[which] -> recv[which] on every rank (on the current stream; no host synchronisation with RCCL)."""
        if self._host_staged:                      # (gloo rehearsal on one GPU: staged through the host)
            import torch
            recv = torch.empty((self.recv[which].numel() // 4, 4), dtype=torch.int32)
            self.dist.all_gather_into_tensor(recv, self.send[which].view(-1, 4).cpu(), group=self.group)
            self.recv[which].view(-1, 4).copy_(recv)
            return
        # (flat [n, 4] views: the ranks' blocks concatenate along dim 0, the form every backend takes)
        self.dist.all_gather_into_tensor(self.recv[which].view(-1, 4), self.send[which].view(-1, 4), group=self.group)

    def merged(self, which, slot, images_per_rank):
        """Host-side merge of one slot of a gathered round, as DetectionGatherer.merged."""
        from ._native import DET_DTYPE
        recv = self.recv[which][:, slot].cpu().numpy()
        parts, base = [], 0
        for r in range(self.world):
            total, worst, present, cap = (int(x) for x in recv[r, 0])
            if worst > cap:
                raise OverflowError(f"rank {r}: a detection shard holds {worst} records, capacity {cap}")
            if total > self.rows:
                raise OverflowError(f"rank {r}: {total} detections, the gathered prefix holds {self.rows}")
            d = recv[r, 1: 1 + total].copy().view(DET_DTYPE).reshape(-1)
            d["image"] += base
            parts.append(d)
            base += int(images_per_rank[r])
        out = np.concatenate(parts) if parts else np.zeros(0, DET_DTYPE)
        return out[np.lexsort((out["c"], out["r"], out["level"], out["image"]))]


# ------------------------------------------------------------------------------ the end-of-batch exchange
def _comm_device(group=None):
    """Where collective payloads live: the GPU for RCCL ("nccl"), the host for gloo."""
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def agree_capacity(scan, group=None):
    """Bring every rank's detection buffer to a capacity no rank overflows.  `scan` is the rank's scan state:
    ``scan.cap`` (records per shard), ``scan.need()`` (largest shard fill of the last scan; may exceed cap)
    and ``scan.grow(cap)`` (re-allocate to `cap` records per shard and scan again).  All ranks take the same
    decisions: the MAX over ranks of need and of cap is what every rank compares and grows to, so buffer sizes
    never diverge -- also when a single rank overflowed.  Returns the number of rounds (0 = nobody overflowed)."""
    import torch
    import torch.distributed as dist
    dev = _comm_device(group)
    rounds = 0
    while True:
        t = torch.tensor([int(scan.need()), int(scan.cap), -int(scan.cap)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        need, cap_max, cap_min = int(t[0]), int(t[1]), -int(t[2])
        if need <= cap_min and cap_min == cap_max:            # the same three numbers on every rank: same decision
            return rounds
        target = cap_max if need <= cap_max else int(need * 1.5) + 16
        if scan.cap != target or scan.need() > scan.cap:
            scan.grow(target)
        rounds += 1


def gather_records(recs, n_images, group=None, dst=0, first_image=None):
    """Gather every rank's valid detection records (int32 [n, 4] = WbDet rows with LOCAL image indices) to rank
    `dst`: an all-gather of (n, n_images) per rank, then one gather of the prefixes padded to the longest.
    Image indices become global: rank r's image i -> first_image[r] + i, by default the ranks' contiguous shards
    in rank order.  Returns on `dst` the merged WbDet array in reference order (image, level, r, c); on the
    other ranks None."""
    import torch
    import torch.distributed as dist
    from ._native import DET_DTYPE
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _comm_device(group)
    recs = torch.as_tensor(recs, dtype=torch.int32).reshape(-1, 4)
    meta = torch.zeros((world, 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(meta, torch.tensor([[recs.shape[0], int(n_images)]], dtype=torch.int64, device=dev), group=group)
    meta = meta.cpu().numpy()
    n_max = int(meta[:, 0].max())
    send = torch.zeros((max(n_max, 1), 4), dtype=torch.int32, device=dev)
    send[: recs.shape[0]] = recs.to(dev)
    dst_global = dist.get_global_rank(group, dst) if group is not None else dst
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst_global, group=group)
    if rank != dst:
        return None
    if first_image is None:
        first_image = np.concatenate([[0], np.cumsum(meta[:-1, 1])])
    parts = []
    for r in range(world):
        d = bufs[r][: int(meta[r, 0])].cpu().numpy().copy().view(DET_DTYPE).reshape(-1)
        d["image"] += int(first_image[r])
        parts.append(d)
    out = np.concatenate(parts)
    return out[np.lexsort((out["c"], out["r"], out["level"], out["image"]))]


def reduce_alive(alive_levels, group=None):
    """Sum of alive[level, stage] over the ranks (int64), on every rank."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.ascontiguousarray(alive_levels, np.int64)).to(_comm_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


class _EngineScan:
    """agree_capacity's view of a PyramidEngine + cascade."""

    def __init__(self, eng, dm):
        self.eng, self.dm = eng, dm
        self.stt = eng.run(dm)

    @property
    def cap(self):
        return self.eng.detb.cap

    def need(self):
        return self.eng.detb.max_count()

    def grow(self, cap):
        from . import _native as nat
        self.eng.det_capacity = int(cap) * nat.WB_DET_SHARDS
        self.eng._alloc_det()
        self.stt = self.eng.run_cascade(self.dm, ranks=self.stt.get("ranks", False))


def detect_sharded(model, images, group=None, dst=0):
    """Detect on a batch that is split over the ranks of `group` (one process per GPU).  `images` is THIS rank's
    shard [b, H, W] (host array or device tensor; b may differ between ranks, the shards are consecutive in rank
    order -- `shard_range` cuts a global batch that way).  Every rank scans its shard with no data-path
    collective; then the exchange described in the module docstring.  Returns

        det    on rank `dst`: all detections as a WbDet array with GLOBAL image indices in reference order
               (image, level, r, c); None on the other ranks
        alive  [b, levels, stages] of the local shard
        total  [levels, stages] summed over all ranks (every rank)

    and adds the GLOBAL n_loc / n_weak to the model's counters on every rank (reference model.py:248,252 are
    plain sums over the images scanned)."""
    from . import engine as _engine
    from . import channels as _channels
    b = int(images.shape[0])
    shrink, n_per_oct, smooth, spec = _channels.read_opts(model.channel_opts)
    H, W = int(images.shape[1]), int(images.shape[2])
    dtype = _engine.array_dtype(images)               # (NotImplementedError before any collective: the same on every rank)
    dm = model.device_cascade()
    T = len(model)
    eng = _engine.get_engine(H, W, dtype, shrink, n_per_oct, smooth, max(b, 1), channels=spec)
    if eng.plan.n_levels == 0:
        # images smaller than the window: no level, no window, nothing to exchange -- the plan depends on (H, W,
        # channel_opts) only, so every rank returns here together (reference model.py:171-177 yields nothing either)
        from ._native import DET_DTYPE
        import torch.distributed as dist
        rank = dist.get_rank(group)
        return (np.zeros(0, DET_DTYPE) if rank == dst else None), np.zeros((b, 0, T), np.int64), np.zeros((0, T), np.int64)
    if b:
        eng.load_images(images)
    scan = _EngineScan(eng, dm)                       # (a rank with an empty shard scans one blank image and drops it)
    agree_capacity(scan, group)
    L = eng.plan.n_levels
    alive = scan.stt["alive"][:b, :, :T].cpu().numpy().astype(np.int64).reshape(b, L, T)
    counts = eng.shard_counts(dm)
    recs = eng.detb.valid_records(counts).cpu() if b else np.zeros((0, 4), np.int32)
    det = gather_records(recs, b, group, dst)
    tot = reduce_alive(np.concatenate([alive.sum(axis=0).reshape(-1), [b]]), group)
    total, n_images = tot[:-1].reshape(L, T), int(tot[-1])
    model.n_loc += n_images * eng.plan.n_loc(dm.m, dm.n)
    model.n_weak += int(total.sum())
    return det, alive, total
