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


def _to_host(t):
    """A device int32 [n, 4] tensor as a host ndarray that OWNS page-locked memory from torch's caching host allocator: one
    DMA, no second copy on the host (25 MB of records: ~1 ms; through a reused staging buffer the copy out of it cost
    another 2.5 ms).  The block goes back to the allocator's cache when the array is dropped."""
    import torch
    if t.device.type != "cuda":
        return t.numpy()
    buf = torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)
    buf.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return buf.numpy()                                             # (the array keeps the tensor's storage alive)


def gather_records(recs, n_images, group=None, dst=0, first_image=None, presorted=False):
    """Gather every rank's valid detection records (int32 [n, 4] = WbDet rows with LOCAL image indices; a device tensor
    stays on the device with RCCL) to rank `dst`: an all-gather of (n, n_images) per rank, then one gather of the
    prefixes padded to the longest.  Image indices become global: rank r's image i -> first_image[r] + i, by default
    the ranks' contiguous shards in rank order.  Returns on `dst` the merged WbDet array in reference order
    (image, level, r, c); on the other ranks None.
    presorted: every rank's records are already in that order (detect_sharded orders them on the device) -- with
    contiguous shards the merge is then the concatenation in rank order, no host sort."""
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
    if world == 1:
        bufs = [recs]                                          # (nothing to move)
    else:
        send = torch.zeros((max(n_max, 1), 4), dtype=torch.int32, device=dev)
        send[: recs.shape[0]] = recs.to(dev)
        dst_global = dist.get_global_rank(group, dst) if group is not None else dst
        bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, bufs, dst=dst_global, group=group)
    if rank != dst:
        return None
    contiguous = first_image is None
    if first_image is None:
        first_image = np.concatenate([[0], np.cumsum(meta[:-1, 1])])
    counts = [int(meta[r, 0]) for r in range(world)]
    if all(bf.device.type == "cuda" for bf in bufs) and sum(counts):
        # image indices made global on the device (rank r's block + first_image[r]), then ONE read-back of all ranks' records
        parts = []
        for r in range(world):
            blk = bufs[r][: counts[r]]
            if counts[r] and int(first_image[r]):
                blk = blk.clone() if world == 1 else blk       # (never the caller's own tensor)
                blk[:, 0] += int(first_image[r])
            parts.append(blk)
        allr = torch.cat(parts) if world > 1 else parts[0]
        out = np.ascontiguousarray(_to_host(allr)).view(DET_DTYPE).reshape(-1)
    else:
        host = np.concatenate([bufs[r][: counts[r]].cpu().numpy() for r in range(world)]) if world else np.zeros((0, 4), np.int32)
        out = np.ascontiguousarray(host).view(DET_DTYPE).reshape(-1)
        at = 0
        for r in range(world):
            out["image"][at: at + counts[r]] += int(first_image[r])
            at += counts[r]
    if presorted and contiguous:
        return out
    return out[np.lexsort((out["c"], out["r"], out["level"], out["image"]))]


def reduce_alive(alive_levels, group=None):
    """Sum of alive[level, stage] over the ranks (int64), on every rank."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.ascontiguousarray(alive_levels, np.int64)).to(_comm_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


_PAIR_ENGINES = {}
LAST_TIMING = {}            # detect_sharded's last call on this rank: scan_s (bench.py reports the per-rank share)


def _chunk_engine(H, W, dtype, shrink, n_per_oct, smooth, batch, spec, slot):
    """The engine a chunk of `batch` images is scanned on: two per configuration, so that consecutive chunks alternate
    between two streams (slot 0 is the package-wide cached engine of that configuration, slot 1 a second one)."""
    import torch
    from . import engine as _engine
    if slot == 0:
        return _engine.get_engine(H, W, dtype, shrink, n_per_oct, smooth, batch, channels=spec)
    key = (int(H), int(W), np.dtype(dtype).str, int(shrink), int(n_per_oct), int(smooth), int(batch), spec.key,
           torch.cuda.current_device())
    e = _PAIR_ENGINES.get(key)
    if e is None:
        if len(_PAIR_ENGINES) >= 2:
            _PAIR_ENGINES.pop(next(iter(_PAIR_ENGINES)))
        e = _PAIR_ENGINES[key] = _engine.PyramidEngine(H, W, dtype, shrink, n_per_oct, smooth, batch, channels=spec)
    return e


class _ShardScan:
    """A rank's scan of its shard, in chunks of `batch` images that alternate between two engines on two streams (chunk
    k + 1's image copy and first kernels overlap chunk k's tail); every chunk's valid records are packed on the device
    (wb_det_pack_launch) into a slot of their own.  Also agree_capacity's view of the rank: ``cap``, ``need()``, ``grow()``."""

    def __init__(self, model, images, batch, per_image_alive=True):
        import torch
        from . import engine as _engine
        from . import channels as _channels
        self.images = images
        self.b = b = int(images.shape[0])
        self.H, self.W = int(images.shape[1]), int(images.shape[2])
        self.opts = _channels.read_opts(model.channel_opts)
        self.dtype = _engine.array_dtype(images)          # (NotImplementedError before any collective: the same on every rank)
        self.dm = model.device_cascade()
        self.T = len(model)
        self.per_image_alive = per_image_alive
        batch = max(1, int(batch))
        self.bounds = [(lo, min(lo + batch, b)) for lo in range(0, b, batch)]
        shrink, n_per_oct, smooth, spec = self.opts
        mk = lambda size, slot: _chunk_engine(self.H, self.W, self.dtype, shrink, n_per_oct, smooth, size, spec, slot)
        # (an empty shard keeps a one-image engine: its capacity still takes part in the agreement)
        self.engines = [mk(hi - lo, k % 2) for k, (lo, hi) in enumerate(self.bounds)] or [mk(1, 0)]
        self.plan = self.engines[0].plan
        self.dev = self.engines[0].dev
        self.streams = [torch.cuda.Stream(), torch.cuda.Stream()] if len(self.bounds) > 1 else [torch.cuda.current_stream()]
        self.slots = None
        self.hdr = np.zeros((0, 4), np.int64)
        cap = max(e.detb.cap for e in set(self.engines))
        self._set_cap(cap)

    def _unique_engines(self):
        seen, out = set(), []
        for e in self.engines:
            if id(e) not in seen:
                seen.add(id(e))
                out.append(e)
        return out

    def _set_cap(self, cap):
        from . import _native as nat
        for e in self._unique_engines():
            if e.detb.cap != cap:
                e.det_capacity = int(cap) * nat.WB_DET_SHARDS
                e._alloc_det()
        self.slots = None

    @property
    def cap(self):
        return self.engines[0].detb.cap

    def run(self):
        """Scan every chunk; afterwards self.hdr[k] = (valid records, fullest shard, records present, shard capacity) of
        chunk k on the host -- ONE synchronisation."""
        import torch
        if self.plan.n_levels == 0 or not self.bounds:
            return
        L, T = self.plan.n_levels, self.T
        main = torch.cuda.current_stream()
        if self.slots is None:
            self.slots = [torch.empty((1 + e.detb.NS * e.detb.cap, 4), dtype=torch.int32, device=self.dev) for e in self.engines]
            self.hdr_d = torch.zeros((len(self.bounds), 4), dtype=torch.int32, device=self.dev)
            self.alive_sum = torch.zeros((len(self.streams), L, max(T, 1)), dtype=torch.int64, device=self.dev)
            self.alive_d = (torch.zeros((self.b, L, max(T, 1)), dtype=torch.int32, device=self.dev)
                            if self.per_image_alive else None)
        self.alive_sum.zero_()
        for st in self.streams:
            if st is not main:
                st.wait_stream(main)
        for k, ((lo, hi), eng) in enumerate(zip(self.bounds, self.engines)):
            st = self.streams[k % len(self.streams)]
            with torch.cuda.stream(st):
                eng.load_images(self.images[lo:hi])
                stt = eng.batch_enqueue(self.dm)
                eng.pack(out=self.slots[k])
                self.hdr_d[k].copy_(self.slots[k][0])
                if T:
                    al = stt["alive"][:, :, :T]
                    self.alive_sum[k % len(self.streams), :, :T] += al.sum(dim=0, dtype=torch.int64)
                    if self.alive_d is not None:
                        self.alive_d[lo:hi, :, :T] = al
        for st in self.streams:
            if st is not main:
                main.wait_stream(st)
        self.hdr = self.hdr_d.cpu().numpy().astype(np.int64)

    def need(self):
        return int(self.hdr[:, 1].max(initial=0))

    def grow(self, cap):
        self._set_cap(cap)
        self.run()

    def records(self):
        """This rank's valid records, int32 [n, 4] on the device, image indices local to the shard, in the reference's
        order (image, level, r, c)."""
        import torch
        from .engine import sort_records
        parts = []
        for k, (lo, hi) in enumerate(self.bounds):
            n = int(self.hdr[k, 0])
            if n:
                part = self.slots[k][1:1 + n]
                if lo:
                    part = part.clone()
                    part[:, 0] += lo
                parts.append(part)
        if not parts:
            return torch.zeros((0, 4), dtype=torch.int32, device=self.dev)
        return sort_records(parts[0] if len(parts) == 1 else torch.cat(parts))


def detect_sharded(model, images, group=None, dst=0, batch=64, per_image_alive=True):
    """Detect on a batch that is split over the ranks of `group` (one process per GPU).  `images` is THIS rank's
    shard [b, H, W] (host array or device tensor; b may differ between ranks, the shards are consecutive in rank
    order -- `shard_range` cuts a global batch that way).  Every rank scans its shard with no data-path collective, in
    chunks of `batch` images alternating between two engines on two streams (BASELINE configs[3]: 512 images over 8
    GPUs = one chunk of 64 per rank; a single GPU takes the same 512 as eight chunks); then the exchange described in
    the module docstring.  Returns

        det    on rank `dst`: all detections as a WbDet array with GLOBAL image indices in reference order
               (image, level, r, c); None on the other ranks
        alive  [b, levels, stages] of the local shard (None with per_image_alive=False: one read-back less)
        total  [levels, stages] summed over all ranks (every rank)

    and adds the GLOBAL n_loc / n_weak to the model's counters on every rank (reference model.py:248,252 are
    plain sums over the images scanned; the loop this replaces is scripts/waldboost-detect.py:64-67)."""
    import torch.distributed as dist
    from ._native import DET_DTYPE
    scan = _ShardScan(model, images, batch, per_image_alive)
    b, T, plan = scan.b, scan.T, scan.plan
    if plan.n_levels == 0:
        # images smaller than the window: no level, no window, nothing to exchange -- the plan depends on (H, W,
        # channel_opts) only, so every rank returns here together (reference model.py:171-177 yields nothing either)
        rank = dist.get_rank(group)
        return ((np.zeros(0, DET_DTYPE) if rank == dst else None),
                (np.zeros((b, 0, T), np.int64) if per_image_alive else None), np.zeros((0, T), np.int64))
    import time
    t0 = time.perf_counter()
    scan.run()
    LAST_TIMING.clear()
    LAST_TIMING["scan_s"] = time.perf_counter() - t0      # this rank's own scan, up to its one synchronisation
    agree_capacity(scan, group)
    LAST_TIMING["agree_s"] = time.perf_counter() - t0 - LAST_TIMING["scan_s"]
    L = plan.n_levels
    t1 = time.perf_counter()
    recs = scan.records()
    if recs.device.type == "cuda":
        import torch
        torch.cuda.current_stream().synchronize()
    LAST_TIMING["order_s"] = time.perf_counter() - t1
    t1 = time.perf_counter()
    det = gather_records(recs, b, group, dst, presorted=True)
    LAST_TIMING["gather_s"] = time.perf_counter() - t1
    alive = None
    if per_image_alive:
        alive = (scan.alive_d[:, :, :T].cpu().numpy().astype(np.int64) if b else np.zeros((0, L, T), np.int64)).reshape(b, L, T)
    mine = scan.alive_sum.sum(dim=0)[:, :T].cpu().numpy().reshape(-1) if (b and T) else np.zeros(L * T, np.int64)
    tot = reduce_alive(np.concatenate([mine, [b]]), group)
    total, n_images = tot[:-1].reshape(L, T), int(tot[-1])
    model.n_loc += n_images * plan.n_loc(scan.dm.m, scan.dm.n)
    model.n_weak += int(total.sum())
    return det, alive, total
