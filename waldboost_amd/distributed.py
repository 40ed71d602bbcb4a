"""Multi-GPU sharding of image batches and the gather of detections.

The path shards embarrassingly (images are independent): one process per GPU,
each scans a contiguous chunk of the batch with its own resident cascade; the only exchange is
one gather of the fixed-size detection records at the end (RCCL over xGMI with backend "nccl";
"gloo" in the CPU tests).  The payload is tiny (16 B per detection), so this is a single
latency-bound collective of a fixed-size prefix of each rank's detection buffer -- no ring, no
bucketing.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous chunk [lo, hi) of n_items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DetectionGatherer:
    """All-gathers every rank's sharded detection buffer (engine.DetBuffer.buf: counters in the
    first rows, WbDet records after) -- one fixed-size collective, no host read-back of counts."""

    def __init__(self, detb, group=None):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.NS, self.cap = detb.NS, detb.cap
        self.rows = detb.buf.shape[0]
        self.recv = torch.zeros((self.world * self.rows, 4), dtype=torch.int32, device=detb.buf.device)
        # gloo has no all_gather for device tensors: a rehearsal of the multi-rank loop on one GPU
        # (bench.py --backend gloo) stages through the host; the real path is RCCL ("nccl")
        self._host_staged = detb.buf.is_cuda and dist.get_backend(group) == "gloo"

    def gather(self, detb, async_op=False):
        if self._host_staged:
            import torch
            recv = torch.empty(self.recv.shape, dtype=torch.int32)
            self.dist.all_gather_into_tensor(recv, detb.buf.cpu(), group=self.group)
            self.recv.copy_(recv)
            return None
        return self.dist.all_gather_into_tensor(self.recv, detb.buf, group=self.group, async_op=async_op)

    def merged(self, images_per_rank):
        """Host-side merge on any rank: records of all ranks with image indices made global
        (rank r's local image i -> sum(images_per_rank[:r]) + i), in reference order.  Raises if
        a shard overflowed."""
        from ._native import DET_DTYPE
        recv = self.recv.cpu().numpy().reshape(self.world, self.rows, 4)
        parts, base = [], 0
        for r in range(self.world):
            counts = recv[r, : self.NS // 4].reshape(-1).view(np.uint32)
            if counts.max(initial=0) > self.cap:
                raise OverflowError(f"rank {r}: a detection shard holds {counts.max()} records, capacity {self.cap}")
            recs = recv[r, self.NS // 4:].reshape(self.NS, self.cap, 4)
            for s in range(self.NS):
                d = recs[s, : counts[s]].copy().view(DET_DTYPE).reshape(-1)
                d["image"] += base
                parts.append(d)
            base += int(images_per_rank[r])
        out = np.concatenate(parts) if parts else np.zeros(0, DET_DTYPE)
        order = np.lexsort((out["c"], out["r"], out["level"], out["image"]))
        return out[order]


def detect_sharded(model, images, group=None):
    """Detect on a batch that is split over the ranks of `group` (one process per GPU): every rank
    passes the SAME global batch [B,H,W] (host array) and scans its contiguous shard; the detection
    records of all ranks are then gathered with one collective.  Returns, on every rank, the merged
    records (numpy structured array of WbDet with GLOBAL image indices, reference order) and the
    per-image alive counts of the local shard."""
    import torch.distributed as dist
    from . import engine as _engine
    from . import channels as _channels
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = int(images.shape[0])
    lo, hi = shard_range(B, rank, world)
    counts = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
    nb = max(counts)                                   # every rank runs the same (padded) batch size
    shrink, n_per_oct, smooth, spec = _channels.read_opts(model.channel_opts)
    H, W = int(images.shape[1]), int(images.shape[2])
    eng = _engine.get_engine(H, W, images.dtype, shrink, n_per_oct, smooth, nb, channels=spec)
    local = np.zeros((nb, H, W), images.dtype)
    local[: hi - lo] = images[lo:hi]
    dm = model.device_cascade()
    eng.load_images(local)
    stt = eng.run(dm)
    eng.ensure_capacity(dm)
    g = DetectionGatherer(eng.detb, group)
    g.gather(eng.detb)
    merged = g.merged([nb] * world)                    # image index = rank * nb + local index
    # drop the padding images and renumber to the global batch
    keep = np.zeros(merged.size, bool)
    glob = np.zeros(merged.size, np.int32)
    for r in range(world):
        rlo = shard_range(B, r, world)[0]
        sel = (merged["image"] >= r * nb) & (merged["image"] < r * nb + counts[r])
        keep |= sel
        glob[sel] = merged["image"][sel] - r * nb + rlo
    out = merged[keep].copy()
    out["image"] = glob[keep]
    order = np.lexsort((out["c"], out["r"], out["level"], out["image"]))
    T = len(model)
    alive = stt["alive"][: hi - lo, :, :T].cpu().numpy().astype(np.int64)
    return out[order], alive
