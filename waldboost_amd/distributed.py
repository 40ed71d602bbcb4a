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
    """All-gathers the first `cap`+1 rows of each rank's detection buffer (row 0 = header with
    the count, rows 1.. = WbDet records as 4 int32 words)."""

    def __init__(self, cap, device, group=None):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.cap = int(cap)
        self.recv = torch.zeros((self.world, self.cap + 1, 4), dtype=torch.int32, device=device)

    def gather(self, det_buf, async_op=False):
        """det_buf: int32 [>=cap+1, 4] (PyramidEngine.det_buf).  Returns the work handle (or None)."""
        send = det_buf[: self.cap + 1]
        return self.dist.all_gather_into_tensor(self.recv.view(-1, 4), send.contiguous(), group=self.group,
                                                async_op=async_op)

    def merged(self, images_per_rank):
        """Host-side merge on any rank: records of all ranks with image indices made global
        (rank r's local image i -> sum(images_per_rank[:r]) + i).  Raises if a rank overflowed
        the gathered prefix."""
        from ._native import DET_DTYPE
        recv = self.recv.cpu().numpy()
        parts, base = [], 0
        for r in range(self.world):
            n = int(recv[r, 0, 0]) & 0xFFFFFFFF
            if n > self.cap:
                raise OverflowError(f"rank {r} produced {n} detections, gather prefix holds {self.cap}")
            d = recv[r, 1:1 + n].copy().view(DET_DTYPE).reshape(-1)
            d["image"] += base
            parts.append(d)
            base += int(images_per_rank[r])
        out = np.concatenate(parts) if parts else np.zeros(0, DET_DTYPE)
        order = np.lexsort((out["c"], out["r"], out["level"], out["image"]))
        return out[order]
