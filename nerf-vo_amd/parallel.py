"""Multi-GPU data parallelism for the mapping step: one process per MI355X, rays sharded across
ranks (each rank draws its own 4096 rays from its replica of the keyframe buffer), parameters
replicated, and exactly ONE exchange per iteration -- a sum all-reduce of the flat gradient buffer
over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests).  The
reference has no distributed path; this is the exchange section 8e specifies.

The flat fp32 gradient buffer (13.85 M scalars, 55.4 MB) is reduced in a few large buckets rather than
per-parameter: xGMI is point-to-point (7 links x ~153 GB/s), so the ring cost is per-link-bandwidth
bound (~0.63 ms for 55 MB at 8 GPUs) and small messages only add latency.  Buckets are issued
asynchronously so that later buckets overlap earlier ones' completion.
"""
from __future__ import annotations

import torch


class GradientAllReduce:
    def __init__(self, dist_module, bucket_numel: int = 4 * 1024 * 1024, group=None):
        self.dist = dist_module
        self.bucket_numel = int(bucket_numel)
        self.group = group
        self.world_size = dist_module.get_world_size(group) if dist_module.is_initialized() else 1

    def __call__(self, flat_grad: torch.Tensor, segments=None) -> None:
        """In-place sum over ranks.  ``segments``: optional iterable of (offset, size) ranges to
        reduce (e.g. skip the proposal networks on steps where they are not updated)."""
        if not self.dist.is_initialized():
            return  # single process without a process group: identity
        ranges = [(0, flat_grad.numel())] if segments is None else list(segments)
        handles = []
        for off, size in ranges:
            for lo in range(off, off + size, self.bucket_numel):
                hi = min(off + size, lo + self.bucket_numel)
                handles.append(self.dist.all_reduce(flat_grad[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group,
                                                    async_op=True))
        for h in handles:
            h.wait()


def shard_ray_count(global_rays: int, world_size: int, rank: int) -> int:
    """Strong-scaling helper: number of rays rank ``rank`` owns out of a fixed global batch."""
    base, rem = divmod(global_rays, world_size)
    return base + (1 if rank < rem else 0)
