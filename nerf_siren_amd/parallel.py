"""Data-parallel glue: one process per GPU, rays sharded across ranks, one
all-reduce(mean) of the flat fp32 gradient per step (RCCL over xGMI).

Counterpart of train.py:48-49 (Lightning DDP) for the renderer: the only exchange
step of the path is the parameter gradient (2 x 595 844 fp32 = 4 766 752 B).  Both
models' gradients live in ONE contiguous buffer (every p.grad is a view into it),
so the collective is a single ncclAllReduce -- at this size RCCL is latency bound
on the fully connected xGMI mesh, and one call beats per-tensor or bucketed calls.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, models, world_size: int | None = None, group=None):
        self.params = [p for m in models for p in m.parameters() if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.group = group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def all_reduce(self):
        """mean over ranks, in place (DDP semantics)."""
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / self.world)
        return self.flat


def shard_rays(n_total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a ray buffer for this rank (rays are independent units)."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)
