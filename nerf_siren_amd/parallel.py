"""Data-parallel glue: one process per GPU, rays sharded across ranks, one
all-reduce(mean) of the flat fp32 gradient per step (RCCL over xGMI).

Counterpart of train.py:48-49 (Lightning DDP) for the renderer: the only exchange
step of the path is the parameter gradient (2 x 595 844 fp32 = 4 766 752 B).  Both
models' gradients live in ONE contiguous buffer (FlatGradAllReduce.joint: the HIP backward
of model i writes slice i, every p.grad is a view into it),
so the collective is a single ncclAllReduce -- at this size RCCL is latency bound
on the fully connected xGMI mesh, and one call beats per-tensor or bucketed calls.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def flat_grad_alias(params):
    """If the gradients of `params` are consecutive slices of ONE allocation, in this order -- which is how the
    HIP backward writes them (ops.flat_views) -- return a 1-D tensor aliasing that memory (no copy), else None.
    (autograd hands p.grad over detached, so `._base` is not available: the storage and the offsets are compared.)"""
    gs = [p.grad for p in params]
    if not gs or any(g is None or not g.is_contiguous() or g.dtype != torch.float32 for g in gs):
        return None
    st = gs[0].untyped_storage()
    off = gs[0].storage_offset()
    start = off
    for g in gs:
        if g.untyped_storage().data_ptr() != st.data_ptr() or g.storage_offset() != off:
            return None
        off += g.numel()
    return torch.empty(0, device=gs[0].device, dtype=torch.float32).set_(st, start, (off - start,))


class FlatGradAllReduce:
    """all-reduce(mean) of the models' gradients with as few collectives as storage allows.

    The HIP backward writes a model's 24 gradients into one contiguous buffer (ops.flat_views), so
    after loss.backward() every p.grad of a NeRF is a view of the same base tensor: that base is
    all-reduced directly (one collective per model, no copies).  Gradients that do not share a base
    (foreign modules, accumulated grads) are flattened into a scratch buffer instead.
    """

    def __init__(self, models, world_size: int | None = None, group=None):
        self.models = list(models)
        self.params = [p for m in self.models for p in m.parameters() if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.group = group
        self.flat = None
        # ONE buffer for the gradients of all models: the HIP backward of model i writes into slice i
        # (rendering.FieldRender reads model._grad_target), so the step's exchange is a single collective
        self.joint = None
        try:
            from . import ops
            ps0 = [p for m in self.models for p in m.parameters()]
            if ps0 and all(hasattr(m, "param_list") and sum(p.numel() for p in m.parameters()) == ops.PARAM_NUMEL
                           for m in self.models) and all(p.is_cuda for p in ps0):
                n = ops.PARAM_NUMEL
                self.joint = torch.empty(len(self.models) * n, device=ps0[0].device, dtype=torch.float32)
                for i, m in enumerate(self.models):
                    m._grad_target = self.joint[i * n:(i + 1) * n]
        except Exception:                       # foreign modules: per-model / loose paths below
            self.joint = None

    def _bases(self):
        bases, loose = [], []
        for m in self.models:
            ps = list(m.param_list()) if hasattr(m, "param_list") else list(m.parameters())
            ps = [p for p in ps if p.requires_grad and p.grad is not None]
            if not ps:
                continue
            b = flat_grad_alias(ps)
            if b is not None:
                bases.append(b)
            else:
                loose.extend(ps)
        return bases, loose

    def zero_(self):
        for p in self.params:
            p.grad = None

    def all_reduce(self, average: bool = True):
        """mean over ranks, in place (DDP semantics). Returns the reduced buffers.
        average=False leaves the SUM in the buffers: the caller folds 1/world into the optimizer
        (training.FusedAdam.step(grad_scale=1/world)) and saves a pass over the gradients."""
        bases, loose = self._bases()
        if self.joint is not None and not loose and len(bases) == len(self.models) and all(
                b.data_ptr() == m._grad_target.data_ptr() for b, m in zip(bases, self.models)):
            if self.world > 1:                  # every model's gradients sit in the joint buffer: one collective
                dist.all_reduce(self.joint, op=dist.ReduceOp.SUM, group=self.group)
                if average:
                    self.joint.mul_(1.0 / self.world)
            return [self.joint]
        if self.world > 1:
            for b in bases:
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
                if average:
                    b.mul_(1.0 / self.world)
            if loose:
                flat = torch.cat([p.grad.reshape(-1) for p in loose])
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                if average:
                    flat.mul_(1.0 / self.world)
                off = 0
                for p in loose:
                    p.grad.copy_(flat[off:off + p.grad.numel()].view_as(p.grad))
                    off += p.grad.numel()
        return bases


def shard_rays(n_total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a ray buffer for this rank (rays are independent units)."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)
