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

    The HIP backward writes a model's gradients into one contiguous buffer (model.grad_views), so after
    loss.backward() every p.grad of a model is a view of the same base tensor: that base is all-reduced directly (no
    copies).  All models share ONE joint buffer (model i owns slice i), so without overlap the step's exchange is a
    single collective.  Gradients that do not share a base (foreign modules, accumulated grads) are flattened into a
    scratch buffer instead.

    overlap=True (world > 1): the two field passes of render_rays are independent in the backward (rendering.py:243-245
    .detach()), and autograd runs the fine pass first.  As soon as a model's backward has ENQUEUED its gradient kernels,
    its slice is all-reduced asynchronously (RCCL's own stream waits for the compute stream at that point), so the fine
    model's collective runs under the coarse model's backward; all_reduce() then only waits.  This needs every model to be
    applied at most once per backward pass (a second application is refused loudly: its contribution would be added to
    a buffer that is already on the wire).
    """

    def __init__(self, models, world_size: int | None = None, group=None, overlap: bool = False):
        self.models = list(models)
        self.params = [p for m in self.models for p in m.parameters() if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.group = group
        self.flat = None
        self.overlap = bool(overlap) and self.world > 1
        self._works = {}
        # ONE buffer for the gradients of all models: the HIP backward of model i writes into slice i
        # (rendering._claim_grad_target reads model._grad_target)
        self.joint = None
        ps0 = [p for m in self.models for p in m.parameters()]
        if ps0 and len({p.device for p in ps0}) == 1 and all(
                hasattr(m, "param_list") and hasattr(m, "grad_views")
                and sum(p.numel() for p in m.param_list()) == getattr(m, "grad_numel", -1) for m in self.models):
            # the models of this package (NeRF: 595 844 floats, SirenField: 529 156): slice i of the joint buffer
            sizes = [m.grad_numel for m in self.models]
            self.joint = torch.empty(sum(sizes), device=ps0[0].device, dtype=torch.float32)
            off = 0
            for i, (m, n) in enumerate(zip(self.models, sizes)):
                m._grad_target = self.joint[off:off + n]
                off += n
                # a model belongs to the LAST reducer built over it: an earlier overlapping reducer's hook must not keep
                # launching collectives into that reducer's stale bookkeeping
                m._grad_ready_hook = (lambda target, i=i: self._launch(i, target)) if self.overlap else None
                m._grad_target_claimed = False

    def _launch(self, i, target):
        """Called by the model's backward right after its gradient kernels were enqueued into `target`."""
        if i in self._works:
            raise RuntimeError("FlatGradAllReduce(overlap=True): a model was applied more than once in one backward "
                               "pass; its gradient slice is already being reduced")
        self._works[i] = dist.all_reduce(target, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _bases(self):
        bases, loose = [], []
        for m in self.models:
            ps = list(m.param_list()) if hasattr(m, "param_list") else list(m.parameters())
            listed = {id(p) for p in ps}
            ps = [p for p in ps if p.requires_grad and p.grad is not None]
            # trainable tensors outside the flat layout (a SirenField's conditioning rows): reduced with the loose ones
            extra = [p for p in m.parameters() if id(p) not in listed and p.requires_grad and p.grad is not None]
            loose.extend(extra)
            if not ps:
                continue
            b = flat_grad_alias(ps)
            if b is not None:
                bases.append(b)
            else:
                loose.extend(ps)
        return bases, loose

    def reset(self):
        """Step boundary: forget collectives launched by a backward pass that never reached all_reduce() (an exception
        mid-backward) and release the models' gradient targets, so the next step starts clean instead of failing with
        'applied more than once'."""
        for w in self._works.values():
            try:
                w.wait()
            except Exception:
                pass
        self._works = {}
        for m in self.models:
            if hasattr(m, "_grad_target_claimed"):
                m._grad_target_claimed = False

    def zero_(self):
        self.reset()
        for p in self.params:
            p.grad = None

    def all_reduce(self, average: bool = True):
        """mean over ranks, in place (DDP semantics). Returns the reduced buffers.
        average=False leaves the SUM in the buffers: the caller folds 1/world into the optimizer
        (training.FusedAdam.step(grad_scale=1/world)) and saves a pass over the gradients."""
        bases, loose = self._bases()
        works, self._works = self._works, {}
        if self.joint is not None and len(bases) == len(self.models) and all(
                b.data_ptr() == m._grad_target.data_ptr() for b, m in zip(bases, self.models)):
            self._reduce_loose(loose, average)
            if self.world > 1:
                if works:                       # slices already on the wire: wait for them, reduce the others now
                    for i, m in enumerate(self.models):
                        if i in works:
                            works[i].wait()     # the compute stream waits for the collective
                        else:
                            dist.all_reduce(m._grad_target, op=dist.ReduceOp.SUM, group=self.group)
                else:                           # every model's gradients sit in the joint buffer: one collective
                    dist.all_reduce(self.joint, op=dist.ReduceOp.SUM, group=self.group)
                if average:
                    self.joint.mul_(1.0 / self.world)
            return [self.joint]
        if works:
            raise RuntimeError("FlatGradAllReduce(overlap=True): a gradient slice was reduced early but the parameters' "
                               "gradients no longer alias it (accumulated or chunked backward)")
        if self.world > 1:
            for b in bases:
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
                if average:
                    b.mul_(1.0 / self.world)
            self._reduce_loose(loose, average)
        return bases

    def _reduce_loose(self, loose, average):
        if self.world <= 1 or not loose:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in loose])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        if average:
            flat.mul_(1.0 / self.world)
        off = 0
        for p in loose:
            p.grad.copy_(flat[off:off + p.grad.numel()].view_as(p.grad))
            off += p.grad.numel()


def shard_rays(n_total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a ray buffer for this rank (rays are independent units).  When world does not divide
    n_total the last ranks get fewer rays (possibly none): a mean-of-per-rank-means is then NOT the batch mean -- weight
    each rank's mean loss with shard_loss_weight(), or use shard_indices(pad=True) for the reference's DistributedSampler
    behaviour."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def shard_loss_weight(n_total: int, rank: int, world: int) -> float:
    """Factor for this rank's MEAN loss so that the all-reduce(mean) of the gradients equals the gradient of the mean over
    all n_total rays:  (1/world) sum_r w_r mean_r = (1/n_total) sum_r sum_{i in r} l_i  with  w_r = n_r * world / n_total.
    1.0 for every rank when world divides n_total; 0.0 for a rank whose shard is empty (it still takes part in the
    collective -- with zero gradients)."""
    lo, hi = shard_rays(n_total, rank, world)
    return (hi - lo) * world / n_total if n_total > 0 else 0.0


def shard_indices(n_total: int, rank: int, world: int, pad: bool = True):
    """Ray indices of this rank the way torch.utils.data.DistributedSampler(shuffle=False) deals them (the reference trains
    under Lightning DDP, train.py:41-66, whose sampler pads the index list by wrapping around until world divides it and
    gives rank r the indices r, r + world, r + 2 world, ...): every rank gets ceil(n_total / world) rays, up to world - 1
    rays are seen twice per epoch, and mean-of-means is the mean over the padded list.  pad=False: the same interleaving
    without the padding (uneven; combine with shard_loss_weight-style weighting by len())."""
    idx = list(range(n_total))
    if pad and n_total > 0 and n_total % world:
        need = (n_total + world - 1) // world * world - n_total
        idx += (idx * ((need + n_total - 1) // n_total + 1))[:need]
    return idx[rank::world]
