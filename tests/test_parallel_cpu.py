"""gloo tests of the data-parallel glue (run on CPU; world 2 and 8): sharded rays + flat-gradient all-reduce(mean) ==
single-process gradient of the mean loss, for ray counts the world does and does not divide."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_siren_amd.parallel import FlatGradAllReduce, shard_indices, shard_loss_weight, shard_rays


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _models():
    torch.manual_seed(0)
    return [torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3)),
            torch.nn.Linear(8, 3)]


def _loss(models, rays):
    return sum(((m(rays)) ** 2).mean() for m in models)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1)
    rays = torch.randn(64, 8)
    lo, hi = shard_rays(64, rank, world)
    models = _models()
    red = FlatGradAllReduce(models, world)
    red.zero_()
    _loss(models, rays[lo:hi]).backward()
    # model 1: gradients living in one flat base (what the HIP backward produces); model 0: loose tensors
    flat1 = torch.cat([p.grad.reshape(-1) for p in models[1].parameters()])
    off = 0
    for p in models[1].parameters():
        p.grad = flat1[off:off + p.numel()].view_as(p).detach()     # autograd hands gradients over detached (no ._base)
        off += p.numel()
    bases = red.all_reduce()
    assert len(bases) == 1 and bases[0].data_ptr() == flat1.data_ptr() and bases[0].numel() == flat1.numel()
    flat = torch.cat([p.grad.reshape(-1) for m in models for p in m.parameters()])
    q.put((rank, flat.numpy()))       # by value: a tensor travels as a file descriptor the parent must fetch while this process lives
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_grad_allreduce_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: torch.from_numpy(v) for r, v in (q.get(timeout=90) for _ in range(world))}
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    torch.manual_seed(1)
    rays = torch.randn(64, 8)
    models = _models()
    _loss(models, rays).backward()              # equal shards -> mean of shard losses == full mean
    ref = torch.cat([p.grad.reshape(-1) for m in models for p in m.parameters()])
    for r in range(world):
        assert torch.allclose(got[r], ref, atol=1e-6), r
    assert torch.equal(got[0], got[1])


def _worker_uneven(rank, world, port, q, n, mode):
    """mode 'weight': contiguous uneven shards, each rank's mean loss weighted by shard_loss_weight (== the batch mean);
    mode 'pad': DistributedSampler semantics (wrap-around padding, == the mean over the padded index list)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    torch.manual_seed(1)
    rays = torch.randn(n, 8)
    models = _models()
    red = FlatGradAllReduce(models, world)
    red.zero_()
    if mode == "weight":
        lo, hi = shard_rays(n, rank, world)
        if hi > lo:
            (_loss(models, rays[lo:hi]) * shard_loss_weight(n, rank, world)).backward()
        else:                                                     # an empty shard still takes part in the collective
            for m in models:
                for p in m.parameters():
                    p.grad = torch.zeros_like(p)
    else:
        _loss(models, rays[shard_indices(n, rank, world)]).backward()
    red.all_reduce()
    q.put((rank, torch.cat([p.grad.reshape(-1) for m in models for p in m.parameters()]).numpy()))   # by value (see _worker)
    dist.barrier()
    dist.destroy_process_group()


def _run_world(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: torch.from_numpy(v) for r, v in (q.get(timeout=150) for _ in range(world))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,n", [(8, 64), (8, 61), (2, 13), (8, 5)])
def test_uneven_and_eight_rank_shards(world, n):
    """Round-2 verdict (weak #8): parallel.shard_rays gives the last rank fewer rays when world does not divide n, and
    mean-of-per-rank-means then differs from the batch mean.  Both remedies, at world 2 and at world 8 (eight gloo ranks on
    the CPU; (8, 5) leaves three ranks with EMPTY shards): (a) weighting each rank's mean loss by shard_loss_weight
    reproduces the single-process gradient of the mean over all n rays; (b) shard_indices(pad=True) reproduces what the
    reference's Lightning DDP does (DistributedSampler wrap-around padding): the mean over the padded index list."""
    def single(idx):
        torch.manual_seed(1)
        rays = torch.randn(n, 8)
        models = _models()
        _loss(models, rays[idx]).backward()
        return torch.cat([p.grad.reshape(-1) for m in models for p in m.parameters()])
    got = _run_world(_worker_uneven, world, n, "weight")
    ref = single(list(range(n)))
    for r in range(world):
        assert torch.allclose(got[r], ref, atol=2e-6), (r, float((got[r] - ref).abs().max()))
        assert torch.equal(got[r], got[0])                       # replicas stay identical
    got = _run_world(_worker_uneven, world, n, "pad")
    padded = [i for r in range(world) for i in shard_indices(n, r, world)]
    assert len(padded) == -(-n // world) * world
    ref = single(padded)
    for r in range(world):
        assert torch.allclose(got[r], ref, atol=2e-6), (r, float((got[r] - ref).abs().max()))


def test_shard_weights_and_padding():
    from torch.utils.data import DistributedSampler

    class _DS:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n
    for n, w in ((10, 3), (1023, 8), (5, 8), (64, 8), (8192, 8)):
        assert abs(sum(shard_loss_weight(n, r, w) for r in range(w)) - w) < 1e-9
        for r in range(w):
            assert list(DistributedSampler(_DS(n), num_replicas=w, rank=r, shuffle=False)) == shard_indices(n, r, w)
    assert all(shard_loss_weight(8192, r, 8) == 1.0 for r in range(8))


def test_shard_rays_partition():
    for n, w in ((1024, 8), (10, 3), (5, 8), (0, 2)):
        spans = [shard_rays(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


class _ToyModel(torch.nn.Module):
    """Stands in for NeRF on the CPU: one weight vector, gradients written into model._grad_target by the backward."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([1.0, 2.0, 3.0, 4.0]))
        self._grad_target = torch.zeros(4)

    grad_numel = 4

    def param_list(self):
        return [self.w]

    def grad_views(self, flat):
        return [flat[0:4]]


class _ToyRender(torch.autograd.Function):
    """Same write-into-the-shared-target protocol as rendering.FieldRender.backward."""

    @staticmethod
    def forward(ctx, model, scale, w):
        ctx.model, ctx.scale = model, scale
        return (w * scale).sum()

    @staticmethod
    def backward(ctx, g):
        from nerf_siren_amd.rendering import _claim_grad_target
        out = _claim_grad_target(ctx.model, torch.device("cpu"))
        val = torch.full((4,), float(ctx.scale)) * g
        if out is None:
            return None, None, val
        out[0].copy_(val)
        return None, None, out[0]


def test_grad_target_is_claimed_once_per_backward_pass():
    """A model applied twice in ONE graph (NeRFSystem.forward chunking, two render_rays calls in one loss): the second
    backward node must not overwrite the first contribution that autograd still holds as an alias of the shared target
    (round-1 advisor finding: [2,4,6,8]-style doubling instead of the sum)."""
    m = _ToyModel()
    loss = _ToyRender.apply(m, 1.0, m.w) + _ToyRender.apply(m, 10.0, m.w)
    loss.backward()
    assert torch.equal(m.w.grad, torch.full((4,), 11.0))
    assert not m._grad_target_claimed                      # released at the end of the pass
    # next pass, gradient cleared: the target is written again (no copy), and is the gradient's memory
    m.w.grad = None
    _ToyRender.apply(m, 3.0, m.w).backward()
    assert torch.equal(m.w.grad, torch.full((4,), 3.0)) and m.w.grad.data_ptr() == m._grad_target.data_ptr()
    # accumulation across passes (grad not cleared, still aliasing the target): must add, not overwrite
    _ToyRender.apply(m, 5.0, m.w).backward()
    assert torch.equal(m.w.grad, torch.full((4,), 8.0))


def test_reducer_replacement_and_failed_backward():
    """Round-2 advisor findings on FlatGradAllReduce: (1) a later non-overlapping reducer built over the same models must
    disarm the earlier reducer's grad-ready hook (it used to stay live and launch collectives into the stale reducer);
    (2) a backward pass that dies after the target was claimed / a collective launched must not poison the next step --
    zero_() / reset() is the step boundary that clears both."""
    m = _ToyModel()
    first = FlatGradAllReduce([m], world_size=2, overlap=True)
    assert first.overlap and m._grad_ready_hook is not None
    second = FlatGradAllReduce([m], world_size=1)
    assert m._grad_ready_hook is None and not second.overlap      # the model belongs to the LAST reducer built over it
    # a pass that raises after the claim: the engine callback that releases the claim never runs
    m._grad_target_claimed = True
    second._works = {0: type("W", (), {"wait": staticmethod(lambda: None)})()}
    second.zero_()
    assert not m._grad_target_claimed and second._works == {}
    _ToyRender.apply(m, 2.0, m.w).backward()
    assert torch.equal(m.w.grad, torch.full((4,), 2.0))
