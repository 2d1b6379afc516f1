#!/usr/bin/env python
"""Headline benchmark: ray-samples/s of render_rays() on Blender-lego-like rays
(400x400 pinhole, N_samples=64 + N_importance=64), BASELINE.json configs[1]
(batch 1024 rays per GPU), synthetic rays and seeded random-init weights.

One "step" = one pass of the hot path over one batch of 1024 rays per rank:
  --mode train (default): render_rays forward (coarse+fine, perturb=1,
      noise_std=1) + MSE(coarse)+MSE(fine) + backward through both MLPs +
      gradient all-reduce (RCCL, N>1; the fine model's slice goes on the wire while the
      coarse model's backward runs) + Adam step  -- system.py:257-275
  --mode infer: render_rays(test_time=True, perturb=0, noise_std=0) under
      no_grad -- eval.py:85-96
A ray-sample = one field evaluation: 64 coarse + 128 fine = 192 per ray.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (a child
`python -m torch.distributed.run`, before this process makes any GPU call) and relays rank 0's JSON line; under
torch.distributed.run it is one rank.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fine NeRF MLP forward, fp32 MFMA bound):
algorithmic FLOPs per launch / its average duration measured with HIP events on the launch stream inside the timed
region (ops.set_profile_hook).  Beside the headline, at N = 1: `infer` (the same field at test time), `siren` (the
FiLM-SIREN field of configs[1]: training and inference steps + the roofline of siren_forward_kernel), `eg3d`
(configs[4]: ImportanceRenderer forward / forward+backward / dense 128^3 query with the gather roofline of
triplane_kernel), `opt_in` (split-bf16 math), `psnr` (committed teacher-scene protocol) and `cpu_baseline` = the
reference's CPU PyTorch path restated op for op (oracle/torch_cpu_ref.py) on this box's physical cores, on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FULL = 1_186_816        # NeRF full forward per sample
FLOP_SIGMA = 982_528         # sigma-only forward per sample
FLOP_TRAIN = 3_489_024       # fwd + dW + dX per sample
FLOP_SIREN = 1_053_696       # FiLM-SIREN full forward per sample (+2 304 sin)
PEAK_F32_MFMA = 157.3        # TFLOP/s dense (MI355X_MICROARCH.md)
PEAK_HBM = 8.0               # TB/s
PEAK_L2_GATHER = 16.8        # TB/s: rows gathered from an L2-resident table, chip-wide lower bound (MI355X_MICROARCH.md "Indexed rows")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["train", "infer"], default="train")
    ap.add_argument("--batch", type=int, default=1024, help="rays per rank per step (configs[1])")
    ap.add_argument("--field", choices=["nerf", "siren"], default="nerf",
                    help="headline field: nerf = the reference's live 8x256 ReLU NeRF (system.py:183-190); siren = its "
                         "FiLM-SIREN field (always reported in the `siren` object at N = 1)")
    ap.add_argument("--math", choices=["fp32", "bf16x3"], default="fp32",
                    help="fp32 = exact fp32 MFMA (default); bf16x3 = opt-in split-bf16 math")
    ap.add_argument("--optimizer", choices=["fused", "torch"], default="fused",
                    help="training: nerf_siren_amd.training.FusedAdam + FusedMSELoss (one launch each) or torch.optim.Adam "
                         "+ elementwise loss -- same arithmetic")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one joint all-reduce after the whole backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true", help="skip the PSNR-parity check (tests/golden/g15_psnr.npz protocol)")
    ap.add_argument("--no-opt-in", action="store_true", help="skip the extra timed loop on the opt-in split-bf16 math")
    ap.add_argument("--no-extra", action="store_true", help="skip the infer / siren / eg3d objects")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work in the bounded cpu_baseline sample")
    return ap.parse_args()


# -------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks as children BEFORE anything here touches the GPU
# -------------------------------------------------------------------------------------------------
def spawn_ranks(n: int) -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# -------------------------------------------------------------------------------------------------
# CPU baseline: the reference's torch-CPU path (restated), all physical cores
# -------------------------------------------------------------------------------------------------
def cpu_baseline(mode, budget_s=12.0):
    from nerf_siren_amd import synth
    from oracle import torch_cpu_ref as TR
    ps = [synth.nerf_params(1, False), synth.nerf_params(2, False)]
    r = TR.timed_sample(mode, ps, lambda i: synth.blender_rays(1024, seed=123 + i),
                        lambda i: synth.hash_uniform((1024, 3), 5 + i), budget_s=budget_s, n_rays=1024)
    return {"value": r["ray_samples_per_s"], "unit": "ray-samples/s", "cores": r["threads"], "kind": "port",
            "sample": f"{r['steps']} steps of 1024 rays x (64+128) samples, {mode} step "
                      f"({'fwd+MSE+bwd+Adam' if mode == 'train' else 'test_time render'}), the reference's torch-CPU op "
                      f"sequence restated (oracle/torch_cpu_ref.py), {r['seconds']:.1f} s after one warm-up step",
            "best_step_value": r["ray_samples_per_s_best"], "cpu_model": r["cpu_model"],
            "physical_cores": r["physical_cores"], "nproc": r["nproc"], "allowed_cpus": r["allowed_cpus"],
            "cgroup_cpu_quota": r["cgroup_cpu_quota"], "thread_calibration_gemm_gflops": r["gemm_gflops"],
            "torch_parallel_info": r["parallel_info"].strip().replace("\n", "; "),
            "anchor": "the imported reference itself on 8 threads in the build container: 0.107-0.119 M ray-samples/s "
                      "training, 0.56-0.69 M inference (BASELINE.md section 2, oracle/torch_cpu_ref.py header)"}


def psnr_check(dev):
    """The metric's '+ PSNR': the reference trained on a teacher scene on CPU and its validation-PSNR trajectory is a
    committed fixture (tools/make_psnr_golden.py -> tests/golden/g15_psnr.npz); the same steps (same images, batches,
    injected draws, initial weights, learning-rate schedule) run here on the HIP path.  Rank 0 at N = 1 only."""
    import numpy as np
    import torch
    from nerf_siren_amd import Embedding, NeRF, render_rays, synth
    from nerf_siren_amd.training import FusedAdam, FusedMSELoss
    path = os.path.join(ROOT, "tests", "golden", "g19_psnr_long.npz")       # 1 500 steps, 400x400 validation view
    if not os.path.exists(path):
        path = os.path.join(ROOT, "tests", "golden", "g15_psnr.npz")        # 240 steps, 32x32
    if not os.path.exists(path):
        return None
    g = dict(np.load(path))
    S, F, B = int(g["cfg_S"]), int(g["cfg_F"]), int(g["cfg_batch"])
    steps, every = int(g["cfg_steps"]), int(g["cfg_eval_every"])
    milestones = [int(v) for v in g["cfg_lr_milestones"]] if "cfg_lr_milestones" in g else []
    gamma = float(g["cfg_lr_gamma"]) if "cfg_lr_gamma" in g else 1.0
    ms = []
    for seed in (11, 12):
        m = NeRF()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_params(seed, structured=False).items()})
        ms.append(m.to(dev))
    emb = [Embedding(3, 10), Embedding(3, 4)]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)          # noqa: E731
    rays_np, val_np = synth.psnr_rays(g)
    rays, tgt, val_rays, val_tgt = T(rays_np), T(g["target"]), T(val_np), T(g["val_target"])
    opt, loss_fn = FusedAdam(ms, lr=float(g["cfg_lr"]), eps=1e-8), FusedMSELoss(unit_grad=True)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=milestones, gamma=gamma) if milestones else None
    psnr = []
    for step in range(steps + 1):
        if step % every == 0:
            with torch.no_grad():
                mse, n = 0.0, 0
                for i in range(0, val_rays.shape[0], 1 << 15):
                    r = render_rays(ms, emb, val_rays[i:i + (1 << 15)], S, False, 0, 0, F, 1 << 15, True, False)
                    mse += float(((r["rgb_fine"] - val_tgt[i:i + (1 << 15)]) ** 2).sum())
                    n += r["rgb_fine"].numel()
            psnr.append(float(-10 * np.log10(mse / n)))
        if step == steps:
            break
        idx = torch.from_numpy(synth.psnr_batch_indices(step, rays.shape[0], B)).to(dev)
        rg = {k: T(v) for k, v in synth.psnr_step_rng(step, B, S, F).items()}
        res = render_rays(ms, emb, rays[idx], S, False, 1.0, 0.0, F, 1 << 15, True, False, rng=rg)
        loss = loss_fn(res, tgt[idx])
        opt.zero_grad()
        loss.backward()
        opt.step()
        if sched is not None:
            sched.step()
    ref = [float(v) for v in g["psnr"]]
    return {"value_db": psnr[-1], "reference_db": ref[-1], "max_abs_diff_db": float(np.abs(np.array(psnr) - ref).max()),
            "trajectory_db": [round(v, 3) for v in psnr], "reference_trajectory_db": [round(v, 3) for v in ref],
            "protocol": f"teacher scene, {steps} Adam steps of {B} rays ({S}+{F})"
                        + (f", lr x{gamma} at steps {milestones}" if milestones else "")
                        + f", validation every {every} steps on {val_rays.shape[0]} rays; reference = /root/reference on CPU (tests/golden/" + os.path.basename(path) + ")"}


class KernelTimer:
    """ops profiling hook: HIP events on the launch stream around the field-MLP launches of the fine pass."""

    def __init__(self, torch):
        self.torch = torch
        self.ev = {}

    class _Span:
        def __init__(self, owner, key):
            t = owner.torch
            self.a, self.b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
            self.a.record()
            owner.ev.setdefault(key, []).append((self.a, self.b))

        def done(self):
            self.b.record()

    def __call__(self, name, n_per_ray):
        if n_per_ray != 128:
            return None
        return KernelTimer._Span(self, name)

    def clear(self):
        self.ev.clear()

    def mean_ms(self, name):
        ev = self.ev.get(name)
        if not ev:
            return float("nan")
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)


def eg3d_bench(dev, steps, warmup):
    """configs[4]: EG3D tri-plane renderer, planes (1,3,32,256,256), M = 4096 rays x (64+64) samples, forward and
    forward+backward (gradients to planes and decoder), plus the dense 128^3 = 2 097 152-point run_model query."""
    import numpy as np
    import torch
    from nerf_siren_amd import ImportanceRenderer, OSGDecoder, synth
    planes = torch.from_numpy(synth.triplanes(1, res=256)).to(dev)
    dec = OSGDecoder(32, {"decoder_lr_mul": 1.0, "decoder_output_dim": 3})
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in synth.osg_params(1).items()})
    dec = dec.to(dev)
    ren = ImportanceRenderer()
    opts = dict(synth.EG3D_OPTIONS)
    M = 4096
    o, d = synth.eg3d_rays(M, 3)
    o, d = torch.from_numpy(o[None]).to(dev), torch.from_numpy(d[None]).to(dev)

    def timeit(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def fwd():
        with torch.no_grad():
            ren(planes, dec, o, d, opts)
    t_f = timeit(fwd)
    pl = planes.clone().requires_grad_(True)

    def fb():
        pl.grad = None
        for p_ in dec.parameters():
            p_.grad = None
        out = ren(pl, dec, o, d, opts)
        (out[3].square().mean() + out[4].mean() + out[0].square().mean()).backward()
    t_fb = timeit(fb)
    g = np.linspace(-1.5, 1.5, 128, dtype=np.float32)
    pts = torch.from_numpy(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(1, -1, 3)).to(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kms = []

    def dense():
        with torch.no_grad():
            a.record()
            ren.run_model(planes, dec, pts, None, opts)
            b.record()
        torch.cuda.synchronize()
        kms.append(a.elapsed_time(b))
    for _ in range(warmup):
        dense()
    kms.clear()
    for _ in range(steps):
        dense()
    t_d = float(np.mean(kms)) * 1e-3
    npts = pts.shape[1]
    gather = npts * (12 + 1536 + 16) / t_d / 1e12            # R 12 B coords + 1 536 B texels / W 16 B per sample
    return {"workload": "configs[4]: planes (1,3,32,256,256), M=4096 rays x (64+64) samples; dense query 128^3 points",
            "forward_ms": t_f * 1e3, "forward_samples_per_s": M * 128 / t_f,
            "forward_backward_ms": t_fb * 1e3, "forward_backward_samples_per_s": M * 128 / t_fb,
            "dense_query_ms": t_d * 1e3, "dense_points_per_s": npts / t_d,
            "roofline": {"bound": "l2-gather", "kernel": "triplane_kernel<1,false> (dense run_model: 12 bilinear taps x 32 ch "
                         "gathered per point from the 25 MB channels-last plane stack, decoder fused)", "achieved": gather,
                         "peak": PEAK_L2_GATHER, "unit": "TB/s", "frac": gather / PEAK_L2_GATHER, "traffic": None,
                         "bytes_per_launch": npts * (12 + 1536 + 16), "avg_launch_ms": t_d * 1e3,
                         "note": "algorithmic gather bytes (before reuse between neighbouring samples); the plane stack is "
                                 "L2 / Infinity-Cache resident, so the bound is the cache-served row-gather rate "
                                 "(MI355X_MICROARCH.md 'Indexed rows': 16.8 TB/s from L2, 8.6 TB/s from the Infinity Cache), "
                                 "not HBM (HBM traffic is ~60 MB per launch); the fused 32-64-4 decoder runs on the VALU"}}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs for a 1-GPU box (never set by the driver): BENCH_BACKEND=gloo BENCH_SHARE_GPU=1 lets the
    # ranks share cuda:0 and reduce through gloo, which exercises the same code path as RCCL
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    share = os.environ.get("BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if (world == 1 or share) else local_rank
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import nerf_siren_amd
    from nerf_siren_amd import Embedding, NeRF, SemanticNeRF, SirenField, render_rays, synth
    from nerf_siren_amd import ops
    from nerf_siren_amd.parallel import FlatGradAllReduce
    from nerf_siren_amd.training import FusedAdam, FusedMSELoss

    B = args.batch
    nerf_siren_amd.set_math(args.math)
    emb = [Embedding(3, 10), Embedding(3, 4)]
    # a pool of batches resident in HBM before the timed region (each rank its own shard of rays)
    n_pool = 8
    rays_pool = [torch.from_numpy(synth.blender_rays(B, seed=1000 * rank + i)).to(dev) for i in range(n_pool)]
    tgt_pool = [torch.from_numpy(synth.hash_uniform((B, 3), 77 + 1000 * rank + i)).to(dev) for i in range(n_pool)]

    def make_models(field):
        ms = []
        for seed in (1, 2):
            if field == "siren":
                sm = SemanticNeRF()
                sm.load_state_dict({k: torch.from_numpy(v) for k, v in synth.siren_params(seed).items()})
                m = SirenField(sm, torch.from_numpy(synth.hash_normal((1, 2304), 10 + seed)),
                               torch.from_numpy(synth.hash_normal((1, 2304), 20 + seed)))
            else:
                m = NeRF()
                m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_params(seed, structured=False).items()})
            ms.append(m.to(dev))
        return ms

    timer = KernelTimer(torch)
    ops.set_profile_hook(timer)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize; MAX over ranks."""
        for i in range(warmup):
            step(i)
        barrier()
        timer.clear()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        barrier()
        dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item())

    def make_step(models, mode):
        if mode == "infer":
            def step(i):
                with torch.no_grad():
                    render_rays(models, emb, rays_pool[i % n_pool], 64, False, 0, 0, 64, 1024 * 32, True, True)
            return step
        if args.optimizer == "fused":
            opt = FusedAdam(models, lr=5e-4, eps=1e-8)           # utils/__init__.py:20 (Adam, eps 1e-8), one launch/model
            loss_fn = FusedMSELoss(unit_grad=True)               # losses.py:10-20 + autograd, one launch
        else:
            opt = torch.optim.Adam([p for m in models for p in m.param_list()], lr=5e-4, eps=1e-8)
            loss_fn = None
        reducer = FlatGradAllReduce(models, world, overlap=not args.no_overlap)

        def step(i):
            res = render_rays(models, emb, rays_pool[i % n_pool], 64, False, 1.0, 1.0, 64, 1024 * 32, True, False)
            t = tgt_pool[i % n_pool]
            if loss_fn is not None:
                loss = loss_fn(res, t)
            else:
                loss = ((res["rgb_coarse"] - t) ** 2).mean() + ((res["rgb_fine"] - t) ** 2).mean()   # losses.py:15-20
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if loss_fn is not None:
                reducer.all_reduce(average=False)               # the 1/world factor rides in the Adam kernel
                opt.step(grad_scale=1.0 / world)
            else:
                reducer.all_reduce()
                opt.step()
        return step

    siren = args.field == "siren"
    train = args.mode == "train"
    models = make_models(args.field)
    step = make_step(models, args.mode)
    dt = timed(step, args.steps, args.warmup)

    if siren:
        kname = "siren_forward_rays_train" if train else ("siren_forward_rays_fast" if args.math == "bf16x3" else "siren_forward_rays")
    else:
        kname = "nerf_forward_rays_fast" if args.math == "bf16x3" else "nerf_forward_rays"
    kern_ms = timer.mean_ms(kname)

    # The same K steps once more on the OPT-IN split-bf16 math (fp32-level accuracy on the bf16 matrix cores, same parity
    # tests) -- reported beside the headline, never as `value`.
    opt_in = None
    if args.math == "fp32" and not siren and not args.no_opt_in:
        nerf_siren_amd.set_math("bf16x3")
        dt2 = timed(step, args.steps, args.warmup)
        nerf_siren_amd.set_math("fp32")
        opt_in = {"math": "bf16x3 (exact 3-way bf16 split of both operands, 6 bf16 MFMA per product, fp32 accumulate)",
                  "value": world * B * 192 * args.steps / dt2, "unit": "ray-samples/s", "ms_per_step": dt2 / args.steps * 1e3}

    # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc summary (tools/pmc_summary.py);
    # counters cannot be collected from inside the timed run
    def pmc_traffic(want):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            for k, e in pmc.items():
                if want in k and "hbm_bytes" in e and B == 1024:
                    return e["hbm_bytes"]
        except Exception:
            pass
        return None
    if siren:
        traffic = pmc_traffic("siren_forward_kernel<true, false, true," if train else "siren_forward_kernel<true, false, false,")
    else:
        traffic = pmc_traffic("nerf_forward_kernel<false, false, true>" if train else "nerf_forward_kernel<false, false, false>")
    flops_per_launch = B * 128 * (FLOP_SIREN if siren else FLOP_FULL)
    achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12
    # exact fp32 MFMA: 157.3 TF dense; split-bf16 (six bf16 MFMAs per fp32-equivalent product): 2500/6
    peak = PEAK_F32_MFMA if args.math == "fp32" else 2500.0 / 6.0

    extra = {}
    if world == 1 and not args.no_extra and args.math == "fp32":
        ks, kw = max(5, args.steps // 2), max(2, args.warmup // 2)
        if not siren:
            # the same field at test time (eval.py:85-96)
            st = make_step(models, "infer") if train else make_step(make_models("nerf"), "train")
            d2 = timed(st, ks, kw)
            k2 = timer.mean_ms("nerf_forward_rays")
            extra["infer" if train else "train"] = {
                "ms_per_step": d2 / ks * 1e3, "value": B * 192 * ks / d2, "unit": "ray-samples/s",
                "roofline": {"bound": "mfma", "kernel": "nerf_forward_kernel (fine MLP, 128 samples/ray)",
                             "achieved": B * 128 * FLOP_FULL / (k2 * 1e-3) / 1e12, "peak": PEAK_F32_MFMA, "unit": "TFLOP/s",
                             "frac": B * 128 * FLOP_FULL / (k2 * 1e-3) / 1e12 / PEAK_F32_MFMA, "avg_launch_ms": k2}}
            # the FiLM-SIREN field of configs[1] (models/nerf.py:142-216) behind the same render_rays: training + inference
            sm = make_models("siren")
            d_t = timed(make_step(sm, "train"), ks, kw)
            k_t = timer.mean_ms("siren_forward_rays_train")
            d_i = timed(make_step(sm, "infer"), ks, kw)
            k_i = timer.mean_ms("siren_forward_rays")
            nerf_siren_amd.set_math("bf16x3")                     # the opt-in split-bf16 inference kernel of the same field
            d_f = timed(make_step(sm, "infer"), ks, kw)
            nerf_siren_amd.set_math("fp32")
            fl = B * 128 * FLOP_SIREN
            extra["siren"] = {
                "workload": f"configs[1] with the FiLM-SIREN field (9 FiLM layers x 256, 529 156 parameters) coarse+fine, "
                            f"batch_size={B}",
                "train_ms_per_step": d_t / ks * 1e3, "train_value": B * 192 * ks / d_t,
                "infer_ms_per_step": d_i / ks * 1e3, "infer_value": B * 192 * ks / d_i, "unit": "ray-samples/s",
                "opt_in_infer": {"math": "bf16x3", "ms_per_step": d_f / ks * 1e3, "value": B * 192 * ks / d_f},
                "roofline": {"bound": "mfma", "kernel": "siren_forward_kernel<true,false,false> (fine pass, 128 samples/ray, "
                             "inference)", "achieved": fl / (k_i * 1e-3) / 1e12, "peak": PEAK_F32_MFMA, "unit": "TFLOP/s",
                             "frac": fl / (k_i * 1e-3) / 1e12 / PEAK_F32_MFMA, "flops_per_launch": fl, "avg_launch_ms": k_i,
                             "traffic": pmc_traffic("siren_forward_kernel<true, false, false,")},
                "roofline_train_forward": {"bound": "mfma", "kernel": "siren_forward_kernel<true,false,true> (fine pass, saves "
                                           "activations)", "achieved": fl / (k_t * 1e-3) / 1e12, "peak": PEAK_F32_MFMA,
                                           "unit": "TFLOP/s", "frac": fl / (k_t * 1e-3) / 1e12 / PEAK_F32_MFMA,
                                           "avg_launch_ms": k_t,
                                           "traffic": pmc_traffic("siren_forward_kernel<true, false, true,")}}
            del sm
        ops.set_profile_hook(None)
        extra["eg3d"] = eg3d_bench(dev, ks, kw)

    if rank == 0:
        total_samples = world * B * 192 * args.steps
        out = {
            "metric": "ray-samples/sec, Blender-lego 400x400 synthetic rays, 64c+64f"
                      + (" (training step)" if train else " (inference, test_time)"),
            "value": total_samples / dt, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.math == "fp32" else "f32 (3xbf16 split, 6 bf16 MFMA per product)", "data": "synthetic",
            "config": {"workload": f"configs[1]: Blender-lego 400x400 rays, N_samples=64 N_importance=64, "
                                   f"batch_size={B} rays/GPU, {'FiLM-SIREN 9x256' if siren else 'NeRF 8x256'} coarse+fine, "
                                   f"mode={args.mode}",
                       "rays_per_gpu": B, "samples_per_ray": 192, "mode": args.mode,
                       "parallelism": (f"dp{world}, one rank per GPU, {'fine-model all-reduce overlapped with the coarse backward' if not args.no_overlap else 'one joint all-reduce'}"
                                       if world > 1 else "single")},
            "rays_per_s": world * B * args.steps / dt,
            "roofline": {"bound": "mfma", "kernel": ("siren_forward_kernel" if siren else ("nerf_forward_kernel" if args.math == "fp32" else
                                                                          "nerf_forward_bf16x3_kernel"))
                                   + " (fine MLP, 128 samples/ray" + (", saves activations)" if train else ")"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "flops_per_launch": flops_per_launch, "avg_launch_ms": kern_ms},
        }
        out.update(extra)
        if opt_in is not None:
            out["opt_in"] = opt_in
        if world == 1 and train and not siren and not args.no_psnr:
            out["psnr"] = psnr_check(dev)
        if world == 1 and not args.no_cpu_baseline and not siren:
            out["cpu_baseline"] = cpu_baseline(args.mode, budget_s=args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
