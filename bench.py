#!/usr/bin/env python
"""Headline benchmark: ray-samples/s of render_rays() on Blender-lego-like rays
(400x400 pinhole, N_samples=64 + N_importance=64), BASELINE.json configs[1]
(batch 1024 rays per GPU), synthetic rays and seeded random-init weights.

One "step" = one pass of the hot path over one batch of 1024 rays per rank:
  --mode train (default): render_rays forward (coarse+fine, perturb=1,
      noise_std=1) + MSE(coarse)+MSE(fine) + backward through both MLPs +
      gradient all-reduce (RCCL, N>1) + Adam step  -- system.py:257-275
  --mode infer: render_rays(test_time=True, perturb=0, noise_std=0) under
      no_grad -- eval.py:85-96
A ray-sample = one field evaluation: 64 coarse + 128 fine = 192 per ray.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fine
NeRF MLP forward, fp32 MFMA bound): algorithmic FLOPs per launch / its average
duration measured with HIP events on the launch stream inside the timed region.
`cpu_baseline` is the numpy oracle timed on this box's host cores on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FULL = 1_186_816        # NeRF full forward per sample
FLOP_SIGMA = 982_528         # sigma-only forward per sample
FLOP_TRAIN = 3_489_024       # fwd + dW + dX per sample
FLOP_SIREN = 1_053_696       # FiLM-SIREN full forward per sample (+2 304 sin)
PEAK_F32_MFMA = 157.3        # TFLOP/s dense (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["train", "infer"], default="train")
    ap.add_argument("--batch", type=int, default=1024, help="rays per rank per step (configs[1])")
    ap.add_argument("--field", choices=["nerf", "siren"], default="nerf",
                    help="nerf = the reference's live 8x256 ReLU NeRF; siren = its FiLM-SIREN field (inference only)")
    ap.add_argument("--math", choices=["fp32", "bf16x3"], default="fp32",
                    help="fp32 = exact fp32 MFMA (default); bf16x3 = opt-in split-bf16 inference math (--mode infer)")
    ap.add_argument("--optimizer", choices=["fused", "torch"], default="fused",
                    help="training: nerf_siren_amd.training.FusedAdam + FusedMSELoss (one launch each) or torch.optim.Adam "
                         "+ elementwise loss -- same arithmetic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true", help="skip the PSNR-parity check (tests/golden/g15_psnr.npz protocol)")
    ap.add_argument("--no-opt-in", action="store_true", help="skip the extra timed loop on the opt-in split-bf16 math")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work in the bounded cpu_baseline sample")
    return ap.parse_args()


def cpu_baseline(mode, budget_s=12.0, chunk=256, max_rays=8192):
    """numpy oracle (kind 'port') on the host cores: chunks of `chunk` rays of the same workload until
    ~budget_s seconds of CPU work are done (bounded sample; the chunking only bounds memory)."""
    from nerf_siren_amd import synth
    from oracle import nerf_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    params = [synth.nerf_params(1, False), synth.nerf_params(2, False)]
    done, dt, i = 0, 0.0, 0
    while dt < budget_s and done < max_rays:
        n = chunk
        rays = synth.blender_rays(n, seed=123 + i)
        t0 = time.perf_counter()
        if mode == "infer":
            O.render_rays(params, rays, 64, False, 0.0, 0.0, 64, True, True)
        else:
            rng = {"perturb_rand": synth.hash_uniform((n, 64), 1), "noise_coarse": synth.hash_normal((n, 64), 2),
                   "u": synth.hash_uniform((n, 64), 3), "noise_fine": synth.hash_normal((n, 128), 4)}
            res = O.render_rays(params, rays, 64, False, 1.0, 1.0, 64, True, False, rng=rng, keep=True)
            tgt = synth.hash_uniform((n, 3), 5)
            g = {"rgb_coarse": 2 * (res["rgb_coarse"] - tgt) / (3 * n), "rgb_fine": 2 * (res["rgb_fine"] - tgt) / (3 * n)}
            O.render_rays_backward(params, res, g, True)
        dt += time.perf_counter() - t0
        done += n
        i += 1
    return {"value": done * 192 / dt, "unit": "ray-samples/s", "cores": int(cores), "kind": "port",
            "sample": f"{done} rays x (64+128) samples in chunks of {chunk}, {mode} step "
                      f"(fwd{'+bwd' if mode == 'train' else ''}), numpy oracle, {dt:.1f} s"}


def psnr_check(dev):
    """The metric's '+ PSNR': the reference trained 240 Adam steps on a teacher scene on CPU and its validation-PSNR
    trajectory is a committed fixture (tools/make_psnr_golden.py -> tests/golden/g15_psnr.npz); the same steps (same
    images, batches, injected draws, initial weights) run here on the HIP path.  A few seconds; rank 0 at N = 1 only."""
    import numpy as np
    import torch
    from nerf_siren_amd import Embedding, NeRF, render_rays, synth
    from nerf_siren_amd.training import FusedAdam, FusedMSELoss
    path = os.path.join(ROOT, "tests", "golden", "g15_psnr.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    S, F, B = int(g["cfg_S"]), int(g["cfg_F"]), int(g["cfg_batch"])
    steps, every = int(g["cfg_steps"]), int(g["cfg_eval_every"])
    ms = []
    for seed in (11, 12):
        m = NeRF()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_params(seed, structured=False).items()})
        ms.append(m.to(dev))
    emb = [Embedding(3, 10), Embedding(3, 4)]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)          # noqa: E731
    rays, tgt, val_rays, val_tgt = T(g["rays"]), T(g["target"]), T(g["val_rays"]), T(g["val_target"])
    opt, loss_fn = FusedAdam(ms, lr=float(g["cfg_lr"]), eps=1e-8), FusedMSELoss(unit_grad=True)
    psnr = []
    for step in range(steps + 1):
        if step % every == 0:
            with torch.no_grad():
                r = render_rays(ms, emb, val_rays, S, False, 0, 0, F, 1 << 15, True, False)
            psnr.append(float(-10 * torch.log10(((r["rgb_fine"] - val_tgt) ** 2).mean())))
        if step == steps:
            break
        idx = torch.from_numpy(synth.psnr_batch_indices(step, rays.shape[0], B)).to(dev)
        rg = {k: T(v) for k, v in synth.psnr_step_rng(step, B, S, F).items()}
        res = render_rays(ms, emb, rays[idx], S, False, 1.0, 0.0, F, 1 << 15, True, False, rng=rg)
        loss = loss_fn(res, tgt[idx])
        opt.zero_grad()
        loss.backward()
        opt.step()
    ref = [float(v) for v in g["psnr"]]
    return {"value_db": psnr[-1], "reference_db": ref[-1], "max_abs_diff_db": float(np.abs(np.array(psnr) - ref).max()),
            "trajectory_db": [round(v, 3) for v in psnr], "reference_trajectory_db": [round(v, 3) for v in ref],
            "protocol": f"teacher scene, {steps} Adam steps of {B} rays ({S}+{F}), validation every {every} steps; "
                        "reference = /root/reference on CPU (tests/golden/g15_psnr.npz)"}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs for a 1-GPU box (never set by the driver): BENCH_BACKEND=gloo BENCH_SHARE_GPU=1 lets two
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

    from nerf_siren_amd import Embedding, NeRF, render_rays, synth
    from nerf_siren_amd import ops
    from nerf_siren_amd.parallel import FlatGradAllReduce

    B = args.batch
    if args.math != "fp32":
        if args.field == "siren" and args.mode != "infer":
            raise SystemExit("--field siren supports --mode infer only")
        import nerf_siren_amd
        nerf_siren_amd.set_math(args.math)
    models = []
    siren = args.field == "siren"
    if siren and args.mode != "infer":
        raise SystemExit("--field siren supports --mode infer only (the reference never trains its SIREN field)")
    for seed in (1, 2):
        if siren:
            from nerf_siren_amd import SemanticNeRF, SirenField
            sm = SemanticNeRF()
            sm.load_state_dict({k: torch.from_numpy(v) for k, v in synth.siren_params(seed).items()})
            m = SirenField(sm, torch.from_numpy(synth.hash_normal((1, 2304), 10 + seed)),
                           torch.from_numpy(synth.hash_normal((1, 2304), 20 + seed)))
        else:
            m = NeRF()
            m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_params(seed, structured=False).items()})
        models.append(m.to(dev))
    emb = [Embedding(3, 10), Embedding(3, 4)]
    # a pool of batches resident in HBM before the timed region (each rank its own shard of rays)
    n_pool = 8
    rays_pool = [torch.from_numpy(synth.blender_rays(B, seed=1000 * rank + i)).to(dev) for i in range(n_pool)]
    tgt_pool = [torch.from_numpy(synth.hash_uniform((B, 3), 77 + 1000 * rank + i)).to(dev) for i in range(n_pool)]

    train = args.mode == "train"
    if train:
        params = [p for m in models for p in m.parameters()]
        from nerf_siren_amd.training import FusedAdam, FusedMSELoss
        if args.optimizer == "fused":
            opt = FusedAdam(models, lr=5e-4, eps=1e-8)           # utils/__init__.py:20 (Adam, eps 1e-8), one launch/model
            loss_fn = FusedMSELoss(unit_grad=True)               # losses.py:10-20 + autograd, one launch
        else:
            opt = torch.optim.Adam(params, lr=5e-4, eps=1e-8)
            loss_fn = None
        reducer = FlatGradAllReduce(models, world)

    # ---- per-kernel timing of the dominant kernel (fine MLP forward) -------------------------
    ev = []
    orig_fwd = ops.nerf_forward_rays

    def timed_fwd(packed, rays, z, sigma_only=False, save=False):
        if z.shape[1] != 128:
            return orig_fwd(packed, rays, z, sigma_only, save)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = orig_fwd(packed, rays, z, sigma_only, save)
        b.record()
        ev.append((a, b))
        return out

    import nerf_siren_amd.rendering as R
    R.ops.nerf_forward_rays = timed_fwd
    if args.math == "bf16x3":
        orig_fast = ops.nerf_forward_rays_fast

        def timed_fast(packed, fast, rays, z, sigma_only=False, save=False):
            if z.shape[1] != 128:
                return orig_fast(packed, fast, rays, z, sigma_only, save)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = orig_fast(packed, fast, rays, z, sigma_only, save)
            b.record()
            ev.append((a, b))
            return out

        ops.nerf_forward_rays_fast = timed_fast
    if siren:
        orig_siren = ops.siren_forward_rays

        def timed_siren(packed, rays, z, freq, phase, rpc, sigma_only=False):
            if z.shape[1] != 128:
                return orig_siren(packed, rays, z, freq, phase, rpc, sigma_only)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = orig_siren(packed, rays, z, freq, phase, rpc, sigma_only)
            b.record()
            ev.append((a, b))
            return out

        ops.siren_forward_rays = timed_siren
        orig_siren_fast = ops.siren_forward_rays_fast

        def timed_siren_fast(packed, fast, rays, z, freq, phase, rpc, sigma_only=False):
            if z.shape[1] != 128:
                return orig_siren_fast(packed, fast, rays, z, freq, phase, rpc, sigma_only)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = orig_siren_fast(packed, fast, rays, z, freq, phase, rpc, sigma_only)
            b.record()
            ev.append((a, b))
            return out

        ops.siren_forward_rays_fast = timed_siren_fast

    def step(i):
        rays = rays_pool[i % n_pool]
        if train:
            res = render_rays(models, emb, rays, 64, False, 1.0, 1.0, 64, 1024 * 32, True, False)
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
        else:
            with torch.no_grad():
                render_rays(models, emb, rays, 64, False, 0, 0, 64, 1024 * 32, True, True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    ev.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")

    # The same K steps once more on the OPT-IN split-bf16 math (fp32-level accuracy on the bf16 matrix cores, same parity
    # tests) -- reported beside the headline, never as `value`.
    opt_in = None
    if args.math == "fp32" and not siren and not args.no_opt_in:
        import nerf_siren_amd
        nerf_siren_amd.set_math("bf16x3")
        for i in range(args.warmup):
            step(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        barrier()
        dt2 = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dt2, op=dist.ReduceOp.MAX)
        nerf_siren_amd.set_math("fp32")
        dt2 = float(dt2.item())
        opt_in = {"math": "bf16x3 (exact 3-way bf16 split of both operands, 6 bf16 MFMA per product, fp32 accumulate)",
                  "value": world * B * 192 * args.steps / dt2, "unit": "ray-samples/s", "ms_per_step": dt2 / args.steps * 1e3}
    # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc summary (tools/pmc_summary.py);
    # counters cannot be collected from inside the timed run
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        want = ("siren_forward_kernel<true, false>" if siren else
                ("nerf_forward_kernel<false, false, true>" if train else "nerf_forward_kernel<false, false, false>"))
        for k, e in pmc.items():
            if want in k and "hbm_bytes" in e and B == 1024:
                traffic = e["hbm_bytes"]
    except Exception:
        pass
    flops_per_launch = B * 128 * (FLOP_SIREN if siren else FLOP_FULL)
    achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12
    # exact fp32 MFMA: 157.3 TF dense; split-bf16 (six bf16 MFMAs per fp32-equivalent product): 2500/6
    peak = PEAK_F32_MFMA if args.math == "fp32" else 2500.0 / 6.0

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
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "rays_per_s": world * B * args.steps / dt,
            "roofline": {"bound": "mfma", "kernel": ("siren_forward_kernel" if siren else ("nerf_forward_kernel" if args.math == "fp32" else
                                                                          "nerf_forward_bf16x3_kernel"))
                                   + " (fine MLP, 128 samples/ray)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "flops_per_launch": flops_per_launch, "avg_launch_ms": kern_ms},
        }
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
