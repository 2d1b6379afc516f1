#!/usr/bin/env python
"""Headline benchmark: ray-samples/s of render_rays() on Blender-lego-like rays (400x400 pinhole, N_samples=64 +
N_importance=64), BASELINE.json configs[1] -- "SIREN-MLP, batch_size=1024, 1xMI355X" -- synthetic rays and seeded
random-init weights.  The headline field is the reference's FiLM-SIREN field (models/nerf.py:142-216, 9 FiLM layers x 256)
behind render_rays; the reference's live ReLU NeRF 8x256 (system.py:181-192) runs beside it as the `nerf` object.

One "step" = one pass of the hot path over one batch of rays per rank:
  --mode train (default): render_rays forward (coarse+fine, perturb=1, noise_std=1) + MSE(coarse)+MSE(fine) + backward
      through both fields + gradient all-reduce (RCCL, N>1; the fine model's slice goes on the wire while the coarse
      model's backward runs) + Adam step  -- system.py:257-275
  --mode infer: render_rays(test_time=True, perturb=0, noise_std=0) under no_grad -- eval.py:85-96
A ray-sample = one field evaluation: 64 coarse + 128 fine = 192 per ray.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (a child
`python -m torch.distributed.run`, before this process makes any GPU call) and relays rank 0's JSON line; under
torch.distributed.run it is one rank.  --scaling weak (default): --batch rays per rank; --scaling strong: --global-batch
rays (default 8192, configs[2]) split over the ranks.

Prints ONE JSON line (rank 0).  `roofline`: all three training kernels of the field (forward-with-save, dX chain, dW GEMM;
inference: the two forward kernels) from HIP events recorded by the library on the launch stream INSIDE the timed region
(nerfmi_profile_start/report, csrc/render.hip): algorithmic FLOPs of the launches / their measured time, against the dense
fp32 MFMA peak; `roofline.kernel` names the kernel with the largest share of the step and `roofline.frac` is ITS fraction
(what `rocprofv3 --kernel-trace --stats` gives as FLOPs x points / (launches x AverageNs)); `roofline.step` is the
step-level fraction FLOP/step / ms_per_step / peak.  Beside the headline, at N = 1: `infer` (the same field at test time),
`nerf` (the reference's live ReLU NeRF: training + inference, same roofline breakdown), `eg3d` (configs[4]), `opt_in`
(split-bf16 math), `psnr` (committed teacher-scene protocols) and `cpu_baseline` = the reference's CPU PyTorch path on the
SAME workload restated op for op (oracle/torch_cpu_ref.py) on this box's cores, on a bounded sample.  At N > 1: `dist`
(what torch.distributed / RCCL saw) and `comm` (the all-reduce alone and its exposed part under overlap).
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

# algorithmic FLOPs per sample (2 x multiply-adds of the dense layers)
FLOP = {
    "nerf": {"fwd": 1_186_816, "fwd_sigma": 982_528, "chain": 1_115_392, "dw": 1_186_816},     # train 3 489 024
    "siren": {"fwd": 1_053_696, "fwd_sigma": 919_552, "chain": 1_050_624, "dw": 1_053_696},    # train 3 158 016 (+2 304 sin)
}
PEAK_F32_MFMA = 157.3        # TFLOP/s dense (MI355X_MICROARCH.md)
PEAK_HBM = 8.0               # TB/s
PEAK_L2_GATHER = 16.8        # TB/s: rows gathered from an L2-resident table (MI355X_MICROARCH.md "Indexed rows")
PEAK_MALL_GATHER = 8.6       # TB/s: the same from the Infinity Cache

# kernel tags of the library's profiler (csrc/*.hip KernelSpan) -> (FLOP key, rocprof kernel-name fragment for the PMC file)
KERNELS = {
    "siren": {"train": [("siren_forward_kernel<save>", "fwd", "siren_forward_kernel<true, false, true,"),
                        ("siren_backward_chain_kernel", "chain", "siren_backward_chain_kernel"),
                        ("siren_dw_kernel", "dw", "siren_dw_kernel")],
              "infer": [("siren_forward_kernel", "fwd", "siren_forward_kernel<true, false, false,"),
                        ("siren_forward_kernel<sigma_only>", "fwd_sigma", "siren_forward_kernel<true, true, false,")]},
    "nerf": {"train": [("nerf_forward_kernel<save>", "fwd", "nerf_forward_kernel<false, false, true>"),
                       ("nerf_backward_chain_kernel", "chain", "nerf_backward_chain_kernel"),
                       ("nerf_dw_kernel", "dw", "nerf_dw_kernel")],
             "infer": [("nerf_forward_kernel", "fwd", "nerf_forward_kernel<false, false, false>"),
                       ("nerf_forward_kernel<sigma_only>", "fwd_sigma", "nerf_forward_kernel<false, true, false>")]},
}
KERNELS_BF16X3 = {
    "nerf": {"train": [("nerf_forward_bf16x3_kernel<save>", "fwd", None), ("nerf_backward_chain_bf16x3_kernel", "chain", None),
                       ("nerf_dw_bf16x3_kernel", "dw", None)],
             "infer": [("nerf_forward_bf16x3_kernel", "fwd", None), ("nerf_forward_bf16x3_kernel<sigma_only>", "fwd_sigma", None)]},
    "siren": {"train": [("siren_forward_bf16x3_kernel<save>", "fwd", None), ("siren_backward_chain_bf16x3_kernel", "chain", None),
                        ("siren_dw_bf16x3_kernel", "dw", None)],
              "infer": [("siren_forward_bf16x3_kernel", "fwd", None), ("siren_forward_bf16x3_kernel<sigma_only>", "fwd_sigma", None)]},
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["train", "infer"], default="train")
    ap.add_argument("--batch", type=int, default=1024, help="rays per rank per step (configs[1]); --scaling weak")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch rays on every rank; strong: --global-batch rays split over the ranks")
    ap.add_argument("--global-batch", type=int, default=8192, help="--scaling strong: rays per step over all ranks (configs[2])")
    ap.add_argument("--field", choices=["siren", "nerf"], default="siren",
                    help="headline field: siren = the FiLM-SIREN field BASELINE.json configs[1] names; nerf = the reference's "
                         "live 8x256 ReLU NeRF (system.py:183-190; always reported in the `nerf` object at N = 1)")
    ap.add_argument("--math", choices=["fp32", "bf16x3"], default="fp32",
                    help="fp32 = exact fp32 MFMA (default); bf16x3 = opt-in split-bf16 math")
    ap.add_argument("--optimizer", choices=["fused", "torch"], default="fused",
                    help="training: nerf_siren_amd.training.FusedAdam + FusedMSELoss (one launch each) or torch.optim.Adam "
                         "+ elementwise loss -- same arithmetic")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one joint all-reduce after the whole backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true", help="skip the PSNR-parity protocols (tests/golden/g15s, g19)")
    ap.add_argument("--no-opt-in", action="store_true", help="skip the extra timed loops on the opt-in split-bf16 math")
    ap.add_argument("--no-extra", action="store_true", help="skip the infer / nerf / eg3d objects")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not record per-kernel HIP events in the timed region (A/B of their cost; `roofline` is then null)")
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
# CPU baseline: the reference's torch-CPU path on the same workload (restated), calibrated thread count
# -------------------------------------------------------------------------------------------------
def cpu_baseline(field, mode, budget_s=12.0):
    from nerf_siren_amd import synth
    from oracle import torch_cpu_ref as TR
    if field == "siren":
        ps = [dict(synth.siren_params(s), frequencies=synth.hash_normal((1, 2304), 10 + s),
                   phase_shifts=synth.hash_normal((1, 2304), 20 + s)) for s in (1, 2)]
        what = ("FiLM-SIREN fields (models/nerf.py:142-216) behind the reference's render_rays through the "
                "forward(x, sigma_only) adapter of tools/make_psnr_golden.py --siren")
    else:
        ps = [synth.nerf_params(1, False), synth.nerf_params(2, False)]
        what = "NeRF 8x256 fields"
    r = TR.timed_sample(mode, ps, lambda i: synth.blender_rays(1024, seed=123 + i),
                        lambda i: synth.hash_uniform((1024, 3), 5 + i), budget_s=budget_s, n_rays=1024)
    return {"value": r["ray_samples_per_s"], "unit": "ray-samples/s", "cores": r["threads"], "kind": "port",
            "sample": f"{r['steps']} steps of 1024 rays x (64+128) samples, {mode} step "
                      f"({'fwd+MSE+bwd+Adam' if mode == 'train' else 'test_time render'}), {what}: the reference's torch-CPU op "
                      f"sequence restated (oracle/torch_cpu_ref.py, pinned on fixtures the imported reference produced), "
                      f"{r['seconds']:.1f} s after one warm-up step",
            "best_step_value": r["ray_samples_per_s_best"], "cpu_model": r["cpu_model"],
            "physical_cores": r["physical_cores"], "nproc": r["nproc"], "allowed_cpus": r["allowed_cpus"],
            "cgroup_cpu_quota": r["cgroup_cpu_quota"], "thread_calibration_gemm_gflops": r["gemm_gflops"],
            "torch_parallel_info": r["parallel_info"].strip().replace("\n", "; "),
            "anchor": "the imported reference itself (NeRF field) on 8 threads in the build container: 0.107-0.119 M "
                      "ray-samples/s training, 0.56-0.69 M inference (BASELINE.md section 2, oracle/torch_cpu_ref.py header)"}


def psnr_check(dev, field):
    """The metric's '+ PSNR': the reference trained on a teacher scene on CPU and its validation-PSNR trajectory is a
    committed fixture (tools/make_psnr_golden.py); the same steps (same images, batches, injected draws, initial weights,
    learning-rate schedule) run here on the HIP path.  field = 'siren': g15s (the reference's SemanticNeRF as the student,
    240 steps); 'nerf': g19 (1 500 steps of 1024 rays, 400x400 validation view).  Rank 0 at N = 1 only."""
    import numpy as np
    import torch
    from nerf_siren_amd import Embedding, NeRF, SemanticNeRF, SirenField, render_rays, synth
    from nerf_siren_amd.training import FusedAdam, FusedMSELoss
    name = "g15s_psnr_siren.npz" if field == "siren" else "g19_psnr_long.npz"
    path = os.path.join(ROOT, "tests", "golden", name)
    if not os.path.exists(path):
        return None
    g = dict(np.load(path))
    S, F, B = int(g["cfg_S"]), int(g["cfg_F"]), int(g["cfg_batch"])
    steps, every = int(g["cfg_steps"]), int(g["cfg_eval_every"])
    milestones = [int(v) for v in g["cfg_lr_milestones"]] if "cfg_lr_milestones" in g else []
    gamma = float(g["cfg_lr_gamma"]) if "cfg_lr_gamma" in g else 1.0
    ms = []
    for seed in (11, 12):
        if field == "siren":
            sm = SemanticNeRF()
            sm.load_state_dict({k: torch.from_numpy(v) for k, v in synth.siren_params(seed).items()})
            m = SirenField(sm, torch.from_numpy(synth.hash_normal((1, 2304), 10 + seed)),
                           torch.from_numpy(synth.hash_normal((1, 2304), 20 + seed)))
        else:
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
                    mse += float(((r["rgb_fine"] - val_tgt[i:i + (1 << 15)]).double() ** 2).sum())
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
    return {"field": field, "value_db": psnr[-1], "reference_db": ref[-1],
            "max_abs_diff_db": float(np.abs(np.array(psnr) - ref).max()),
            "trajectory_db": [round(v, 3) for v in psnr], "reference_trajectory_db": [round(v, 3) for v in ref],
            "protocol": f"teacher scene, {steps} Adam steps of {B} rays ({S}+{F})"
                        + (f", lr x{gamma} at steps {milestones}" if milestones else "")
                        + f", validation every {every} steps on {val_rays.shape[0]} rays; reference = /root/reference on CPU "
                          f"(tests/golden/{name})"}


def pmc_entry(fragment, B):
    """Per-launch counters of a kernel from the committed rocprofv3 --pmc summary (tools/pmc_summary.py; the fine-pass
    launch of the 1024-ray batch) -- counters cannot be collected from inside the timed run."""
    if fragment is None or B != 1024:
        return None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    except Exception:
        return None
    for k, e in pmc.items():
        if fragment in k:
            return e
    return None


def kernel_rooflines(report, field, mode, B, table=KERNELS, peak=PEAK_F32_MFMA):
    """report: ops.profile_report() of a timed region -> per-kernel rooflines + the dominant kernel.
    Per kernel: `achieved` = algorithmic FLOPs of ALL its launches in the region / their summed HIP-event time (coarse and
    fine launches together: FLOP/sample x points per coarse+fine pair / (2 x rocprof's AverageNs)); `fine_launch` = the
    131 072-point launch alone."""
    out, total_ms = {}, 0.0
    for tag, fkey, frag in table[field][mode]:
        spans = {u: v for (t, u), v in report.items() if t == tag}
        if not spans:
            continue
        n = sum(c for c, _ in spans.values())
        ms = sum(m for _, m in spans.values())
        pts = sum(u * c for u, (c, _) in spans.items())
        fl = FLOP[field][fkey]
        e = {"launches": n, "avg_launch_ms": ms / n, "points_per_launch_avg": pts / n, "flop_per_point": fl,
             "achieved": fl * pts / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s", "total_ms": ms}
        e["frac"] = e["achieved"] / peak
        big = max(spans)
        c, m = spans[big]
        e["fine_launch"] = {"points": big, "avg_launch_ms": m / c, "achieved": fl * big / (m / c * 1e-3) / 1e12,
                            "frac": fl * big / (m / c * 1e-3) / 1e12 / peak, "flops_per_launch": fl * big}
        pm = pmc_entry(frag, B)
        if pm is not None:
            e["fine_launch"]["traffic"] = pm.get("hbm_bytes")
            e["fine_launch"]["mfma_busy_frac_pmc"] = pm.get("mfma_busy_frac")
        out[tag] = e
        total_ms += ms
    for e in out.values():
        e["share_of_mlp_time"] = e["total_ms"] / total_ms
    dom = max(out, key=lambda k: out[k]["total_ms"]) if out else None
    return out, dom


def roofline_object(report, field, mode, B, steps, dt, table=KERNELS, peak=PEAK_F32_MFMA, world=1):
    ks, dom = kernel_rooflines(report, field, mode, B, table, peak)
    if dom is None:
        return None
    f = FLOP[field]
    flop_step = (B * 192 * (f["fwd"] + f["chain"] + f["dw"])) if mode == "train" else B * (64 * f["fwd_sigma"] + 128 * f["fwd"])
    d = ks[dom]
    step_tf = flop_step / (dt / steps) / 1e12
    return {"bound": "mfma", "kernel": dom + " (coarse 64 + fine 128 samples/ray launches; the MLP kernel with the largest "
                                             "share of the step)",
            "achieved": d["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": d["frac"],
            "traffic": d["fine_launch"].get("traffic"), "traffic_note": "HBM bytes of the fine-pass launch, committed rocprofv3 "
            "--pmc pass (profiles/pmc_latest.json)", "avg_launch_ms": d["avg_launch_ms"],
            "flops_per_launch": d["flop_per_point"] * d["points_per_launch_avg"],
            "measured": "HIP events recorded by the library on the launch stream around every field-MLP launch (nerfmi_profile_*) "
                        "in a second timed region of the same K steps right behind the headline region (the events cost ~1 % of "
                        "a step, which the headline does not pay)",
            "kernels": ks,
            "step": {"flop_per_step_per_gpu": flop_step, "achieved": step_tf, "frac": step_tf / peak,
                     "note": "algorithmic FLOPs of one step / ms_per_step / peak (per GPU): everything that is not the three "
                             "MLP kernels -- per-ray kernels, loss, Adam, packing, launch gaps, collectives -- counts as loss"}}


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
    alg = npts * (12 + 1536 + 16)                            # R 12 B coords + 1 536 B texels / W 16 B per sample
    # where the texel requests were actually served: the committed rocprofv3 --pmc pass of tools/bench_eg3d.py
    # (profiles/r03_pmc_eg3d.json, tools/pmc_eg3d.sh): TCP->TCC requests, TCC hit rate, TCC_EA read requests, FETCH_SIZE
    served = None
    try:
        served = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_eg3d.json"))).get("dense_query")
    except Exception:
        pass
    roof = {"bound": "l2-gather", "kernel": "triplane_kernel<1,false> (dense run_model: 12 bilinear taps x 32 ch gathered per point "
                                            "from the 25 MB channels-last plane stack, decoder fused)",
            "achieved": alg / t_d / 1e12, "peak": PEAK_L2_GATHER, "unit": "TB/s", "frac": alg / t_d / 1e12 / PEAK_L2_GATHER,
            "traffic": None, "bytes_per_launch": alg, "avg_launch_ms": t_d * 1e3,
            "note": "`achieved` = ALGORITHMIC gather bytes (before reuse between neighbouring samples) / time against the guide's "
                    "L2-served row-gather rate (MI355X_MICROARCH.md 'Indexed rows': 16.8 TB/s from L2, 8.6 TB/s from the Infinity "
                    "Cache); it is a reuse-inclusive figure, not a bandwidth.  `served` (when the PMC pass is committed) says where "
                    "the requests were actually served and what the bytes-on-the-wire fractions are"}
    if served:
        # counters of profiles/r03_pmc_eg3d.json (per launch): 98 % of the vector-L1 line lookups hit in L1 (neighbouring points
        # of the grid share texels), the L2 sees 1.7 M read requests (hit rate 0.78) and the fabric 66 MB -- neither L2 nor the
        # Infinity Cache / HBM is anywhere near its bandwidth; the kernel is bound by the L1 / texture-addresser line rate
        acc = served.get("TCP_TOTAL_CACHE_ACCESSES_sum")
        l1_to_l2 = served.get("TCP_TCC_READ_REQ_sum")
        roof["traffic"] = served.get("hbm_bytes")
        roof["served"] = {"l1_line_lookups_per_launch": acc, "l1_hit_rate": (1 - l1_to_l2 / acc) if acc and l1_to_l2 else None,
                          "l2_read_requests_per_launch": l1_to_l2, "l2_hit_rate": served.get("l2_hit_rate"),
                          "fabric_read_bytes": served.get("fabric_read_bytes"), "fabric_write_bytes": served.get("fabric_write_bytes"),
                          "source": "profiles/r03_pmc_eg3d.json (rocprofv3 --pmc, tools/pmc_eg3d.sh)"}
        if acc:
            peak_lookups = 256 * 2.4e9            # one cache-line lookup per clock per CU at 2.4 GHz
            roof.update({"bound": "l1-line-rate", "achieved": acc / t_d / 1e9, "peak": peak_lookups / 1e9, "unit": "G line-lookups/s",
                         "frac": acc / t_d / peak_lookups,
                         "algorithmic_gather_TBps": alg / t_d / 1e12,
                         "note": "`achieved` = vector-L1 cache-line lookups per launch (TCP_TOTAL_CACHE_ACCESSES, committed PMC "
                                 "pass) / the launch time measured here, against one lookup per clock per CU (256 CUs x 2.4 GHz); "
                                 "`traffic` = fabric (Infinity Cache + HBM) bytes per launch.  Rounds 1-2 divided the ALGORITHMIC "
                                 "gather bytes (1 564 B/point before reuse: `algorithmic_gather_TBps`) by the guide's L2 row-gather "
                                 "rate; the counters show that figure was reuse, not bandwidth: 98 % of the lookups are L1 hits"})
    return {"workload": "configs[4]: planes (1,3,32,256,256), M=4096 rays x (64+64) samples; dense query 128^3 points",
            "forward_ms": t_f * 1e3, "forward_samples_per_s": M * 128 / t_f,
            "forward_backward_ms": t_fb * 1e3, "forward_backward_samples_per_s": M * 128 / t_fb,
            "dense_query_ms": t_d * 1e3, "dense_points_per_s": npts / t_d, "roofline": roof}


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

    if args.scaling == "strong":
        if args.global_batch % world:
            raise SystemExit(f"--scaling strong: --global-batch {args.global_batch} is not divisible by {world} ranks")
        B = args.global_batch // world
    else:
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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup, events=True):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize; MAX over ranks.  -> (seconds, kernel report)"""
        for i in range(warmup):
            step(i)
        barrier()
        comm_events.clear()
        if events and not args.no_kernel_events:
            ops.profile_start()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        barrier()
        dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        ops.profile_stop()
        rep = ops.profile_report() if (events and not args.no_kernel_events) else {}
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item()), rep

    comm_events = []         # (before, after) HIP events around reducer.all_reduce() on the compute stream, N > 1 only

    def make_step(models, mode, overlap=None):
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
        reducer = FlatGradAllReduce(models, world, overlap=(not args.no_overlap) if overlap is None else overlap)

        def step(i):
            res = render_rays(models, emb, rays_pool[i % n_pool], 64, False, 1.0, 1.0, 64, 1024 * 32, True, False)
            t = tgt_pool[i % n_pool]
            if loss_fn is not None:
                loss = loss_fn(res, t)
            else:
                loss = ((res["rgb_coarse"] - t) ** 2).mean() + ((res["rgb_fine"] - t) ** 2).mean()   # losses.py:15-20
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if world > 1:
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record()
            if loss_fn is not None:
                reducer.all_reduce(average=False)               # the 1/world factor rides in the Adam kernel
            else:
                reducer.all_reduce()
            if world > 1:
                eb.record()
                comm_events.append((ea, eb))
            if loss_fn is not None:
                opt.step(grad_scale=1.0 / world)
            else:
                opt.step()
        return step

    field, mode = args.field, args.mode
    train = mode == "train"
    table = KERNELS if args.math == "fp32" else KERNELS_BF16X3
    peak = PEAK_F32_MFMA if args.math == "fp32" else 2500.0 / 6.0    # split-bf16: six bf16 MFMAs per fp32-equivalent product
    models = make_models(field)
    step = make_step(models, mode)
    # The headline: W warm-up + EXACTLY K timed steps with NO instrumentation.  The per-kernel HIP events cost ~1 % of a step
    # (32 event records per training step; same-box A/B in round 3: 5.314 / 5.319 ms without, 5.367 / 5.373 ms with), so they
    # are recorded in a SECOND timed region of the same K steps that follows immediately; `roofline` comes from that one
    # (its own ms/step is reported as roofline.ms_per_step_with_events), `roofline.step` from the headline's ms_per_step.
    dt, _ = timed(step, args.steps, args.warmup, events=False)
    rep, dt_ev = {}, None
    if not args.no_kernel_events:
        dt_ev, rep = timed(step, args.steps, 1)
    roof = roofline_object(rep, field, mode, B, args.steps, dt, table, peak) if rep else None
    if roof is not None:
        roof["ms_per_step_with_events"] = dt_ev / args.steps * 1e3

    def comm_ms():
        torch.cuda.synchronize()
        v = [a.elapsed_time(b) for a, b in comm_events]
        comm_events.clear()
        return sum(v) / len(v) if v else None

    # N > 1: what the ranks saw, and the collective alone vs its exposed part under overlap (a side run with the other setting)
    dist_info = comm = None
    if world > 1 and train:
        ms_main = comm_ms()
        other = bool(args.no_overlap)                            # the main run's overlap flag was `not args.no_overlap`
        ks = max(5, args.steps // 2)
        d_side, _ = timed(make_step(models, mode, overlap=other), ks, 2, events=False)
        ms_side = comm_ms()
        over_ms, plain_ms = (ms_side, ms_main) if other else (ms_main, ms_side)
        over_step, plain_step = (d_side / ks * 1e3, dt / args.steps * 1e3) if other else (dt / args.steps * 1e3, d_side / ks * 1e3)
        nbytes = sum(m.grad_numel for m in models) * 4
        comm = {"allreduce_bytes_per_step": nbytes,
                "allreduce_ms_no_overlap": plain_ms, "exposed_ms_with_overlap": over_ms,
                "ms_per_step_no_overlap": plain_step, "ms_per_step_with_overlap": over_step,
                "how": "HIP events on the compute stream around FlatGradAllReduce.all_reduce(): without overlap that is the one "
                       "joint all-reduce of both fields' gradients; with overlap the fine field's slice was put on the wire "
                       "under the coarse field's backward and what remains is the wait for it + the coarse slice's all-reduce"}
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device": torch.cuda.get_device_name(dev),
                "pci_bus_id": getattr(props, "pci_bus_id", None), "pci_domain_id": getattr(props, "pci_domain_id", None),
                "uuid": str(getattr(props, "uuid", "")), "rays_per_step": B, "pid": os.getpid()}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        try:
            nccl_v = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            nccl_v = None
        dist_info = {"world_size_seen_by_torch_distributed": dist.get_world_size(), "backend": dist.get_backend(),
                     "rccl_version": nccl_v, "env_world_size": world, "ranks": allr,
                     "distinct_devices": len({(r["pci_domain_id"], r["pci_bus_id"], r["device_index"]) for r in allr})}

    extra = {}
    if world == 1 and not args.no_extra and args.math == "fp32":
        ks, kw = max(5, args.steps // 2), max(2, args.warmup // 2)

        def leg(fld, md, ms=None, tbl=KERNELS, pk=PEAK_F32_MFMA):
            ms = ms if ms is not None else make_models(fld)
            st = make_step(ms, md)
            # the instrumented region first (it doubles as warm-up after the switch of workload), then the clean one
            r = timed(st, ks, kw)[1] if not args.no_kernel_events else {}
            d, _ = timed(st, ks, kw, events=False)
            return {"ms_per_step": d / ks * 1e3, "value": B * 192 * ks / d, "unit": "ray-samples/s",
                    "roofline": roofline_object(r, fld, md, B, ks, d, tbl, pk) if r else None}
        other_mode = "infer" if train else "train"
        extra[other_mode] = leg(field, other_mode, models)       # the same field at test time (eval.py:85-96) / in training
        other_field = "nerf" if field == "siren" else "siren"
        om = make_models(other_field)
        extra[other_field] = {
            "workload": "configs[1] with " + ("the reference's live ReLU NeRF 8x256 (system.py:181-192, 595 844 parameters)"
                                              if other_field == "nerf" else "the FiLM-SIREN field (9 FiLM layers x 256, 529 156 "
                                              "parameters)") + f" coarse+fine, batch_size={B}",
            "train": leg(other_field, "train", om), "infer": leg(other_field, "infer", om)}
        if not args.no_opt_in:
            # the same steps on the OPT-IN split-bf16 math (fp32-level accuracy on the bf16 matrix cores, same parity tests;
            # DESIGN.md section 4) -- reported beside the headline, never as `value`
            nerf_siren_amd.set_math("bf16x3")
            nm = om if other_field == "nerf" else models
            sm = models if field == "siren" else om
            extra["opt_in"] = {
                "math": "bf16x3 (exact 3-way bf16 split of both operands, 6 bf16 MFMA per product, fp32 accumulate)",
                "nerf_train": leg("nerf", "train", nm, KERNELS_BF16X3, 2500.0 / 6.0),
                "nerf_infer": leg("nerf", "infer", nm, KERNELS_BF16X3, 2500.0 / 6.0),
                "siren_train": leg("siren", "train", sm, KERNELS_BF16X3, 2500.0 / 6.0),
                "siren_infer": leg("siren", "infer", sm, KERNELS_BF16X3, 2500.0 / 6.0)}
            nerf_siren_amd.set_math("fp32")
        del om
        extra["eg3d"] = eg3d_bench(dev, ks, kw)

    if rank == 0:
        total_samples = world * B * 192 * args.steps
        fname = "FiLM-SIREN 9x256 (models/nerf.py:142-216)" if field == "siren" else "NeRF 8x256 (models/nerf.py:41-124)"
        out = {
            "metric": "ray-samples/sec, Blender-lego 400x400 synthetic rays, 64c+64f"
                      + (" (training step)" if train else " (inference, test_time)"),
            "value": total_samples / dt, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32" if args.math == "fp32" else "f32 (3xbf16 split, 6 bf16 MFMA per product)", "data": "synthetic",
            "config": {"workload": f"configs[1]: Blender-lego 400x400 rays, N_samples=64 N_importance=64, SIREN-MLP = {fname} "
                                   f"coarse+fine, batch_size={B} rays/GPU, mode={mode}"
                       if field == "siren" else
                       f"configs[1] shape (Blender-lego 400x400 rays, 64+64, batch_size={B} rays/GPU) with the reference's live "
                       f"field {fname} coarse+fine, mode={mode}",
                       "field": field, "rays_per_gpu": B, "global_rays_per_step": B * world, "samples_per_ray": 192, "mode": mode,
                       "parallelism": (f"dp{world}, one rank per GPU, {'fine-model all-reduce overlapped with the coarse backward' if not args.no_overlap else 'one joint all-reduce'}"
                                       if world > 1 else "single")},
            "rays_per_s": world * B * args.steps / dt,
            "roofline": roof,
        }
        if dist_info is not None:
            out["dist"] = dist_info
        if comm is not None:
            out["comm"] = comm
        out.update(extra)
        if world == 1 and train and not args.no_psnr and args.math == "fp32":
            out["psnr"] = psnr_check(dev, field)
            if not args.no_extra:
                out["psnr_" + ("nerf" if field == "siren" else "siren")] = psnr_check(dev, "nerf" if field == "siren" else "siren")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(field, mode, budget_s=args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
