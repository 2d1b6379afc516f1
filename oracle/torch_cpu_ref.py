"""The reference's CPU *PyTorch* path, restated  --  TEST INFRASTRUCTURE, NOT PRODUCT.

north_star / BASELINE.md section 3 ask for "the reference's CPU PyTorch path timed on the same box's host cores".  The
reference's files cannot travel to the GPU box, so this module restates the same torch op sequence --
    models/rendering.py:105-262  render_rays / inference (sampling, chunked MLP calls, compositing, sample_pdf, sort)
    models/rendering.py:22-67    sample_pdf
    models/nerf.py:21-38, :83-124 Embedding, NeRF.forward
    models/nerf.py:134-151, :201-216 the FiLM-SIREN field (round 3: the headline field of configs[1])
    losses.py:10-20, utils/__init__.py:20  MSE(coarse)+MSE(fine), torch.optim.Adam(lr 5e-4, eps 1e-8)
-- with torch CPU ops (autograd for the backward), functional style over a dict of parameter tensors.  It exists for ONE
purpose: bench.py's `cpu_baseline` leg (and the CPU test that pins it).  The numpy oracle (nerf_oracle.py) remains the
parity checker; this file is the *speed* baseline: same ops, same chunking (32 768 points per MLP call), same dtype, so
its time is what the reference's own code would take on these cores.

In-place ReLU matters for the timing as well as for fidelity (nn.ReLU(True), nerf.py:68): out of place the training
step is 30 % slower on 8 cores.  Measured in the build container (8 cores): this file 1.82 s per training step of 1024
rays, the imported reference 1.83-1.94 s; inference 0.36 s vs 0.35 s.

Parity status: PINNED -- tests/test_oracle_golden.py::test_torch_cpu_ref_vs_reference runs it on the G7 fixtures (outputs
of the imported reference with the captured random draws): outputs agree to 1e-6, gradients to 1e-5 relative.
Only tests/ and bench.py's cpu_baseline leg may import this module; the product package never does.
"""
from __future__ import annotations

import os
import time

import torch
import torch.nn.functional as Fn


def embed(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """nerf.py:21-38: [x, sin(2^k x), cos(2^k x)]_k as one concatenation."""
    parts = [x]
    for k in range(n_freqs):
        f = float(2 ** k)
        parts += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(parts, -1)


def nerf_mlp(P: dict, x: torch.Tensor, sigma_only: bool = False) -> torch.Tensor:
    """nerf.py:83-124 on a dict of state_dict tensors."""
    xyz = x[:, :63]
    h = xyz
    for i in range(1, 9):
        if i == 5:                                           # skips=[4]
            h = torch.cat([xyz, h], -1)
        h = torch.relu_(Fn.linear(h, P[f"xyz_encoding_{i}.0.weight"], P[f"xyz_encoding_{i}.0.bias"]))   # nn.ReLU(True), :68
    sigma = Fn.linear(h, P["sigma.weight"], P["sigma.bias"])
    if sigma_only:
        return sigma
    final = Fn.linear(h, P["xyz_encoding_final.weight"], P["xyz_encoding_final.bias"])
    d = torch.relu_(Fn.linear(torch.cat([final, x[:, 63:]], -1), P["dir_encoding.0.weight"], P["dir_encoding.0.bias"]))
    rgb = torch.sigmoid(Fn.linear(d, P["rgb.0.weight"], P["rgb.0.bias"]))
    return torch.cat([rgb, sigma], -1)


def siren_mlp(P: dict, x: torch.Tensor, sigma_only: bool = False) -> torch.Tensor:
    """The FiLM-SIREN field of configs[1] as the reference's render_rays would evaluate it: SemanticNeRF.
    forward_with_frequencies_phase_shifts (nerf.py:201-216; FiLMLayer :142-151, UniformBoxWarp(51) :134-140) behind the
    field interface of rendering.py:105-159 -- forward(x, sigma_only) on the EMBEDDED rows, whose first three xyz / direction
    channels are the raw xyz / direction (Embedding keeps its input, nerf.py:35).  The same adapter tools/make_psnr_golden.py
    --siren trained the reference through (RefSirenField) and the product ships as nerf_siren_amd.SirenField.
    P: the 22 state_dict tensors plus 'frequencies' / 'phase_shifts' (1, 2304), one conditioning row for all points."""
    xyz = x[None, :, :3]
    dirs = torch.zeros_like(xyz) if sigma_only else x[None, :, 63:66]
    freq = P["frequencies"] * 15 + 30                        # :202
    phase = P["phase_shifts"]

    def film(name, h, lo, hi):                                # FiLMLayer.forward, :147-151
        h = Fn.linear(h, P[name + ".layer.weight"], P[name + ".layer.bias"])
        f = freq[..., lo:hi].unsqueeze(1).expand_as(h)
        p = phase[..., lo:hi].unsqueeze(1).expand_as(h)
        return torch.sin(f * h + p)
    h = xyz * (2 / 51)                                        # gridwarper, :204
    for i in range(8):
        h = film(f"network.{i}", h, 256 * i, 256 * (i + 1))
    sigma = Fn.linear(h, P["final_layer.weight"], P["final_layer.bias"])
    if sigma_only:
        return sigma[0]
    c = film("color_layer_sine", torch.cat([dirs, h], -1), 2048, 2304)
    rgb = torch.sigmoid(Fn.linear(c, P["color_layer_linear.0.weight"], P["color_layer_linear.0.bias"]))
    return torch.cat([rgb, sigma], -1)[0]


def field_mlp(P: dict, x: torch.Tensor, sigma_only: bool = False) -> torch.Tensor:
    return siren_mlp(P, x, sigma_only) if "frequencies" in P else nerf_mlp(P, x, sigma_only)


def sample_pdf(bins, weights, n_importance, det, u=None, eps=1e-5):
    """rendering.py:22-67."""
    n, nw = weights.shape
    weights = weights + eps
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    if det:
        u = torch.linspace(0, 1, n_importance).expand(n, n_importance)
    elif u is None:
        u = torch.rand(n, n_importance)
    u = u.contiguous()
    inds = torch.searchsorted(cdf.detach(), u, right=True)
    below, above = (inds - 1).clamp_min(0), inds.clamp_max(nw)
    pair = torch.stack([below, above], -1).view(n, 2 * n_importance)
    cdf_g = torch.gather(cdf, 1, pair).view(n, n_importance, 2)
    bins_g = torch.gather(bins, 1, pair).view(n, n_importance, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom[denom < eps] = 1
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def _field_pass(P, rays_o, rays_d, dir_emb, z, noise_std, white_back, weights_only, chunk, noise):
    """inference() of rendering.py:105-190."""
    n, s = z.shape
    xyz = (rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z.unsqueeze(2)).view(-1, 3)
    if not weights_only:
        dir_rep = torch.repeat_interleave(dir_emb, repeats=s, dim=0)
    outs = []
    for i in range(0, xyz.shape[0], chunk):                  # MLP in chunks of `chunk` POINTS (:140-150)
        e = embed(xyz[i:i + chunk], 10)
        if not weights_only:
            e = torch.cat([e, dir_rep[i:i + chunk]], 1)
        outs.append(field_mlp(P, e, sigma_only=weights_only))
    out = torch.cat(outs, 0)
    if weights_only:
        sigmas = out.view(n, s)
        rgbs = None
    else:
        out = out.view(n, s, 4)
        rgbs, sigmas = out[..., :3], out[..., 3]
    deltas = z[:, 1:] - z[:, :-1]
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1)
    deltas = deltas * torch.norm(rays_d.unsqueeze(1), dim=-1)
    if noise is None:
        noise = torch.randn(sigmas.shape)
    alphas = 1 - torch.exp(-deltas * torch.relu(sigmas + noise * noise_std))
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    weights = alphas * torch.cumprod(shifted, -1)[:, :-1]
    wsum = weights.sum(1)
    if weights_only:
        return None, None, weights, wsum
    rgb = torch.sum(weights.unsqueeze(-1) * rgbs, -2)
    depth = torch.sum(weights * z, -1)
    if white_back:
        rgb = rgb + 1 - wsum.unsqueeze(-1)
    return rgb, depth, weights, wsum


def render_rays(params, rays, N_samples=64, use_disp=False, perturb=0.0, noise_std=1.0, N_importance=0,
                chunk=1024 * 32, white_back=False, test_time=False, rng=None):
    """rendering.py:199-262.  params = [coarse, fine] dicts of tensors; rng: optional dict of injected draws
    (perturb_rand, noise_coarse, u, noise_fine) -- otherwise torch.rand / torch.randn as the reference."""
    rng = rng or {}
    n = rays.shape[0]
    rays_o, rays_d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
    dir_emb = embed(rays_d, 4)
    t = torch.linspace(0, 1, N_samples)
    z = near * (1 - t) + far * t if not use_disp else 1 / (1 / near * (1 - t) + 1 / far * t)
    z = z.expand(n, N_samples)
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper, lower = torch.cat([mid, z[:, -1:]], -1), torch.cat([z[:, :1], mid], -1)
        r = rng["perturb_rand"] if "perturb_rand" in rng else torch.rand(z.shape)
        z = lower + (upper - lower) * (perturb * r)
    rgb, depth, w, op = _field_pass(params[0], rays_o, rays_d, dir_emb, z, noise_std, white_back, test_time, chunk,
                                    rng.get("noise_coarse"))
    res = {"opacity_coarse": op} if test_time else {"rgb_coarse": rgb, "depth_coarse": depth, "opacity_coarse": op}
    if N_importance > 0:
        zmid = 0.5 * (z[:, :-1] + z[:, 1:])
        z_new = sample_pdf(zmid, w[:, 1:-1], N_importance, det=(perturb == 0), u=rng.get("u")).detach()
        z, _ = torch.sort(torch.cat([z, z_new], -1), -1)
        rgb, depth, w, op = _field_pass(params[1], rays_o, rays_d, dir_emb, z, noise_std, white_back, False, chunk,
                                        rng.get("noise_fine"))
        res.update(rgb_fine=rgb, depth_fine=depth, opacity_fine=op)
    return res


# -------------------------------------------------------------------------------------------------
# host description + the bounded timing sample used by bench.py
# -------------------------------------------------------------------------------------------------
def host_info() -> dict:
    """CPU model, physical cores, logical CPUs this process may run on, torch's threading configuration."""
    model, pairs, logical = "unknown", set(), 0
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "processor":
                    logical += 1
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None:
                    pairs.add((phys, core))
                    phys = core = None
        if phys is not None:
            pairs.add((phys, core))
    except OSError:
        pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    logical = logical or (os.cpu_count() or 1)
    physical = len(pairs) or logical
    quota = cgroup_cpu_quota()
    return {"cpu_model": model, "physical_cores": physical, "nproc": logical, "allowed_cpus": allowed,
            "cgroup_cpu_quota": quota}


def cgroup_cpu_quota():
    """CPUs' worth of time this process's cgroup may use (cpu.max / cfs_quota), or None when unlimited.  A container on a
    256-thread host is typically limited this way rather than by its affinity mask; starting one thread per physical
    core of the HOST under such a quota makes the throttled threads spin on each other (measured on the 1-GPU box: 128
    threads -> 13 s per step, an order of magnitude slower than 16)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2
            q, p = f.read().split()
            return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:          # cgroup v1
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def pick_threads(info: dict) -> dict:
    """Thread count for the timing: the candidates (cgroup quota, affinity, physical cores and powers of two below them)
    are tried on the path's dominant operation -- the (32 768 x 256) x (256 x 256) fp32 GEMM of one MLP layer on one
    point chunk -- and the fastest wins, so neither an invisible CPU limit nor SMT siblings can make the baseline a
    strawman.  Returns {threads, gemm_gflops: {n: rate}}."""
    cap = min(info["physical_cores"], info["allowed_cpus"])
    cands = {cap}
    if info.get("cgroup_cpu_quota"):
        cands.add(max(1, int(info["cgroup_cpu_quota"])))
    n = 4
    while n < cap:
        cands.add(n)
        n *= 2
    a, w = torch.randn(32768, 256), torch.randn(256, 256)
    rates = {}
    for n in sorted(cands):
        torch.set_num_threads(n)
        for _ in range(2):
            a @ w
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 0.25:
            a @ w
            reps += 1
        rates[n] = reps * 2 * 32768 * 256 * 256 / (time.perf_counter() - t0) / 1e9
    best = max(rates, key=rates.get)
    return {"threads": best, "gemm_gflops": {str(k): round(v, 1) for k, v in rates.items()}}


def timed_sample(mode: str, params_np, make_rays, make_target, budget_s: float = 12.0, n_rays: int = 1024,
                 max_steps: int = 64) -> dict:
    """Run the reference's step (train: forward + MSE x2 + backward + Adam; infer: test_time render under no_grad) on
    batches of n_rays rays until ~budget_s seconds of work are done, after one untimed warm-up step.
    params_np: [coarse, fine] dicts of numpy arrays; make_rays(i) / make_target(i) -> numpy arrays."""
    info = host_info()
    info.update(pick_threads(info))
    torch.set_num_threads(info["threads"])
    # the conditioning rows of a FiLM-SIREN field are inputs, not parameters (SirenField keeps them fixed)
    params = [{k: torch.from_numpy(v.copy()).requires_grad_(mode == "train" and k not in ("frequencies", "phase_shifts"))
               for k, v in p.items()} for p in params_np]
    opt = torch.optim.Adam([t for p in params for t in p.values() if t.requires_grad], lr=5e-4, eps=1e-8) if mode == "train" else None

    def step(i):
        rays = torch.from_numpy(make_rays(i))
        if mode == "train":
            tgt = torch.from_numpy(make_target(i))
            res = render_rays(params, rays, 64, False, 1.0, 1.0, 64, 1024 * 32, True, False)
            loss = ((res["rgb_coarse"] - tgt) ** 2).mean() + ((res["rgb_fine"] - tgt) ** 2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
        else:
            with torch.no_grad():
                render_rays(params, rays, 64, False, 0.0, 0.0, 64, 1024 * 32, True, True)

    step(0)                                                  # warm-up (thread pool, allocator)
    times = []
    i = 1
    while sum(times) < budget_s and len(times) < max_steps:
        t0 = time.perf_counter()
        step(i)
        times.append(time.perf_counter() - t0)
        i += 1
    total = sum(times)
    info.update(steps=len(times), seconds=total, best_step_s=min(times), n_rays=n_rays,
                ray_samples_per_s=len(times) * n_rays * 192 / total, ray_samples_per_s_best=n_rays * 192 / min(times),
                parallel_info=torch.__config__.parallel_info())
    return info
