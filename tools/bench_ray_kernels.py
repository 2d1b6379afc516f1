#!/usr/bin/env python
"""HBM roofline of the per-ray kernels (SURVEY section 8d 'algorithmic bytes'): sampler, compositor (fwd + bwd),
importance resampling at a batch large enough to leave the launch-latency regime.
usage (GPU box): python tools/bench_ray_kernels.py [n_rays]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_siren_amd import ops, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
rays = torch.from_numpy(synth.blender_rays(4096, 3)).to(dev).repeat(N // 4096, 1)
S, F = 64, 64
P = S + F
PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


prand = torch.rand(N, S, device=dev)
z = ops.sample_stratified(rays, S, False, 1.0, prand)
field_c = torch.randn(N * S, 4, device=dev)
field_f = torch.randn(N * P, 4, device=dev)
noise_c, noise_f = torch.randn(N, S, device=dev), torch.randn(N, P, device=dev)
w, _, _, _ = ops.composite(field_c, z, rays, noise_c, 1.0, True)
u = torch.rand(N, F, device=dev)
zf = ops.importance_resample(z, w, F, u)
g3, g1 = torch.randn(N, 3, device=dev), torch.randn(N, device=dev)
cases = [
    ("sample_stratified (jitter)", lambda: ops.sample_stratified(rays, S, False, 1.0, prand), 32 + 4 * S + 4 * S),
    ("composite coarse (64, noise)", lambda: ops.composite(field_c, z, rays, noise_c, 1.0, True), 16 * S + 4 * S + 32 + 4 * S + 4 * S + 20),
    ("composite fine (128, noise)", lambda: ops.composite(field_f, zf, rays, noise_f, 1.0, True), 16 * P + 4 * P + 32 + 4 * P + 4 * P + 20),
    ("composite_backward fine (128)", lambda: ops.composite_backward(field_f, zf, rays, noise_f, 1.0, True, g3, g1, g1),
     16 * P + 4 * P + 32 + 4 * P + 20 + 16 * P),
    ("importance_resample (64 -> 128, u)", lambda: ops.importance_resample(z, w, F, u), 8 * S + 4 * F + 4 * P),
]
for name, fn, bytes_per_ray in cases:
    t = timeit(fn)
    gbs = N * bytes_per_ray / t / 1e9
    print(json.dumps({"kernel": name, "n_rays": N, "us": round(t * 1e6, 1), "algorithmic_bytes_per_ray": bytes_per_ray,
                      "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 3)}))
