#!/usr/bin/env python
"""Race soak of the backward kernels (GPU box): the same backward (dX chain + dW GEMM + slab reduction) launched N times on the
same saved image must give bit-identical gradients every time -- the dW GEMM's narrow tasks run a multi-buffer LDS-DMA ring
behind counted waits and raw barriers (csrc/dw_core.h dw_task4g16); a race there would show as a flipped bit now and then.
usage: python tools/soak_backward.py [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_siren_amd import NeRF, SemanticNeRF, ops, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
nerf = NeRF()
nerf.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_params(11, structured=False).items()})
nerf = nerf.to(dev)
siren = SemanticNeRF()
siren.load_state_dict({k: torch.from_numpy(v) for k, v in synth.siren_params(3).items()})
siren = siren.to(dev)
bad = 0
for n_rays, P in ((1, 41), (7, 143), (256, 64), (1024, 128), (4096, 128)):
    rays = torch.from_numpy(synth.blender_rays(n_rays, 5)).to(dev)
    z = torch.sort(torch.rand(n_rays, P, device=dev) * 4 + 2, -1)[0]
    g = torch.randn(n_rays * P, 4, device=dev)
    fr, ph = torch.randn(1, 2304, device=dev), torch.randn(1, 2304, device=dev)
    for name in ("nerf", "siren", "siren+cond"):
        if name == "nerf":
            _, saved = ops.nerf_forward_rays(nerf.packed(), rays, z, save=True)
            run = lambda: torch.cat([t.reshape(-1) for t in ops.nerf_backward_rays(nerf.packed(), rays, z, saved, g)])
        elif name == "siren":
            _, saved = ops.siren_forward_rays_train(siren.packed(), rays, z, fr, ph, n_rays)
            run = lambda: torch.cat([t.reshape(-1) for t in ops.siren_backward(siren.packed(), saved, g, fr, n_rays * P)])
        else:
            def run():
                gr, df, dp = ops.siren_backward(siren.packed(), saved, g, fr, n_rays * P, cond_grads=True)
                return torch.cat([t.reshape(-1) for t in gr] + [df.reshape(-1), dp.reshape(-1)])
        first = run().clone()
        assert torch.isfinite(first).all(), (name, n_rays, P)
        reps = N if n_rays * P <= 131072 else max(N // 10, 10)
        diff = sum(int(not torch.equal(run(), first)) for _ in range(reps))
        bad += diff
        print(f"{name:11s} {n_rays:5d} x {P:3d} points: {reps} repeats, {diff} differ", flush=True)
print("soak_backward", "OK" if bad == 0 else f"FAILED: {bad} launches differed")
sys.exit(0 if bad == 0 else 1)
