#!/usr/bin/env python
"""Stability soak (GPU box): N training steps on the teacher scene with the fused loss/optimizer, alternating the
exact-fp32 and the opt-in split-bf16 math every 100 steps; checks finiteness, a falling loss and flat memory.
usage: python tools/soak.py [steps] [nerf|siren]   (siren: the FiLM-SIREN field through the same loop, in-kernel draws)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import nerf_siren_amd
from nerf_siren_amd import Embedding, NeRF, render_rays, synth
from nerf_siren_amd.parallel import FlatGradAllReduce
from nerf_siren_amd.training import FusedAdam, FusedMSELoss

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
FIELD = sys.argv[2] if len(sys.argv) > 2 else "nerf"
dev = torch.device("cuda:0")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g15_psnr.npz"))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)          # noqa: E731
rays, tgt, val_rays, val_tgt = T(g["rays"]), T(g["target"]), T(g["val_rays"]), T(g["val_target"])
ms = []
for seed in (11, 12):
    if FIELD == "siren":
        from nerf_siren_amd import SemanticNeRF, SirenField
        sm = SemanticNeRF()
        sm.load_state_dict({k: torch.from_numpy(v) for k, v in synth.siren_params(seed).items()})
        m = SirenField(sm, torch.from_numpy(synth.hash_normal((1, 2304), 10 + seed) * 0.3),
                       torch.from_numpy(synth.hash_normal((1, 2304), 20 + seed)))
    else:
        m = NeRF()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_params(seed, structured=False).items()})
    ms.append(m.to(dev))
emb = [Embedding(3, 10), Embedding(3, 4)]
opt, loss_fn, red = FusedAdam(ms, lr=5e-4 if FIELD == "nerf" else 5e-5), FusedMSELoss(unit_grad=True), FlatGradAllReduce(ms, 1)
sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[steps // 2, 3 * steps // 4], gamma=0.5)
losses, mem = [], []
t0 = time.perf_counter()
for step in range(steps):
    nerf_siren_amd.set_math("bf16x3" if (step // 100) % 2 and FIELD == "nerf" else "fp32")
    idx = torch.randint(0, rays.shape[0], (1024,), device=dev)
    res = render_rays(ms, emb, rays[idx], 64, False, 1.0, 0.0 if FIELD == "nerf" else 0.5, 64, 1 << 15, True, False)
    loss = loss_fn(res, tgt[idx])
    opt.zero_grad()
    loss.backward()
    red.all_reduce(average=False)
    opt.step()
    sched.step()
    if step % 100 == 99:
        losses.append(float(loss.detach()))
        mem.append(torch.cuda.memory_allocated() // 2 ** 20)
        assert np.isfinite(losses[-1]), step
nerf_siren_amd.set_math("fp32")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
with torch.no_grad():
    r = render_rays(ms, emb, val_rays, 64, False, 0, 0, 64, 1 << 15, True, False)
psnr = float(-10 * torch.log10(((r["rgb_fine"] - val_tgt) ** 2).mean()))
for m in ms:
    for k, p in m.named_parameters():
        assert torch.isfinite(p).all(), k
print(json.dumps({"steps": steps, "s": round(dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "loss_every_100": [round(x, 5) for x in losses],
                  "mem_MiB_every_100": mem, "val_psnr_db": round(psnr, 3)}))
assert losses[-1] < losses[0] and max(mem[2:]) - min(mem[2:]) <= 64
