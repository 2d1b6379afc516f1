#!/usr/bin/env python
"""Dense field query throughput: sigma grid of extract_color_mesh.py:117-140 on the device.
usage (GPU box): python tools/bench_grid.py [N]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import nerf_siren_amd
from nerf_siren_amd import NeRF, field_query as FQ, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
m = NeRF()
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_params(2, structured=False).items()})
m = m.to(dev)
rng = ((-1.2, 1.2),) * 3
for math in ("fp32", "bf16x3"):
    nerf_siren_amd.set_math(math)
    for full in (False, True):
        FQ.sigma_grid(m, 64, *rng, return_rgbsigma=full)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        FQ.sigma_grid(m, N, *rng, return_rgbsigma=full)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"case": f"sigma_grid N={N} ({'rgb+sigma' if full else 'sigma only'})", "math": math,
                          "ms": dt * 1e3, "points_per_s": N ** 3 / dt}))
nerf_siren_amd.set_math("fp32")
