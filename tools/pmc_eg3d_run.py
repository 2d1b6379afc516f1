#!/usr/bin/env python
"""Workload of the EG3D counter passes (tools/pmc_eg3d.sh): 3 x the dense 128^3 run_model query (triplane_kernel<1,false>)
and 3 x forward+backward of ImportanceRenderer at M = 4096 rays x (64+64) samples (triplane_kernel / triplane_backward_kernel),
planes (1,3,32,256,256) -- BASELINE.json configs[4].  Few launches, so every dispatch in the counter file is attributable."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nerf_siren_amd import ImportanceRenderer, OSGDecoder, synth

dev = torch.device("cuda:0")
planes = torch.from_numpy(synth.triplanes(1, res=256)).to(dev)
dec = OSGDecoder(32, {"decoder_lr_mul": 1.0, "decoder_output_dim": 3})
dec.load_state_dict({k: torch.from_numpy(v) for k, v in synth.osg_params(1).items()})
dec = dec.to(dev)
ren = ImportanceRenderer()
opts = dict(synth.EG3D_OPTIONS)
g = np.linspace(-1.5, 1.5, 128, dtype=np.float32)
pts = torch.from_numpy(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(1, -1, 3)).to(dev)
with torch.no_grad():
    for _ in range(3):
        ren.run_model(planes, dec, pts, None, opts)
torch.cuda.synchronize()
M = 4096
o, d = synth.eg3d_rays(M, 3)
o, d = torch.from_numpy(o[None]).to(dev), torch.from_numpy(d[None]).to(dev)
pl = planes.clone().requires_grad_(True)
for _ in range(3):
    pl.grad = None
    for p_ in dec.parameters():
        p_.grad = None
    out = ren(pl, dec, o, d, opts)
    (out[3].square().mean() + out[4].mean() + out[0].square().mean()).backward()
torch.cuda.synchronize()
print("done")
