#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[4]): EG3D tri-plane renderer, planes (1,3,32,256,256),
M rays x (64+64) samples forward, and the 128^3 dense run_model query.  Prints one JSON line per case."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nerf_siren_amd import ImportanceRenderer, OSGDecoder, synth


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    dev = torch.device("cuda:0")
    planes = torch.from_numpy(synth.triplanes(1, res=256)).to(dev)
    dec = OSGDecoder(32, {"decoder_lr_mul": 1.0, "decoder_output_dim": 3})
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in synth.osg_params(1).items()})
    dec = dec.to(dev)
    ren = ImportanceRenderer()
    opts = dict(synth.EG3D_OPTIONS)
    for M in (1024, 4096, 16384):
        o, d = synth.eg3d_rays(M, 3)
        o, d = torch.from_numpy(o[None]).to(dev), torch.from_numpy(d[None]).to(dev)
        with torch.no_grad():
            dt = timeit(lambda: ren(planes, dec, o, d, opts))
        print(json.dumps({"case": f"ImportanceRenderer fwd M={M} 64+64", "ms": dt * 1e3, "samples_per_s": M * 128 / dt,
                          "rays_per_s": M / dt}))
    # training step of the renderer: forward + backward to the planes and the decoder (system.py:17-169 path)
    for M in (1024, 4096):
        o, d = synth.eg3d_rays(M, 3)
        o, d = torch.from_numpy(o[None]).to(dev), torch.from_numpy(d[None]).to(dev)
        pl = planes.clone().requires_grad_(True)
        for p_ in dec.parameters():
            p_.requires_grad_(True)

        def train_step():
            pl.grad = None
            for p_ in dec.parameters():
                p_.grad = None
            out = ren(pl, dec, o, d, opts)
            (out[0].square().mean() + out[1].mean()).backward()

        dt = timeit(train_step, n=10)
        print(json.dumps({"case": f"ImportanceRenderer fwd+bwd M={M} 64+64", "ms": dt * 1e3, "samples_per_s": M * 128 / dt}))
    g = np.linspace(-1.5, 1.5, 128, dtype=np.float32)
    pts = torch.from_numpy(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(1, -1, 3)).to(dev)
    with torch.no_grad():
        dt = timeit(lambda: ren.run_model(planes, dec, pts, None, opts))
    nbytes = pts.shape[1] * 12 * 128
    print(json.dumps({"case": "run_model dense 128^3", "ms": dt * 1e3, "points_per_s": pts.shape[1] / dt,
                      "texel_GBps": nbytes / dt / 1e9}))


if __name__ == "__main__":
    main()
