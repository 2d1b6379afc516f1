#!/usr/bin/env python
"""PSNR parity fixture (BASELINE.md section 3 'teacher-scene protocol'): train the REFERENCE
(models/rendering.py + models/nerf.py, CPU) for a few hundred Adam steps on a small synthetic teacher
scene with all random draws injected, and record the validation PSNR trajectory.
tests/test_gpu_parity.py::test_psnr_parity repeats the same run on the HIP path and compares.

The teacher images are rendered by the reference itself from a seeded structured NeRF ("teacher") on
lego-like cameras at 32x32.  Losses/metrics follow losses.py:15-20 (MSE coarse + MSE fine) and
metrics.py:4-13 (psnr = -10 log10 mse).  Optimiser: Adam(lr 5e-4, eps 1e-8) (utils/__init__.py:20).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import models.rendering as R                      # noqa: E402
from models.nerf import Embedding, NeRF           # noqa: E402
from nerf_siren_amd import synth                  # noqa: E402

CFG = dict(res=32, n_train_views=8, batch=512, steps=240, eval_every=60, lr=5e-4, S=64, F=64)


def view_rays(res, view_seed):
    """All res*res rays of one lego-like camera (same geometry as synth.blender_rays)."""
    uv = synth.hash_uniform((1, 2), view_seed * 7919 + 11)
    c2w = synth._look_at_c2w(float(uv[0, 0]) * np.deg2rad(60.0), float(uv[0, 1]) * 2 * np.pi, synth.LEGO_RADIUS)
    focal = np.float32(0.5 * res / np.tan(0.5 * synth.LEGO_ANGLE_X))
    j, i = np.meshgrid(np.arange(res, dtype=np.float32), np.arange(res, dtype=np.float32), indexing="ij")
    dirs = np.stack([(i - res / 2) / focal, -(j - res / 2) / focal, -np.ones_like(i)], -1).reshape(-1, 3).astype(np.float32)
    d = (dirs @ c2w[:, :3].T).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    o = np.broadcast_to(c2w[:, 3], d.shape)
    nf = np.tile(np.array([[2.0, 6.0]], np.float32), (d.shape[0], 1))
    return np.concatenate([o, d, nf], -1).astype(np.float32)


step_rng = synth.psnr_step_rng            # shared with tests/test_gpu_parity.py::test_psnr_parity
batch_indices = synth.psnr_batch_indices


def patched(rng_list):
    class P:
        def __enter__(self):
            self.r, self.n = torch.rand, torch.randn
            q = list(rng_list)
            torch.rand = lambda *s, **k: torch.from_numpy(q.pop(0).copy())
            torch.randn = lambda *s, **k: torch.zeros(*s)
            return self

        def __exit__(self, *a):
            torch.rand, torch.randn = self.r, self.n
    return P()


def main():
    torch.set_num_threads(8)
    c = CFG
    emb = [Embedding(3, 10), Embedding(3, 4)]

    def model(p):
        m = NeRF()
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
        return m

    teacher = [model(synth.nerf_params(7, sigma_bias=-0.5)), model(synth.nerf_params(8, sigma_bias=0.5))]
    rays = np.concatenate([view_rays(c["res"], 300 + v) for v in range(c["n_train_views"])], 0)
    val_rays = view_rays(c["res"], 399)
    with torch.no_grad():
        tgt = R.render_rays(teacher, emb, torch.from_numpy(rays), c["S"], False, 0, 0, c["F"], 1 << 15, True, True)["rgb_fine"].numpy()
        val_tgt = R.render_rays(teacher, emb, torch.from_numpy(val_rays), c["S"], False, 0, 0, c["F"], 1 << 15, True, True)["rgb_fine"].numpy()
    print("teacher rendered", tgt.shape, "mean", tgt.mean(0), "std", tgt.std(0))

    student = [model(synth.nerf_params(11, structured=False)), model(synth.nerf_params(12, structured=False))]
    opt = torch.optim.Adam([p for m in student for p in m.parameters()], lr=c["lr"], eps=1e-8)
    psnr, losses = [], []

    def evaluate():
        with torch.no_grad():
            r = R.render_rays(student, emb, torch.from_numpy(val_rays), c["S"], False, 0, 0, c["F"], 1 << 15, True, False)
        mse = float(((r["rgb_fine"].numpy() - val_tgt) ** 2).mean())
        return -10 * np.log10(mse)

    t0 = time.time()
    for step in range(c["steps"] + 1):
        if step % c["eval_every"] == 0:
            psnr.append(evaluate())
            print(f"step {step} val psnr {psnr[-1]:.3f} ({time.time()-t0:.0f}s)", flush=True)
        if step == c["steps"]:
            break
        idx = batch_indices(step, rays.shape[0], c["batch"])
        rg = step_rng(step, c["batch"], c["S"], c["F"])
        with patched([rg["perturb_rand"], rg["u"]]):
            res = R.render_rays(student, emb, torch.from_numpy(rays[idx]), c["S"], False, 1.0, 0.0, c["F"], 1 << 15, True, False)
        t = torch.from_numpy(tgt[idx])
        loss = ((res["rgb_coarse"] - t) ** 2).mean() + ((res["rgb_fine"] - t) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g15_psnr.npz"), rays=rays, target=tgt, val_rays=val_rays,
                        val_target=val_tgt, psnr=np.array(psnr, np.float32), losses=np.array(losses, np.float32),
                        **{"cfg_" + k: v for k, v in c.items()})
    print("psnr", psnr)


if __name__ == "__main__":
    main()
