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
# --long (g19): the protocol at a horizon where a 0.1 dB bar bites -- 1 500 steps of 1024 rays (configs[1]'s batch), the
# learning rate halved at steps 600 and 1 200 (MultiStepLR, utils/__init__.py:33-50), validation on a FULL 400x400 view
# (160 000 rays) every 300 steps.  Rays are regenerated from the view seeds (synth.psnr_rays); the fixture stores the
# teacher's images only.
CFG_LONG = dict(res=64, n_train_views=24, val_res=400, batch=1024, steps=1500, eval_every=300, lr=5e-4, S=64, F=64,
                lr_milestones=[600, 1200], lr_gamma=0.5)

view_rays = synth.view_rays


class BallScene(torch.nn.Module):
    """--scene ball: ONE smooth object -- a soft-edged unit ball whose colour varies slowly with position -- that an 8x256
    NeRF fits to a high, flat PSNR within the 1 500 steps: the converged regime in which a 0.1 dB bar between two trainers
    is meaningful (see DESIGN.md section 2 on the moving-target scene)."""

    def forward(self, x, sigma_only=False):
        p = x[:, :3]
        r = torch.linalg.norm(p, dim=-1, keepdim=True)
        sigma = 30.0 * torch.sigmoid((1.0 - r) * 8.0)
        if sigma_only:
            return sigma
        rgb = 0.5 + 0.35 * torch.stack([torch.sin(1.5 * p[:, 0] + 0.3), torch.sin(1.5 * p[:, 1] + 1.1),
                                        torch.sin(1.5 * p[:, 2] + 2.0)], -1)
        return torch.cat([rgb, sigma], -1)


class AnalyticScene(torch.nn.Module):
    """--scene spheres: a crisp, learnable teacher for the long protocol -- three coloured soft-edged spheres and a slab,
    evaluated on the raw xyz channels (the first three) of the embedded input.  It stands where the reference expects a NeRF:
    forward(x, sigma_only) -> (B,4) [rgb, sigma] / (B,1), so the reference's own render_rays renders the target images."""
    CENTRES = torch.tensor([[0.55, 0.25, 0.10], [-0.60, -0.35, 0.25], [0.05, -0.55, -0.45]])
    RADII = torch.tensor([0.55, 0.45, 0.40])
    COLOURS = torch.tensor([[0.90, 0.20, 0.15], [0.15, 0.75, 0.25], [0.20, 0.30, 0.90]])

    def forward(self, x, sigma_only=False):
        p = x[:, :3]
        d = torch.linalg.norm(p[:, None, :] - self.CENTRES[None], dim=-1)                   # (B,3)
        occ = torch.sigmoid((self.RADII[None] - d) * 25.0)                                    # soft membership
        slab = torch.sigmoid((0.9 - p[:, :2].abs().max(-1).values) * 25.0) * torch.sigmoid((-0.75 - p[:, 2]) * 25.0) \
            * torch.sigmoid((p[:, 2] + 0.95) * 25.0)
        w = torch.cat([occ, slab[:, None]], -1)                                               # (B,4)
        cols = torch.cat([self.COLOURS, torch.tensor([[0.85, 0.80, 0.55]])], 0)
        stripes = 0.75 + 0.25 * torch.sin(6.0 * p[:, 0:1]) * torch.sin(6.0 * p[:, 1:2])       # mid-frequency texture
        rgb = (w[:, :, None] * cols[None]).sum(1) / w.sum(-1, keepdim=True).clamp_min(1e-6) * stripes
        sigma = 40.0 * w.max(-1, keepdim=True).values
        return sigma if sigma_only else torch.cat([rgb.clamp(0, 1), sigma], -1)


class RefSirenField(torch.nn.Module):
    """--siren: the reference's SemanticNeRF (models/nerf.py:157-216) behind the interface the reference's render_rays
    expects of a field -- forward(x, sigma_only) on the embedded input, whose first three xyz / dir channels are the raw
    xyz / direction (Embedding keeps the input, nerf.py:35).  One conditioning row for all points.  The same adapter the
    product ships as nerf_siren_amd.SirenField."""

    def __init__(self, m, freq, phase):
        super().__init__()
        self.m, self.freq, self.phase = m, torch.from_numpy(freq.copy()), torch.from_numpy(phase.copy())

    def forward(self, x, sigma_only=False):
        xyz = x[None, :, :3]
        dirs = torch.zeros_like(xyz) if sigma_only else x[None, :, 63:66]
        out = self.m.forward_with_frequencies_phase_shifts(xyz, self.freq, self.phase, dirs)[0]      # [rgb, sigma]
        return out[:, 3:] if sigma_only else out


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
    torch.set_num_threads(int(os.environ.get("PSNR_THREADS", "8")))
    long = "--long" in sys.argv
    c = CFG_LONG if long else CFG
    emb = [Embedding(3, 10), Embedding(3, 4)]

    def model(p):
        m = NeRF()
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
        return m

    spheres, ball = "spheres" in sys.argv, "ball" in sys.argv
    if ball:
        teacher = [BallScene(), BallScene()]
    elif spheres:
        teacher = [AnalyticScene(), AnalyticScene()]
    else:
        teacher = [model(synth.nerf_params(7, sigma_bias=-0.5)), model(synth.nerf_params(8, sigma_bias=0.5))]
    rays = np.concatenate([view_rays(c["res"], 300 + v) for v in range(c["n_train_views"])], 0)
    val_rays = view_rays(c.get("val_res", c["res"]), 399)

    def render(models, rr, test_time):
        out = []
        with torch.no_grad():
            for i in range(0, rr.shape[0], 1 << 15):                 # system.py:205 / eval.py:80: 32 768-ray chunks
                out.append(R.render_rays(models, emb, torch.from_numpy(rr[i:i + (1 << 15)]), c["S"], False, 0, 0, c["F"],
                                         1 << 15, True, test_time)["rgb_fine"].numpy())
        return np.concatenate(out, 0)

    tgt, val_tgt = render(teacher, rays, True), render(teacher, val_rays, True)
    print("teacher rendered", tgt.shape, "mean", tgt.mean(0), "std", tgt.std(0))

    siren = "--siren" in sys.argv
    if siren:
        import models.nerf as MN
        MN.np = np                      # nerf.py:131 uses np without importing it
        c = dict(c, lr=5e-5)
        student = []
        for seed in (11, 12):
            sm = MN.SemanticNeRF()
            sm.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.siren_params(seed).items()})
            student.append(RefSirenField(sm, synth.hash_normal((1, 2304), 10 + seed), synth.hash_normal((1, 2304), 20 + seed)))
    else:
        student = [model(synth.nerf_params(11, structured=False)), model(synth.nerf_params(12, structured=False))]
    opt = torch.optim.Adam([p for m in student for p in m.parameters()], lr=c["lr"], eps=1e-8)
    sched = (torch.optim.lr_scheduler.MultiStepLR(opt, milestones=c["lr_milestones"], gamma=c["lr_gamma"])
             if "lr_milestones" in c else None)
    psnr, losses = [], []

    def evaluate():
        mse = float(((render(student, val_rays, False).astype(np.float64) - val_tgt) ** 2).mean())
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
        if sched is not None:
            sched.step()
        losses.append(float(loss))
    if long:
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", ("g19b_psnr_ball.npz" if ball else "g19s_psnr_spheres.npz") if (spheres or ball) else "g19_psnr_long.npz"),
                            target=tgt, val_target=val_tgt,
                            psnr=np.array(psnr, np.float32), losses=np.array(losses, np.float32),
                            **{"cfg_" + k: v for k, v in c.items()})
    else:
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g15s_psnr_siren.npz" if siren else "g15_psnr.npz"), rays=rays, target=tgt, val_rays=val_rays,
                            val_target=val_tgt, psnr=np.array(psnr, np.float32), losses=np.array(losses, np.float32),
                            **{"cfg_" + k: v for k, v in c.items()})
    print("psnr", psnr)


if __name__ == "__main__":
    main()
