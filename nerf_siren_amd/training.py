"""The step right after render_rays in training, on flat buffers:

    FusedMSELoss   losses.py:10-20 (MSELoss) + its autograd + metrics.py:4-13 (psnr): ONE launch
                   (nerfmi_mse_loss) instead of ~10 elementwise/reduction launches
    FusedAdam      utils/__init__.py:20 (torch.optim.Adam(lr, eps=1e-8, weight_decay)): ONE launch per model
                   (nerfmi_adam_step) over a flat parameter buffer, its flat gradient (the buffer the HIP
                   backward wrote, ops.flat_views) and flat moments, instead of 7 multi-tensor launches

Both are drop-ins: FusedMSELoss()(results, targets) like losses.MSELoss; FusedAdam is a torch.optim.Optimizer
(param_groups / lr schedulers work), constructed from the models like the reference's get_optimizer(hparams, models).
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .parallel import flat_grad_alias


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_coarse, rgb_fine, targets, unit_grad):
        out4, gc, gf = ops.mse_loss(rgb_coarse, rgb_fine, targets, 1.0, want_grads=True)
        ctx.save_for_backward(*(t for t in (gc, gf) if t is not None))
        ctx.cfg = (gc is not None, gf is not None, bool(unit_grad))
        ctx.mark_non_differentiable(out4)
        return out4[0].clone() if not unit_grad else out4[:1].view(()), out4

    @staticmethod
    def backward(ctx, g_loss, _g_out4):
        has_c, has_f, unit = ctx.cfg
        saved = list(ctx.saved_tensors)
        gc = saved.pop(0) if has_c else None
        gf = saved.pop(0) if has_f else None
        if not unit:                                   # d loss / d x = saved gradient x upstream scalar
            gc = gc * g_loss if gc is not None else None
            gf = gf * g_loss if gf is not None else None
        return gc, gf, None, None


class FusedMSELoss(nn.Module):
    """losses.py:10-20.  forward(inputs, targets): inputs['rgb_coarse'] (+ inputs['rgb_fine']) vs targets (N,3).
    After a call, `.mse_coarse`, `.mse_fine`, `.psnr` hold device scalars from the same launch (metrics.py:4-13).
    unit_grad=True skips the multiplication of the saved gradients by the upstream gradient: valid when the loss
    is the root of backward() (loss.backward()), as in system.py:257-275."""

    def __init__(self, unit_grad: bool = False):
        super().__init__()
        self.unit_grad = bool(unit_grad)
        self.mse_coarse = self.mse_fine = self.psnr = None

    def forward(self, inputs, targets):
        loss, out4 = _MSEFn.apply(inputs.get('rgb_coarse'), inputs.get('rgb_fine'), targets, self.unit_grad)
        self.mse_coarse, self.mse_fine, self.psnr = out4[1], out4[2], out4[3]
        return loss


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (amsgrad=False) with one kernel launch per model.

    Each model's parameters are re-homed into ONE flat fp32 buffer (every p.data becomes a view of it, in the order
    of `model.param_list()` when the model has one -- the order in which the HIP backward lays out the flat
    gradient -- else `model.parameters()`).  state_dict()/load_state_dict() of the MODELS are unaffected."""

    def __init__(self, models, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        models = list(models) if isinstance(models, (list, tuple)) else [models]
        params = []
        self._entries = []
        for m in models:
            ps = list(m.param_list()) if hasattr(m, "param_list") else [p for p in m.parameters()]
            ps = [p for p in ps if p.requires_grad]
            if not ps:
                continue
            if any(p.dtype != torch.float32 or not p.is_cuda for p in ps):
                raise RuntimeError("FusedAdam: parameters must be fp32 tensors on the GPU (no CPU fallback)")
            n = sum(p.numel() for p in ps)
            flat = torch.empty(n, device=ps[0].device, dtype=torch.float32)
            offs, off = [], 0
            with torch.no_grad():
                for p in ps:
                    k = p.numel()
                    flat[off:off + k].copy_(p.data.reshape(-1))
                    p.data = flat[off:off + k].view(p.shape)
                    offs.append(off)
                    off += k
            self._entries.append(dict(model=m, params=ps, offsets=offs, flat=flat, exp_avg=torch.zeros_like(flat),
                                      exp_avg_sq=torch.zeros_like(flat), scratch=None, step=0))
            params += ps
            if hasattr(m, "mark_parameters_changed"):
                m.mark_parameters_changed()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _flat_grad(self, e):
        ps = e["params"]
        if ps[0].grad is None:
            return None
        b = flat_grad_alias(ps)
        if b is not None and b.numel() == e["flat"].numel():
            return b                                    # the buffer the HIP backward wrote: no copy
        if e["scratch"] is None:
            e["scratch"] = torch.empty_like(e["flat"])
        for p, o in zip(ps, e["offsets"]):              # foreign / accumulated gradients: gather
            k = p.numel()
            if p.grad is None:
                e["scratch"][o:o + k].zero_()
            else:
                e["scratch"][o:o + k].copy_(p.grad.reshape(-1))
        return e["scratch"]

    # checkpoint / resume: the moments live in flat buffers outside torch's per-parameter state
    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"steps": [e["step"] for e in self._entries],
                       "exp_avg": [e["exp_avg"].clone() for e in self._entries],
                       "exp_avg_sq": [e["exp_avg_sq"].clone() for e in self._entries]}
        return sd

    def load_state_dict(self, state_dict):
        fused = state_dict.get("fused")
        if fused is None:
            raise ValueError("not a FusedAdam state_dict (no 'fused' entry)")
        if len(fused["exp_avg"]) != len(self._entries):
            raise ValueError("FusedAdam.load_state_dict: different number of models")
        for g, saved in zip(self.param_groups, state_dict["param_groups"]):
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in saved:
                    g[k] = saved[k]
            if "initial_lr" in saved:
                g["initial_lr"] = saved["initial_lr"]
        with torch.no_grad():
            for e, m, v in zip(self._entries, fused["exp_avg"], fused["exp_avg_sq"]):
                e["exp_avg"].copy_(m.to(e["exp_avg"].device))
                e["exp_avg_sq"].copy_(v.to(e["exp_avg_sq"].device))
        steps = fused["steps"] if "steps" in fused else [int(fused["n_steps"])] * len(self._entries)
        for e, n in zip(self._entries, steps):
            e["step"] = int(n)

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        for e in self._entries:
            gflat = self._flat_grad(e)
            if gflat is None:
                continue                                # no gradient this step: torch.optim.Adam skips the parameter too,
            e["step"] += 1                              # and its bias correction counts only the steps it took
            ops.adam_step(e["flat"], gflat, e["exp_avg"], e["exp_avg_sq"], e["step"], g["lr"], g["betas"], g["eps"],
                          g["weight_decay"], grad_scale)
            if hasattr(e["model"], "mark_parameters_changed"):
                e["model"].mark_parameters_changed()
        return loss
