"""render_rays() -- the reference's renderer API (models/rendering.py:70-262)
on the MI355X-native kernels.

Call sequence per batch of N rays (one HIP stream, no host sync):
  sample_stratified -> nerf_forward_rays(coarse) -> composite
  -> importance_resample (z_mid, cdf scan, searchsorted, lerp, merge-sort)
  -> nerf_forward_rays(fine) -> composite
Training (grad enabled): each field pass is one autograd node whose backward is
composite_backward -> nerf_backward_rays; no gradient crosses sample_pdf
(rendering.py:54 cdf.detach(), :244 .detach()).
"""
from __future__ import annotations

import torch

from . import ops

__all__ = ["render_rays", "sample_pdf", "set_math", "get_math"]

_MATH = "fp32"


def set_math(mode: str):
    """Arithmetic of the MLP matrix products: 'fp32' (default, exact fp32 MFMA) or 'bf16x3' (opt-in: exact three-way
    bf16 splits on the bf16 matrix cores, fp32-level accuracy): all forward passes (inference and the training
    forward that saves activations), the backward dX chain and the 256 x 256 tasks of the dW GEMM, for both fields."""
    global _MATH
    if mode not in ("fp32", "bf16x3"):
        raise ValueError("math mode must be 'fp32' or 'bf16x3'")
    _MATH = mode


def get_math() -> str:
    return _MATH


def _field_infer(model, rays, zz, sigma_only):
    if _MATH == "bf16x3":
        return ops.nerf_forward_rays_fast(model.packed(), model.packed_fast(), rays, zz, sigma_only=sigma_only)
    return ops.nerf_forward_rays(model.packed(), rays, zz, sigma_only=sigma_only)


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5, u=None):
    """models/rendering.py:22-67.  `u` optionally injects the torch.rand draw."""
    if eps != 1e-5:
        raise NotImplementedError("eps is fixed to the reference's 1e-5")
    return ops.sample_pdf(bins, weights, N_importance, det=det, u=u)


def _claim_grad_target(model, device):
    """The views a backward pass may write a model's gradients into, or None (fresh buffer).

    A data-parallel reducer may have given the model a slice of ONE buffer shared by all models
    (parallel.FlatGradAllReduce).  The backward writes there ONCE per backward pass.  If the model is applied
    several times in one graph (NeRFSystem.forward chunks a batch larger than hp.chunk; two render_rays calls summed into
    one loss), autograd still holds the first contribution -- an alias of the target -- when the second node runs (p.grad
    stays None until all of them have arrived), so every later contribution of the same pass goes to a fresh buffer and
    autograd sums them.  The claim is released by an engine callback at the end of the pass.  Across passes: a
    parameter that still holds a gradient in that very memory (zero_grad(set_to_none=False) / accumulation) also
    forces a fresh buffer."""
    target = getattr(model, "_grad_target", None)
    if target is None or target.device != device:
        return None
    if getattr(model, "_grad_target_claimed", False):
        if getattr(model, "_grad_ready_hook", None) is not None:
            raise RuntimeError("FlatGradAllReduce(overlap=True) needs each model applied once per backward pass "
                               "(its gradient slice is already being reduced)")
        return None
    lo, hi = target.data_ptr(), target.data_ptr() + target.numel() * 4
    if any(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in model.parameters()):
        return None
    model._grad_target_claimed = True

    def release():
        model._grad_target_claimed = False
    torch.autograd.Variable._execution_engine.queue_callback(release)
    return model.grad_views(target)


def _grad_ready(model, claimed):
    """The model's gradient kernels have been enqueued into its slice of the joint buffer: let an overlapping reducer
    (parallel.FlatGradAllReduce(overlap=True)) put the slice on the wire while the other model's backward runs."""
    hook = getattr(model, "_grad_ready_hook", None)
    if claimed is not None and hook is not None:
        hook(model._grad_target)


class FieldRender(torch.autograd.Function):
    """inference() of rendering.py:105-190 for the full (rgb, sigma) branch:
    (rays, z) -> rgb, depth, opacity, weights; gradients to the field's parameters (the 24 tensors of a NeRF, the 22
    of a FiLM-SIREN field behind nerf.SirenField)."""

    @staticmethod
    def forward(ctx, model, rays, z, noise, noise_std, white_back, keep, *params):
        # keep: None, or a (dict, key) pair that receives the saved-activation image of this pass (parity tests read the
        # ReLU sign pattern the kernels actually used out of it)
        siren = hasattr(model, "field_rays")
        ctx.n_field_params = len(model.param_list())      # SirenField may append its trainable conditioning rows behind them
        if siren:
            packed = model.model.packed()
            ctx.fast = model.model.packed_fast() if _MATH == "bf16x3" else None
            field, saved = ops.siren_forward_rays_train(packed, rays, z, model.frequencies, model.phase_shifts, rays.shape[0],
                                                        fast=ctx.fast)
        else:
            packed = model.packed()
            if _MATH == "bf16x3":
                field, saved = ops.nerf_forward_rays_fast(packed, model.packed_fast(), rays, z, sigma_only=False, save=True)
            else:
                field, saved = ops.nerf_forward_rays(packed, rays, z, sigma_only=False, save=True)
            ctx.fast = model.packed_fast() if _MATH == "bf16x3" else None
        # noise: the injected randn tensor, a (seed, offset, segment) Philox key (drawn inside the compositor, forward and
        # backward alike) or None
        philox = noise if isinstance(noise, tuple) else None
        noise = None if philox is not None else noise
        weights, rgb, depth, opacity = ops.composite(field, z, rays, noise, noise_std, white_back, philox=philox)
        if keep is not None:
            keep[0][keep[1]] = saved
        ctx.save_for_backward(rays, z, noise if noise is not None else rays.new_empty(0), field, saved, packed)
        ctx.cfg = (noise is not None, float(noise_std), bool(white_back), siren)
        ctx.philox = philox
        ctx.model = model
        ctx.mark_non_differentiable(weights)
        ctx.set_materialize_grads(False)        # absent d/d(depth, opacity) arrive as None -> NULL in the C ABI
        return rgb, depth, opacity, weights

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_opacity, _g_w):
        rays, z, noise, field, saved, packed = ctx.saved_tensors
        has_noise, noise_std, white_back, siren = ctx.cfg
        n_params = len(ops.SIREN_PARAM_ORDER if siren else ops.PARAM_ORDER)
        n_cond = len(ctx.needs_input_grad) - 7 - n_params            # 0, or 2: SirenField's frequencies, phase_shifts
        if g_rgb is None and g_depth is None and g_opacity is None:
            return (None,) * (7 + n_params + n_cond)
        grad_field = ops.composite_backward(field, z, rays, noise if has_noise else None, noise_std, white_back,
                                            g_rgb, g_depth, g_opacity, philox=ctx.philox)
        out = _claim_grad_target(ctx.model, rays.device)
        cond = ()
        if siren and n_cond:
            grads, d_f, d_p = ops.siren_backward(packed, saved, grad_field, ctx.model.frequencies, z.numel(), grads=out,
                                                 cond_grads=True, fast=ctx.fast)
            cond = (d_f.view_as(ctx.model.frequencies), d_p.view_as(ctx.model.phase_shifts))
        elif siren:
            grads = ops.siren_backward(packed, saved, grad_field, ctx.model.frequencies, z.numel(), grads=out, fast=ctx.fast)
        else:
            grads = ops.nerf_backward_rays(packed, rays, z, saved, grad_field, grads=out, fast=ctx.fast)
        _grad_ready(ctx.model, out)
        return (None, None, None, None, None, None, None, *grads, *cond)


class SirenPoints(torch.autograd.Function):
    """SemanticNeRF.forward_with_frequencies_phase_shifts (nerf.py:201-216) with autograd w.r.t. the 22 parameters and --
    when the call shares one conditioning row -- w.r.t. that row (frequencies, phase_shifts)."""

    @staticmethod
    def forward(ctx, model, points, dirs, freq, phase, points_per_cond, *params):
        packed = model.packed()
        out, saved = ops.siren_forward_points_train(packed, points, dirs, freq, phase, points_per_cond)
        ctx.save_for_backward(saved, packed, freq)
        ctx.model, ctx.ppc = model, int(points_per_cond)
        return out

    @staticmethod
    def backward(ctx, g_out):
        saved, packed, freq = ctx.saved_tensors
        out = _claim_grad_target(ctx.model, g_out.device)
        d_f = d_p = None
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            grads, d_f, d_p = ops.siren_backward(packed, saved, g_out.contiguous(), freq, ctx.ppc, grads=out, cond_grads=True)
        else:
            grads = ops.siren_backward(packed, saved, g_out.contiguous(), freq, ctx.ppc, grads=out)
        _grad_ready(ctx.model, out)
        return (None, None, None, d_f, d_p, None, *grads)


class EmbeddedField(torch.autograd.Function):
    """NeRF.forward(x, sigma_only) (models/nerf.py:83-124) with autograd w.r.t. the 24 parameters: the module-level API
    on pre-embedded rows.  sigma_only runs the same saved forward with a zero direction embedding and back-propagates
    [0, 0, 0, d sigma]: the gradients of the colour branch come out as exact zeros (the reference leaves them None)."""

    @staticmethod
    def forward(ctx, model, x, sigma_only, *params):
        packed = model.packed()
        if sigma_only:
            x = torch.cat([x, x.new_zeros((x.shape[0], 27))], 1)
        out, saved = ops.nerf_forward_embedded_train(packed, x.contiguous())
        ctx.save_for_backward(saved, packed)
        ctx.model, ctx.sigma_only = model, bool(sigma_only)
        return out[:, 3:4].contiguous() if sigma_only else out

    @staticmethod
    def backward(ctx, g_out):
        saved, packed = ctx.saved_tensors
        if ctx.sigma_only:
            g_out = torch.cat([g_out.new_zeros((g_out.shape[0], 3)), g_out], 1)
        out = _claim_grad_target(ctx.model, g_out.device)
        grads = ops.nerf_backward_points(packed, saved, g_out.contiguous(), grads=out)
        _grad_ready(ctx.model, out)
        return (None, None, None, *grads)


def _rng(rng, key, shape):
    """The injected draw `key`, shape-checked, or None."""
    t = None if rng is None else rng.get(key)
    if t is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"rng['{key}'] has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


def render_rays(models, embeddings, rays, N_samples=64, use_disp=False, perturb=0, noise_std=1, N_importance=0,
                chunk=1024 * 32, white_back=False, test_time=False, _cls_num=6, network=None, *, rng=None, aux=None):
    """Same positional signature and result dict as models/rendering.py:70-83, :262.

    models: [coarse] or [coarse, fine] nerf_siren_amd.NeRF; embeddings: [Embedding(3,10),
    Embedding(3,4)] (their frequencies are compiled into the fused kernel; the
    list is validated, not called).  rays (N,8) = [o, d, near, far] on the GPU.
    chunk: accepted for compatibility; the fused kernels need no point chunking.
    rng (keyword-only, optional): dict of injected random draws in the reference's
    order -- 'perturb_rand' (N,S) [rendering.py:221], 'noise_coarse' (N,S) [:170],
    'u' (N,F) [:47], 'noise_fine' (N,S+F); missing entries are drawn INSIDE the consuming kernels from
    Philox streams keyed by torch.manual_seed (ops.next_draw_key; ops.render_draws materialises the same streams).  'z_fine' (N,S+F), when given,
    replaces the merged depths of :247 (parity tests condition the fine pass on the reference's own depths:
    sample_pdf is ill-conditioned in ~zero-weight bins).
    aux (keyword-only, optional): a dict that receives the intermediates 'z_coarse', 'weights_coarse', 'z_fine' and, in
    training mode, 'saved_coarse' / 'saved_fine' (the fields' saved-activation images, csrc/mlp_layout.h S_* rows).
    """
    if len(embeddings) != 2 or getattr(embeddings[0], "N_freqs", None) != 10 or \
            getattr(embeddings[1], "N_freqs", None) != 4:
        raise NotImplementedError("render_rays is compiled for Embedding(3,10) / Embedding(3,4) (system.py:181-182)")
    if rays.dim() != 2 or rays.shape[1] != 8:
        raise ValueError(f"rays must be (N_rays, 8), got {tuple(rays.shape)}")
    rays = rays.detach().contiguous().float()
    N = rays.shape[0]
    dev = rays.device
    S, F = int(N_samples), int(N_importance)
    model_coarse = models[0]
    train = torch.is_grad_enabled() and any(p.requires_grad for m in models for p in m.parameters())

    # Random draws (rendering.py:221 rand, :170 randn, :47 rand, :170 randn): injected tensors win (parity tests); every
    # other draw is made INSIDE the consuming kernel from one Philox key per call (seed and offset of torch's CUDA
    # generator of the device, which the call advances: ops.next_draw_key) -- no aten distribution launches, no tensors of draws.
    draw_key = ops.next_draw_key(dev) if (perturb != 0 or noise_std != 0) else None

    # no_grad pass with nothing injected and nothing asked back: the whole sequence from ONE C-ABI call
    # (nerfmi_render_rays_fused: same kernels, same Philox key -> bit-identical to the call-by-call sequence below)
    if not train and rng is None and aux is None and _MATH == "fp32" and N > 0 and perturb >= 0:
        ms = list(models[:2 if F > 0 else 1])
        kinds = {("siren" if hasattr(m, "field_rays") else "nerf") for m in ms}
        if len(kinds) == 1 and (F == 0 or len(models) > 1):
            siren = kinds == {"siren"}
            packed = [(m.model if siren else m).packed() for m in ms]
            conds = [m.cond_rows() for m in ms] if siren else [None]
            return ops.render_rays_fused(int(siren), packed[0], packed[-1] if F > 0 else None, conds[0],
                                         conds[-1] if F > 0 else None, rays, S, F, use_disp, float(perturb),
                                         float(noise_std), white_back, test_time, draw_key)

    pr = _rng(rng, "perturb_rand", (N, S)) if perturb > 0 else None
    z = ops.sample_stratified(rays, S, use_disp, float(perturb), pr, philox=draw_key)

    def noise_for(key, P, seg):
        # rendering.py:170 draws randn even when noise_std == 0 (result x0); the
        # native path skips the draw -- identical outputs.
        if noise_std == 0:
            return None, None
        t = _rng(rng, key, (N, P))
        return (t, None) if t is not None else (None, (draw_key[0], draw_key[1], seg))

    def full_pass(model, zz, key, seg):
        noise, philox = noise_for(key, zz.shape[1], seg)
        if train and any(p.requires_grad for p in model.parameters()):
            cond = model.cond_param_list() if hasattr(model, "cond_param_list") else []
            rgb, depth, opacity, weights = FieldRender.apply(model, rays, zz, noise if philox is None else philox,
                                                            float(noise_std), bool(white_back),
                                                            None if aux is None else (aux, "saved_" + key[6:]),
                                                            *model.param_list(), *cond)
        elif hasattr(model, "field_rays"):                     # FiLM-SIREN adapter (nerf.SirenField)
            field = model.field_rays(rays, zz, sigma_only=False)
            weights, rgb, depth, opacity = ops.composite(field, zz, rays, noise, noise_std, white_back, philox=philox)
        else:
            field = _field_infer(model, rays, zz, False)
            weights, rgb, depth, opacity = ops.composite(field, zz, rays, noise, noise_std, white_back, philox=philox)
        return rgb, depth, opacity, weights

    if test_time:
        # weights_only branch (rendering.py:227-231): sigma-only coarse MLP
        if hasattr(model_coarse, "field_rays"):
            sig = model_coarse.field_rays(rays, z, sigma_only=True)
        else:
            sig = _field_infer(model_coarse, rays, z, True)
        noise, philox = noise_for("noise_coarse", S, 1)
        weights_coarse, _, _, op = ops.composite(sig, z, rays, noise, noise_std, white_back, sigma_only=True, philox=philox)
        result = {"opacity_coarse": op}
    else:
        rgb, depth, op, weights_coarse = full_pass(model_coarse, z, "noise_coarse", 1)
        result = {"rgb_coarse": rgb, "depth_coarse": depth, "opacity_coarse": op}

    if F > 0:
        u = _rng(rng, "u", (N, F)) if perturb != 0 else None                   # det = (perturb == 0), :243
        z_fine = None if rng is None else rng.get("z_fine")
        if z_fine is None:
            z_fine = ops.importance_resample(z, weights_coarse, F, u, philox=draw_key if perturb != 0 else None)
        elif tuple(z_fine.shape) != (N, S + F):
            raise ValueError(f"rng['z_fine'] has shape {tuple(z_fine.shape)}, expected {(N, S + F)}")
        if aux is not None:
            aux.update(z_coarse=z, weights_coarse=weights_coarse, z_fine=z_fine)
        rgb, depth, op, _ = full_pass(models[1], z_fine, "noise_fine", 3)
        result["rgb_fine"] = rgb
        result["depth_fine"] = depth
        result["opacity_fine"] = op
    return result
