"""Tensor-level wrappers over the C ABI (include/nerfmi.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; all
compute happens in libnerfmi.so.  Every op requires contiguous fp32 tensors on
a ROCm device and raises otherwise -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _req(t: torch.Tensor, name: str, shape=None, dtype=torch.float32):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: nerf_siren_amd runs on the GPU only (tensor is on {t.device}); "
                           "there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if shape is not None:
        if t.dim() != len(shape) or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
    if not t.is_contiguous():
        t = t.contiguous()
    return t


# Profiling hook: bench.py brackets the field-MLP launches with HIP events on the launch stream.  hook(name, n_per_ray)
# returns None or an object with .done(), called right after the launch was enqueued (same stream).
_PROFILE_HOOK = None


def set_profile_hook(hook):
    """hook(name: str, n_per_ray: int) -> None | object with .done(); None removes the hook."""
    global _PROFILE_HOOK
    _PROFILE_HOOK = hook


class _Span:
    __slots__ = ("h",)

    def __init__(self, name, n_per_ray):
        self.h = _PROFILE_HOOK(name, n_per_ray) if _PROFILE_HOOK is not None else None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.h is not None:
            self.h.done()
        return False


def _u64(v):
    return int(v) & (2 ** 64 - 1)


def version() -> int:
    return _lib.lib().nerfmi_version()


# Kernel-span profiler of the library (csrc/render.hip): HIP events on the launch stream around every field-MLP kernel.
def profile_start():
    check(_lib.lib().nerfmi_profile_start(), "profile_start")


def profile_stop():
    check(_lib.lib().nerfmi_profile_stop(), "profile_stop")


def profile_report():
    """-> {(kernel tag, points per launch): (launches, total ms)} of the spans recorded since profile_start();
    waits for them."""
    n = _lib.lib().nerfmi_profile_report(None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    _lib.lib().nerfmi_profile_report(buf, int(n) + 1)
    out = {}
    for line in buf.value.decode().splitlines():
        tag, units, cnt, ms = line.split("\t")
        out[(tag, int(units))] = (int(cnt), float(ms))
    return out


# --------------------------------------------------------------------------- a1 (one call per no_grad pass)
def render_rays_fused(field_kind, packed_coarse, packed_fine, cond_coarse, cond_fine, rays, n_samples, n_importance,
                      use_disp, perturb, noise_std, white_back, test_time, draw_key):
    """nerfmi_render_rays_fused: the whole no_grad pass of render_rays from ONE C-ABI call -> the reference's result dict.
    field_kind 0: NeRF packed blobs; 1: FiLM-SIREN packed blobs + cond_* = (2, 2304) [frequencies; phase_shifts]."""
    rays = _req(rays, "rays", (None, 8))
    n, dev = rays.shape[0], rays.device
    S, F = int(n_samples), int(n_importance)
    ws = torch.empty(_lib.lib().nerfmi_render_rays_workspace_floats(n, S, F, int(bool(test_time))), device=dev,
                     dtype=torch.float32)
    # outputs out of one allocation: [rgb_c 3 | depth_c | opacity_c | rgb_f 3 | depth_f | opacity_f] x n
    o = torch.empty((10, n), device=dev, dtype=torch.float32)
    rgb_c, depth_c, op_c = o[0:3].view(-1).view(n, 3), o[3], o[4]
    rgb_f, depth_f, op_f = o[5:8].view(-1).view(n, 3), o[8], o[9]
    seed, offset = draw_key if draw_key is not None else (0, 0)
    check(_lib.lib().nerfmi_render_rays_fused(int(field_kind), ptr(packed_coarse), ptr(packed_fine), ptr(cond_coarse),
                                              ptr(cond_fine), ptr(rays), n, S, F, int(bool(use_disp)), float(perturb),
                                              float(noise_std), int(bool(white_back)), int(bool(test_time)), _u64(seed),
                                              _u64(offset), ptr(ws), None if test_time else ptr(rgb_c),
                                              None if test_time else ptr(depth_c), ptr(op_c), ptr(rgb_f) if F else None,
                                              ptr(depth_f) if F else None, ptr(op_f) if F else None, _stream(rays)),
          "render_rays_fused")
    result = {"opacity_coarse": op_c} if test_time else {"rgb_coarse": rgb_c, "depth_coarse": depth_c, "opacity_coarse": op_c}
    if F:
        result.update(rgb_fine=rgb_f, depth_fine=depth_f, opacity_fine=op_f)
    return result


# --------------------------------------------------------------------------- a2
def sample_stratified(rays, n_samples, use_disp=False, perturb=0.0, perturb_rand=None, philox=None):
    """philox = (seed, offset): draw the jitter in the kernel (segment 0 of that stream) instead of reading perturb_rand."""
    rays = _req(rays, "rays", (None, 8))
    n = rays.shape[0]
    z = torch.empty((n, n_samples), device=rays.device, dtype=torch.float32)
    if perturb > 0 and perturb_rand is None and philox is not None:
        check(_lib.lib().nerfmi_sample_stratified_philox(ptr(rays), _u64(philox[0]), _u64(philox[1]), n, n_samples,
                                                         int(bool(use_disp)), float(perturb), ptr(z), _stream(rays)),
              "sample_stratified_philox")
        return z
    if perturb > 0:
        perturb_rand = _req(perturb_rand, "perturb_rand", (n, n_samples))
        if perturb_rand is None:
            raise ValueError("perturb > 0 needs perturb_rand")
    check(_lib.lib().nerfmi_sample_stratified(ptr(rays), ptr(perturb_rand) if perturb > 0 else None, n, n_samples,
                                              int(bool(use_disp)), float(perturb), ptr(z), _stream(rays)),
          "sample_stratified")
    return z


def _generator(device):
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return torch.cuda.default_generators[idx]


def next_draw_key(device):
    """(seed, offset) of the next render_rays call on `device`, taken from -- and advancing -- torch's own CUDA generator
    of that device: seed = its seed, offset = its Philox offset in units of four 32-bit draws (one unit per call; the
    kernels derive every draw of the call from (seed, offset, segment, element), csrc/rays.hip).  So the draws obey
    torch.manual_seed / torch.cuda.manual_seed (offset back to 0), torch.cuda.get_rng_state / set_rng_state carry them
    across a checkpoint (a resumed run continues the stream instead of replaying it), and torch's own random ops on the
    device and this path never hand out the same offset twice."""
    gen = _generator(device)
    off = gen.get_offset()
    gen.set_offset(off + 4)
    return gen.initial_seed(), off // 4


def get_draw_state(device):
    """{'seed', 'offset'} of the next draw key of `device` (what a checkpoint stores; torch.cuda.get_rng_state(device) holds
    the same information)."""
    gen = _generator(device)
    return {"seed": gen.initial_seed(), "offset": gen.get_offset() // 4}


def set_draw_state(device, state):
    """Restore get_draw_state()'s dict: the next render_rays call on `device` draws what it would have drawn then."""
    gen = _generator(device)
    gen.manual_seed(int(state["seed"]))
    gen.set_offset(int(state["offset"]) * 4)


def render_draws(device, n_rays, S, F, perturb=True, noise=True, seed=None, offset=None):
    """The random draws of one render_rays call from ONE launch (nerfmi_render_draws): dict with 'perturb_rand' (N,S),
    'u' (N,F) when `perturb`, 'noise_coarse' (N,S), 'noise_fine' (N,S+F) when `noise` -- views of one allocation.
    seed / offset default to next_draw_key(device): torch's CUDA generator of the device (torch.manual_seed controls it)."""
    if seed is None or offset is None:
        k = next_draw_key(device)
        seed = k[0] if seed is None else seed
        offset = k[1] if offset is None else offset
    sizes = [n_rays * S if perturb else 0, n_rays * S if noise else 0, n_rays * F if perturb else 0,
             n_rays * (S + F) if (noise and F > 0) else 0]
    buf = torch.empty(sum(sizes), device=device, dtype=torch.float32)
    segs, off = [], 0
    for n in sizes:
        segs.append(buf[off:off + n] if n else None)
        off += n
    check(_lib.lib().nerfmi_render_draws(_u64(seed), _u64(offset), sizes[0], ptr(segs[0]),
                                         sizes[1], ptr(segs[1]), sizes[2], ptr(segs[2]), sizes[3], ptr(segs[3]),
                                         torch.cuda.current_stream(device).cuda_stream), "render_draws")
    out = {}
    if sizes[0]:
        out["perturb_rand"] = segs[0].view(n_rays, S)
    if sizes[1]:
        out["noise_coarse"] = segs[1].view(n_rays, S)
    if sizes[2]:
        out["u"] = segs[2].view(n_rays, F)
    if sizes[3]:
        out["noise_fine"] = segs[3].view(n_rays, S + F)
    return out


# --------------------------------------------------------------------------- a5
def embed(x, n_freqs):
    x = _req(x, "x", (None, 3))
    out = torch.empty((x.shape[0], 3 * (2 * n_freqs + 1)), device=x.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_embed(ptr(x), x.shape[0], n_freqs, ptr(out), _stream(x)), "embed")
    return out


# --------------------------------------------------------------------------- a6
PARAM_ORDER = ([f"xyz_encoding_{i}.0.{k}" for i in range(1, 9) for k in ("weight", "bias")]
               + ["xyz_encoding_final.weight", "xyz_encoding_final.bias", "dir_encoding.0.weight",
                  "dir_encoding.0.bias", "sigma.weight", "sigma.bias", "rgb.0.weight", "rgb.0.bias"])
PARAM_SHAPES = ([(256, 63), (256,)] + [(256, 256), (256,)] * 3 + [(256, 319), (256,)] + [(256, 256), (256,)] * 3
                + [(256, 256), (256,), (128, 283), (128,), (1, 256), (1,), (3, 128), (3,)])


PARAM_SIZES = [int(torch.Size(s).numel()) for s in PARAM_SHAPES]
PARAM_NUMEL = sum(PARAM_SIZES)          # 595 844


def flat_views(flat):
    """The 24 parameter-shaped views of one contiguous buffer (state_dict order): a model's gradient is
    ONE allocation, so the data-parallel all-reduce needs no flatten/unflatten copies."""
    out, off = [], 0
    for shape, n in zip(PARAM_SHAPES, PARAM_SIZES):
        out.append(flat[off:off + n].view(shape))
        off += n
    return out


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def nerf_pack(params, out=None):
    """params: the 24 tensors in PARAM_ORDER -> packed fragment-order blob."""
    ps = []
    for name, shape, t in zip(PARAM_ORDER, PARAM_SHAPES, params):
        ps.append(_req(t.detach(), name, shape))
    n = _lib.lib().nerfmi_nerf_packed_floats()
    if out is None:
        out = torch.empty(n, device=ps[0].device, dtype=torch.float32)
    check(_lib.lib().nerfmi_nerf_pack(_ptr_array(ps), ptr(out), _stream(out)), "nerf_pack")
    return out


def nerf_saved_floats(n_points):
    return _lib.lib().nerfmi_nerf_saved_floats(n_points)


def nerf_forward_rays(packed, rays, z, sigma_only=False, save=False):
    rays = _req(rays, "rays", (None, 8))
    z = _req(z, "z", (rays.shape[0], None))
    packed = _req(packed, "packed", (_lib.lib().nerfmi_nerf_packed_floats(),))
    n, p = z.shape
    out = torch.empty((n * p, 1 if sigma_only else 4), device=rays.device, dtype=torch.float32)
    saved = None
    if save:
        saved = torch.empty(nerf_saved_floats(n * p), device=rays.device, dtype=torch.float32)
    with _Span("nerf_forward_rays", p):
        check(_lib.lib().nerfmi_nerf_forward_rays(ptr(packed), ptr(rays), ptr(z), n, p, int(bool(sigma_only)), ptr(out),
                                                  ptr(saved), _stream(rays)), "nerf_forward_rays")
    if save:
        saved._nerfmi_math = "f32"        # the fp32 and split-bf16 paths keep their images in different element orders
    return (out, saved) if save else out


def nerf_pack_fast(packed):
    """bf16x3 split image of the forward weights (csrc/mlp_bf16x3.hip)."""
    out = torch.empty(_lib.lib().nerfmi_nerf_fast_bytes(), device=packed.device, dtype=torch.uint8)
    check(_lib.lib().nerfmi_nerf_pack_fast(ptr(packed), ptr(out), _stream(packed)), "nerf_pack_fast")
    return out


def nerf_forward_rays_fast(packed, fast, rays, z, sigma_only=False, save=False):
    rays = _req(rays, "rays", (None, 8))
    z = _req(z, "z", (rays.shape[0], None))
    n, p = z.shape
    out = torch.empty((n * p, 1 if sigma_only else 4), device=rays.device, dtype=torch.float32)
    saved = torch.empty(nerf_saved_floats(n * p), device=rays.device, dtype=torch.float32) if save else None
    with _Span("nerf_forward_rays_fast", p):
        check(_lib.lib().nerfmi_nerf_forward_rays_fast(ptr(packed), ptr(fast), ptr(rays), ptr(z), n, p,
                                                       int(bool(sigma_only)), ptr(out), ptr(saved), _stream(rays)),
              "nerf_forward_rays_fast")
    if save:
        saved._nerfmi_math = "bf16x3"
    return (out, saved) if save else out


def nerf_forward_embedded(packed, x, sigma_only=False):
    x = _req(x, "x", (None, 63 if sigma_only else 90))
    packed = _req(packed, "packed", (_lib.lib().nerfmi_nerf_packed_floats(),))
    out = torch.empty((x.shape[0], 1 if sigma_only else 4), device=x.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_nerf_forward_embedded(ptr(packed), ptr(x), x.shape[0], int(bool(sigma_only)), ptr(out),
                                                  _stream(x)), "nerf_forward_embedded")
    return out


def nerf_forward_embedded_train(packed, x):
    """Training forward of NeRF.forward(x) on pre-embedded rows x (B, 90) -> (out (B,4), saved)."""
    x = _req(x, "x", (None, 90))
    packed = _req(packed, "packed", (_lib.lib().nerfmi_nerf_packed_floats(),))
    n = x.shape[0]
    out = torch.empty((n, 4), device=x.device, dtype=torch.float32)
    saved = torch.empty(nerf_saved_floats(n), device=x.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_nerf_forward_embedded_train(ptr(packed), ptr(x), n, ptr(out), ptr(saved), _stream(x)),
          "nerf_forward_embedded_train")
    saved._nerfmi_math = "f32"
    return out, saved


def nerf_backward_points(packed, saved, grad_out, grads=None):
    """Backward of nerf_forward_embedded_train: grad_out (B,4) -> the 24 gradients (the inputs carry none)."""
    grad_out = _req(grad_out, "grad_out", (None, 4))
    n = grad_out.shape[0]
    if grads is None:
        grads = flat_views(torch.empty(PARAM_NUMEL, device=grad_out.device, dtype=torch.float32))
    ws = torch.empty(_lib.lib().nerfmi_nerf_backward_workspace_floats(n), device=grad_out.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_nerf_backward_rays(ptr(packed), None, None, n, 1, ptr(saved), ptr(grad_out), _ptr_array(grads),
                                               ptr(ws), _stream(grad_out)), "nerf_backward_rays")
    return grads


def nerf_backward_rays(packed, rays, z, saved, grad_out, grads=None, fast=None):
    """-> list of 24 gradient tensors (PARAM_ORDER), written not accumulated.
    fast: the split-bf16 image (nerf_pack_fast) to run the dX chain on the bf16 matrix cores (opt-in math)."""
    rays = _req(rays, "rays", (None, 8))
    z = _req(z, "z", (rays.shape[0], None))
    n, p = z.shape
    grad_out = _req(grad_out, "grad_out", (n * p, 4))
    if grads is None:
        grads = flat_views(torch.empty(PARAM_NUMEL, device=rays.device, dtype=torch.float32))
    ws = torch.empty(_lib.lib().nerfmi_nerf_backward_workspace_floats(n * p), device=rays.device,
                     dtype=torch.float32)
    if getattr(saved, "_nerfmi_math", None) not in (None, "f32" if fast is None else "bf16x3"):
        raise ValueError("nerf_backward_rays: the saved image was written by the %s forward; its backward must use the same math "
                         "(the fp32 and split-bf16 paths order the image differently)" % saved._nerfmi_math)
    if fast is not None:
        check(_lib.lib().nerfmi_nerf_backward_rays_fast(ptr(packed), ptr(fast), n, p, ptr(saved), ptr(grad_out),
                                                        _ptr_array(grads), ptr(ws), _stream(rays)),
              "nerf_backward_rays_fast")
    else:
        check(_lib.lib().nerfmi_nerf_backward_rays(ptr(packed), ptr(rays), ptr(z), n, p, ptr(saved), ptr(grad_out),
                                                   _ptr_array(grads), ptr(ws), _stream(rays)), "nerf_backward_rays")
    return grads


# --------------------------------------------------------------------------- a7
SIREN_PARAM_ORDER = ([f"network.{i}.layer.{k}" for i in range(8) for k in ("weight", "bias")]
                     + ["final_layer.weight", "final_layer.bias", "color_layer_sine.layer.weight",
                        "color_layer_sine.layer.bias", "color_layer_linear.0.weight", "color_layer_linear.0.bias"])
SIREN_PARAM_SHAPES = ([(256, 3), (256,)] + [(256, 256), (256,)] * 7
                      + [(1, 256), (1,), (256, 259), (256,), (3, 256), (3,)])


def siren_pack(params, out=None):
    ps = [_req(t.detach(), n, s) for n, s, t in zip(SIREN_PARAM_ORDER, SIREN_PARAM_SHAPES, params)]
    n = _lib.lib().nerfmi_siren_packed_floats()
    if out is None:
        out = torch.empty(n, device=ps[0].device, dtype=torch.float32)
    check(_lib.lib().nerfmi_siren_pack(_ptr_array(ps), ptr(out), _stream(out)), "siren_pack")
    return out


def siren_forward_points(packed, points, dirs, freq, phase, points_per_cond, sigma_only=False):
    points = _req(points, "points", (None, 3))
    n = points.shape[0]
    dirs = _req(dirs, "ray_directions", (n, 3)) if dirs is not None else None
    freq = _req(freq, "frequencies", (None, 2304))
    phase = _req(phase, "phase_shifts", (freq.shape[0], 2304))
    if freq.shape[0] * points_per_cond < n:
        raise ValueError("frequencies has too few rows for the points")
    out = torch.empty((n, 1 if sigma_only else 4), device=points.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_siren_forward_points(ptr(packed), ptr(points), ptr(dirs), ptr(freq), ptr(phase), n,
                                                 int(points_per_cond), int(bool(sigma_only)), ptr(out),
                                                 _stream(points)), "siren_forward_points")
    return out


def siren_forward_rays(packed, rays, z, freq, phase, rays_per_cond, sigma_only=False):
    rays = _req(rays, "rays", (None, 8))
    z = _req(z, "z", (rays.shape[0], None))
    n, p = z.shape
    freq = _req(freq, "frequencies", (None, 2304))
    phase = _req(phase, "phase_shifts", (freq.shape[0], 2304))
    if freq.shape[0] * rays_per_cond < n:
        raise ValueError("frequencies has too few rows for the rays")
    out = torch.empty((n * p, 1 if sigma_only else 4), device=rays.device, dtype=torch.float32)
    with _Span("siren_forward_rays", p):
        check(_lib.lib().nerfmi_siren_forward_rays(ptr(packed), ptr(rays), ptr(z), ptr(freq), ptr(phase), n, p,
                                                   int(rays_per_cond), int(bool(sigma_only)), ptr(out), _stream(rays)),
              "siren_forward_rays")
    return out


SIREN_PARAM_SIZES = [int(torch.Size(s).numel()) for s in SIREN_PARAM_SHAPES]
SIREN_PARAM_NUMEL = sum(SIREN_PARAM_SIZES)          # 529 156


def siren_flat_views(flat):
    """The 22 parameter-shaped views of one contiguous buffer (SIREN_PARAM_ORDER), cf. flat_views."""
    out, off = [], 0
    for shape, n in zip(SIREN_PARAM_SHAPES, SIREN_PARAM_SIZES):
        out.append(flat[off:off + n].view(shape))
        off += n
    return out


def _siren_cond(freq, phase, n_groups, per_cond):
    freq = _req(freq, "frequencies", (None, 2304))
    phase = _req(phase, "phase_shifts", (freq.shape[0], 2304))
    if per_cond < 1 or freq.shape[0] * per_cond < n_groups:
        raise ValueError("frequencies has too few rows")
    return freq, phase


def siren_forward_rays_train(packed, rays, z, freq, phase, rays_per_cond, fast=None):
    """Training forward of the FiLM-SIREN field behind the ray sampler -> (out (n*p,4), saved).
    fast: the split-bf16 image (siren_pack_fast) to run it on the bf16 matrix cores (opt-in math)."""
    rays = _req(rays, "rays", (None, 8))
    z = _req(z, "z", (rays.shape[0], None))
    n, p = z.shape
    freq, phase = _siren_cond(freq, phase, n, int(rays_per_cond))
    packed = _req(packed, "packed", (_lib.lib().nerfmi_siren_packed_floats(),))
    out = torch.empty((n * p, 4), device=rays.device, dtype=torch.float32)
    saved = torch.empty(_lib.lib().nerfmi_siren_saved_floats(n * p), device=rays.device, dtype=torch.float32)
    with _Span("siren_forward_rays_train", p):
        if fast is not None:
            check(_lib.lib().nerfmi_siren_forward_rays_train_fast(ptr(packed), ptr(fast), ptr(rays), ptr(z), ptr(freq), ptr(phase),
                                                                  n, p, int(rays_per_cond), ptr(out), ptr(saved), _stream(rays)),
                  "siren_forward_rays_train_fast")
        else:
            check(_lib.lib().nerfmi_siren_forward_rays_train(ptr(packed), ptr(rays), ptr(z), ptr(freq), ptr(phase), n, p,
                                                             int(rays_per_cond), ptr(out), ptr(saved), _stream(rays)),
                  "siren_forward_rays_train")
    saved._nerfmi_math = "f32" if fast is None else "bf16x3"   # the two paths keep their images in different element orders
    return out, saved


def siren_forward_points_train(packed, points, dirs, freq, phase, points_per_cond):
    points = _req(points, "points", (None, 3))
    n = points.shape[0]
    dirs = _req(dirs, "ray_directions", (n, 3))
    freq, phase = _siren_cond(freq, phase, n, int(points_per_cond))
    packed = _req(packed, "packed", (_lib.lib().nerfmi_siren_packed_floats(),))
    out = torch.empty((n, 4), device=points.device, dtype=torch.float32)
    saved = torch.empty(_lib.lib().nerfmi_siren_saved_floats(n), device=points.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_siren_forward_points_train(ptr(packed), ptr(points), ptr(dirs), ptr(freq), ptr(phase), n,
                                                       int(points_per_cond), ptr(out), ptr(saved), _stream(points)),
          "siren_forward_points_train")
    saved._nerfmi_math = "f32"
    return out, saved


def siren_backward(packed, saved, grad_out, freq, points_per_cond, grads=None, cond_grads=False, fast=None):
    """-> list of 22 gradient tensors (SIREN_PARAM_ORDER), written not accumulated.
    cond_grads=True (the launch must share ONE conditioning row): -> (grads, d frequencies (1, 2304), d phase_shifts (1, 2304))
    through nerfmi_siren_backward_cond.  fast: the split-bf16 image (siren_pack_fast): dX chain and the 256 x 256 dW tasks on
    the bf16 matrix cores (opt-in math)."""
    grad_out = _req(grad_out, "grad_out", (None, 4))
    n = grad_out.shape[0]
    freq = _req(freq, "frequencies", (None, 2304))
    if points_per_cond < 1 or freq.shape[0] * points_per_cond < n:
        raise ValueError("frequencies has too few rows")
    packed = _req(packed, "packed", (_lib.lib().nerfmi_siren_packed_floats(),))
    saved = _req(saved, "saved", (_lib.lib().nerfmi_siren_saved_floats(n),))
    if grads is None:
        grads = siren_flat_views(torch.empty(SIREN_PARAM_NUMEL, device=grad_out.device, dtype=torch.float32))
    ws = torch.empty(_lib.lib().nerfmi_siren_backward_workspace_floats(n), device=grad_out.device, dtype=torch.float32)
    if cond_grads and points_per_cond < n:
        raise ValueError("siren_backward(cond_grads=True) needs a launch that shares one conditioning row")
    if getattr(saved, "_nerfmi_math", None) not in (None, "f32" if fast is None else "bf16x3"):
        raise ValueError("siren_backward: the saved image was written by the %s forward; its backward must use the same math "
                         "(the fp32 and split-bf16 paths order the image differently)" % saved._nerfmi_math)
    if fast is not None:
        d_cond = torch.empty((2, 2304), device=grad_out.device, dtype=torch.float32) if cond_grads else None
        check(_lib.lib().nerfmi_siren_backward_fast(ptr(packed), ptr(fast), ptr(saved), ptr(grad_out), ptr(freq), n,
                                                    int(points_per_cond), _ptr_array(grads),
                                                    ptr(d_cond[0]) if cond_grads else None, ptr(d_cond[1]) if cond_grads else None,
                                                    ptr(ws), _stream(grad_out)), "siren_backward_fast")
        return (grads, d_cond[0:1], d_cond[1:2]) if cond_grads else grads
    if cond_grads:
        d_cond = torch.empty((2, 2304), device=grad_out.device, dtype=torch.float32)
        check(_lib.lib().nerfmi_siren_backward_cond(ptr(packed), ptr(saved), ptr(grad_out), ptr(freq), n, _ptr_array(grads),
                                                    ptr(d_cond[0]), ptr(d_cond[1]), ptr(ws), _stream(grad_out)),
              "siren_backward_cond")
        return grads, d_cond[0:1], d_cond[1:2]
    check(_lib.lib().nerfmi_siren_backward(ptr(packed), ptr(saved), ptr(grad_out), ptr(freq), n, int(points_per_cond),
                                           _ptr_array(grads), ptr(ws), _stream(grad_out)), "siren_backward")
    return grads


def siren_pack_fast(packed):
    fast = torch.empty(_lib.lib().nerfmi_siren_fast_bytes(), device=packed.device, dtype=torch.uint8)
    check(_lib.lib().nerfmi_siren_pack_fast(ptr(packed), ptr(fast), _stream(packed)), "siren_pack_fast")
    return fast


def siren_forward_rays_fast(packed, fast, rays, z, freq, phase, rays_per_cond, sigma_only=False):
    rays = _req(rays, "rays", (None, 8))
    z = _req(z, "z", (rays.shape[0], None))
    n, p = z.shape
    freq = _req(freq, "frequencies", (None, 2304))
    phase = _req(phase, "phase_shifts", (freq.shape[0], 2304))
    if freq.shape[0] * rays_per_cond < n:
        raise ValueError("frequencies has too few rows for the rays")
    out = torch.empty((n * p, 1 if sigma_only else 4), device=rays.device, dtype=torch.float32)
    with _Span("siren_forward_rays_fast", p):
        check(_lib.lib().nerfmi_siren_forward_rays_fast(ptr(packed), ptr(fast), ptr(rays), ptr(z), ptr(freq), ptr(phase), n, p,
                                                        int(rays_per_cond), int(bool(sigma_only)), ptr(out), _stream(rays)),
              "siren_forward_rays_fast")
    return out


# --------------------------------------------------------------------------- a8
def composite(field, z, rays, noise=None, noise_std=0.0, white_back=False, sigma_only=False, want_weights=True,
              philox=None):
    """philox = (seed, offset, segment): draw the density noise in the kernel (segment 1 coarse / 3 fine) instead of reading
    `noise`; composite_backward must be given the same triple."""
    rays = _req(rays, "rays", (None, 8))
    n = rays.shape[0]
    z = _req(z, "z", (n, None))
    p = z.shape[1]
    field = _req(field.reshape(n, p) if sigma_only else field.reshape(n, p, 4), "field")
    noise = _req(noise, "noise", (n, p)) if (noise is not None and noise_std != 0) else None
    dev = rays.device
    weights = torch.empty((n, p), device=dev, dtype=torch.float32) if want_weights else None
    opacity = torch.empty((n,), device=dev, dtype=torch.float32)
    rgb = depth = None
    if not sigma_only:
        rgb = torch.empty((n, 3), device=dev, dtype=torch.float32)
        depth = torch.empty((n,), device=dev, dtype=torch.float32)
    if noise is None and philox is not None and noise_std != 0:
        check(_lib.lib().nerfmi_composite_philox(ptr(field), int(bool(sigma_only)), ptr(z), ptr(rays), _u64(philox[0]),
                                                 _u64(philox[1]), int(philox[2]), float(noise_std), n, p,
                                                 int(bool(white_back)), ptr(weights), ptr(rgb), ptr(depth), ptr(opacity),
                                                 _stream(rays)), "composite_philox")
    else:
        check(_lib.lib().nerfmi_composite(ptr(field), int(bool(sigma_only)), ptr(z), ptr(rays), ptr(noise),
                                          float(noise_std), n, p, int(bool(white_back)), ptr(weights), ptr(rgb),
                                          ptr(depth), ptr(opacity), _stream(rays)), "composite")
    return weights, rgb, depth, opacity


def composite_backward(field, z, rays, noise, noise_std, white_back, g_rgb, g_depth, g_opacity, philox=None):
    rays = _req(rays, "rays", (None, 8))
    n = rays.shape[0]
    z = _req(z, "z", (n, None))
    p = z.shape[1]
    field = _req(field.reshape(n, p, 4), "field")
    noise = _req(noise, "noise", (n, p)) if (noise is not None and noise_std != 0) else None
    g_rgb = _req(g_rgb, "g_rgb", (n, 3)) if g_rgb is not None else None          # None = zero (NULL in the C ABI)
    g_depth = _req(g_depth, "g_depth", (n,)) if g_depth is not None else None
    g_opacity = _req(g_opacity, "g_opacity", (n,)) if g_opacity is not None else None
    grad_field = torch.empty((n * p, 4), device=rays.device, dtype=torch.float32)
    if noise is None and philox is not None and noise_std != 0:
        check(_lib.lib().nerfmi_composite_backward_philox(ptr(field), ptr(z), ptr(rays), _u64(philox[0]), _u64(philox[1]),
                                                          int(philox[2]), float(noise_std), n, p, int(bool(white_back)),
                                                          ptr(g_rgb), ptr(g_depth), ptr(g_opacity), ptr(grad_field),
                                                          _stream(rays)), "composite_backward_philox")
    else:
        check(_lib.lib().nerfmi_composite_backward(ptr(field), ptr(z), ptr(rays), ptr(noise), float(noise_std), n, p,
                                                   int(bool(white_back)), ptr(g_rgb), ptr(g_depth), ptr(g_opacity),
                                                   ptr(grad_field), _stream(rays)), "composite_backward")
    return grad_field


# --------------------------------------------------------------------------- a3 / a4
def sample_pdf(bins, weights, n_importance, det=False, u=None, return_aux=False):
    """sample_pdf(bins, weights, N_importance, det) of models/rendering.py:22-67.
    u: the torch.rand(N, N_importance) draw when not det (required then)."""
    weights = _req(weights, "weights", (None, None))
    n, nw = weights.shape
    bins = _req(bins, "bins", (n, nw + 1))
    if not det:
        if u is None:
            u = torch.rand((n, n_importance), device=bins.device, dtype=torch.float32)
        u = _req(u, "u", (n, n_importance))
    else:
        u = None
    dev = bins.device
    samples = torch.empty((n, n_importance), device=dev, dtype=torch.float32)
    cdf = torch.empty((n, nw + 1), device=dev, dtype=torch.float32) if return_aux else None
    inds = torch.empty((n, n_importance), device=dev, dtype=torch.int64) if return_aux else None
    check(_lib.lib().nerfmi_sample_pdf(ptr(bins), ptr(weights), ptr(u), n, nw, n_importance, ptr(cdf), ptr(inds),
                                       ptr(samples), _stream(bins)), "sample_pdf")
    return (samples, cdf, inds) if return_aux else samples


def search_lerp(bins, cdf, u):
    cdf = _req(cdf, "cdf", (None, None))
    n, nb = cdf.shape
    bins = _req(bins, "bins", (n, nb))
    u = _req(u, "u", (n, None))
    f = u.shape[1]
    samples = torch.empty((n, f), device=cdf.device, dtype=torch.float32)
    inds = torch.empty((n, f), device=cdf.device, dtype=torch.int64)
    check(_lib.lib().nerfmi_search_lerp(ptr(bins), ptr(cdf), ptr(u), n, nb - 1, f, ptr(inds), ptr(samples),
                                        _stream(cdf)), "search_lerp")
    return inds, samples


def searchsorted(a, v, out=None, side="left"):
    """torchsearchsorted.searchsorted(a, v, out=None, side='left')
    (torchsearchsorted/src/torchsearchsorted/searchsorted.py:20-53)."""
    assert len(a.shape) == 2, "input `a` must be 2-D."
    assert len(v.shape) == 2, "input `v` mus(t) be 2-D."
    assert (a.shape[0] == v.shape[0] or a.shape[0] == 1 or v.shape[0] == 1), \
        "`a` and `v` must have the same number of rows or one of them must have only 1 row"
    assert a.device == v.device, "`a` and `v` must be on the same device"
    if side not in ("left", "right"):
        raise ValueError("side must be 'left' or 'right'")
    a = _req(a, "a")
    v = _req(v, "v")
    nrow = max(a.shape[0], v.shape[0])
    if out is None:
        out = torch.empty((nrow, v.shape[1]), device=v.device, dtype=torch.long)
    else:
        assert out.device == v.device and out.dtype == torch.long and tuple(out.shape) == (nrow, v.shape[1])
        assert out.is_contiguous()
    check(_lib.lib().nerfmi_searchsorted(ptr(a), ptr(v), a.shape[0], v.shape[0], a.shape[1], v.shape[1],
                                         int(side == "left"), ptr(out), _stream(v)), "searchsorted")
    return out


def merge_sorted(za, zb):
    za = _req(za, "za", (None, None))
    zb = _req(zb, "zb", (za.shape[0], None))
    n = za.shape[0]
    out = torch.empty((n, za.shape[1] + zb.shape[1]), device=za.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_merge_sorted(ptr(za), ptr(zb), n, za.shape[1], zb.shape[1], ptr(out), _stream(za)),
          "merge_sorted")
    return out


def importance_resample(z_coarse, weights_coarse, n_importance, u=None, want_new=False, philox=None):
    """u = None and philox = None: deterministic u = linspace(0, 1, F) (rendering.py:44); philox = (seed, offset): draw u in
    the kernel (segment 2 of that stream)."""
    z_coarse = _req(z_coarse, "z_coarse", (None, None))
    n, s = z_coarse.shape
    weights_coarse = _req(weights_coarse, "weights_coarse", (n, s))
    u = _req(u, "u", (n, n_importance))
    z_fine = torch.empty((n, s + n_importance), device=z_coarse.device, dtype=torch.float32)
    z_new = torch.empty((n, n_importance), device=z_coarse.device, dtype=torch.float32) if want_new else None
    if u is None and philox is not None:
        check(_lib.lib().nerfmi_importance_resample_philox(ptr(z_coarse), ptr(weights_coarse), _u64(philox[0]), _u64(philox[1]),
                                                           n, s, n_importance, ptr(z_new), ptr(z_fine), _stream(z_coarse)),
              "importance_resample_philox")
    else:
        check(_lib.lib().nerfmi_importance_resample(ptr(z_coarse), ptr(weights_coarse), ptr(u), n, s, n_importance,
                                                    ptr(z_new), ptr(z_fine), _stream(z_coarse)), "importance_resample")
    return (z_fine, z_new) if want_new else z_fine


# --------------------------------------------------------------------------- f2: loss + optimizer
def mse_loss(rgb_coarse, rgb_fine, targets, grad_out=1.0, want_grads=True):
    """losses.py:10-20 + its autograd + metrics.py psnr in one launch.
    Returns (out4 = [loss, mse_coarse, mse_fine, psnr], grad_coarse or None, grad_fine or None)."""
    targets = _req(targets, "targets")
    n = targets.numel()
    rc = _req(rgb_coarse, "rgb_coarse", tuple(targets.shape)) if rgb_coarse is not None else None
    rf = _req(rgb_fine, "rgb_fine", tuple(targets.shape)) if rgb_fine is not None else None
    if rc is None and rf is None:
        raise ValueError("mse_loss needs rgb_coarse and/or rgb_fine")
    out4 = torch.empty(4, device=targets.device, dtype=torch.float32)
    gc = torch.empty_like(rc) if (want_grads and rc is not None) else None
    gf = torch.empty_like(rf) if (want_grads and rf is not None) else None
    check(_lib.lib().nerfmi_mse_loss(ptr(rc), ptr(rf), ptr(targets), n, float(grad_out), ptr(out4), ptr(gc), ptr(gf),
                                     _stream(targets)), "mse_loss")
    return out4, gc, gf


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
              grad_scale=1.0):
    """torch.optim.Adam's update on flat fp32 buffers, in place (one launch)."""
    n = param.numel()
    for name, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _req(t, name)
        if t.numel() != n or not t.is_contiguous():
            raise ValueError(f"adam_step: {name} must be a contiguous buffer of {n} floats")
    check(_lib.lib().nerfmi_adam_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), n, float(lr),
                                      float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                                      float(grad_scale), _stream(param)), "adam_step")


# --------------------------------------------------------------------------- f1: ray generation
def ray_directions(H, W, focal, device):
    """datasets/ray_utils.py:5-24 get_ray_directions -> (H, W, 3) on `device`."""
    out = torch.empty((int(H), int(W), 3), device=device, dtype=torch.float32)
    check(_lib.lib().nerfmi_ray_directions(int(H), int(W), float(focal), ptr(out), _stream(out)), "ray_directions")
    return out


def get_rays(directions, c2w):
    """datasets/ray_utils.py:27-50 get_rays(directions (...,3), c2w (3,4)) -> rays_o (n,3), rays_d (n,3)."""
    directions = _req(directions, "directions")
    c2w = _req(c2w, "c2w", (3, 4))
    if directions.shape[-1] != 3:
        raise ValueError("directions must end in a dimension of 3")
    n = directions.numel() // 3
    o = torch.empty((n, 3), device=directions.device, dtype=torch.float32)
    d = torch.empty_like(o)
    check(_lib.lib().nerfmi_get_rays(ptr(directions), ptr(c2w), n, ptr(o), ptr(d), _stream(o)), "get_rays")
    return o, d


def get_ndc_rays(H, W, focal, near, rays_o, rays_d):
    """datasets/ray_utils.py:53-93 (near: float)."""
    rays_o = _req(rays_o, "rays_o", (None, 3))
    rays_d = _req(rays_d, "rays_d", (rays_o.shape[0], 3))
    o, d = torch.empty_like(rays_o), torch.empty_like(rays_d)
    check(_lib.lib().nerfmi_ndc_rays(int(H), int(W), float(focal), float(near), ptr(rays_o), ptr(rays_d), rays_o.shape[0],
                                     ptr(o), ptr(d), _stream(o)), "ndc_rays")
    return o, d


def generate_rays(c2w, H, W, focal, pixel_index=None, ndc=False, near=2.0, far=6.0):
    """(n_rays, 8) [o, d, near, far] straight from (c2w (n_images,3,4), focal, pixel index): the fused form of
    get_ray_directions + get_rays (+ get_ndc_rays) + the packing of blender.py:60-69 / llff.py:234-250."""
    c2w = _req(c2w.reshape(-1, 3, 4), "c2w")
    n_img = c2w.shape[0]
    if pixel_index is None:
        n = n_img * int(H) * int(W)
    else:
        pixel_index = _req(pixel_index, "pixel_index", (None,), dtype=torch.int64)
        n = pixel_index.shape[0]
    rays = torch.empty((n, 8), device=c2w.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_generate_rays(ptr(c2w), n_img, int(H), int(W), float(focal), ptr(pixel_index), n,
                                          int(bool(ndc)), float(near), float(far), ptr(rays), _stream(rays)),
          "generate_rays")
    return rays
