"""ctypes binding of the C ABI in include/nerfmi.h (libnerfmi.so, gfx950).

There is NO fallback: if the HIP library is missing or a call fails this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NERFMI_LIB: an experiment build of the same ABI (tools/exp_*.py, nerf_siren_amd/build.py NERFMI_LIB_OUT)
LIB_PATH = os.environ.get("NERFMI_LIB") or os.path.join(_HERE, "lib", "libnerfmi.so")

_f = C.c_void_p          # device pointers travel as integers
_i = C.c_int
_i64 = C.c_int64
_fl = C.c_float

SIGNATURES = {
    "nerfmi_version": (C.c_int, []),
    "nerfmi_last_error": (C.c_char_p, []),
    "nerfmi_sample_stratified": (_i, [_f, _f, _i, _i, _i, _fl, _f, _f]),
    "nerfmi_sample_stratified_philox": (_i, [_f, C.c_uint64, C.c_uint64, _i, _i, _i, _fl, _f, _f]),
    "nerfmi_composite_philox": (_i, [_f, _i, _f, _f, C.c_uint64, C.c_uint64, _i, _fl, _i, _i, _i, _f, _f, _f, _f, _f]),
    "nerfmi_composite_backward_philox": (_i, [_f, _f, _f, C.c_uint64, C.c_uint64, _i, _fl, _i, _i, _i, _f, _f, _f, _f, _f]),
    "nerfmi_importance_resample_philox": (_i, [_f, _f, C.c_uint64, C.c_uint64, _i, _i, _i, _f, _f, _f]),
    "nerfmi_render_draws": (_i, [C.c_uint64, C.c_uint64, _i64, _f, _i64, _f, _i64, _f, _i64, _f, _f]),
    "nerfmi_embed": (_i, [_f, _i64, _i, _f, _f]),
    "nerfmi_nerf_packed_floats": (C.c_size_t, []),
    "nerfmi_nerf_pack": (_i, [C.POINTER(C.c_void_p), _f, _f]),
    "nerfmi_nerf_saved_floats": (C.c_size_t, [_i64]),
    "nerfmi_nerf_forward_rays": (_i, [_f, _f, _f, _i, _i, _i, _f, _f, _f]),
    "nerfmi_nerf_fast_bytes": (C.c_size_t, []),
    "nerfmi_nerf_pack_fast": (_i, [_f, _f, _f]),
    "nerfmi_nerf_backward_rays_fast": (_i, [_f, _f, _i, _i, _f, _f, C.POINTER(C.c_void_p), _f, _f]),
    "nerfmi_nerf_forward_rays_fast": (_i, [_f, _f, _f, _f, _i, _i, _i, _f, _f, _f]),
    "nerfmi_nerf_forward_embedded": (_i, [_f, _f, _i64, _i, _f, _f]),
    "nerfmi_nerf_forward_embedded_train": (_i, [_f, _f, _i64, _f, _f, _f]),
    "nerfmi_nerf_backward_workspace_floats": (C.c_size_t, [_i64]),
    "nerfmi_nerf_backward_rays": (_i, [_f, _f, _f, _i, _i, _f, _f, C.POINTER(C.c_void_p), _f, _f]),
    "nerfmi_siren_packed_floats": (C.c_size_t, []),
    "nerfmi_siren_fast_bytes": (C.c_size_t, []),
    "nerfmi_siren_pack_fast": (_i, [_f, _f, _f]),
    "nerfmi_siren_forward_rays_fast": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i64, _i, _f, _f]),
    "nerfmi_siren_pack": (_i, [C.POINTER(C.c_void_p), _f, _f]),
    "nerfmi_siren_forward_points": (_i, [_f, _f, _f, _f, _f, _i64, _i64, _i, _f, _f]),
    "nerfmi_siren_forward_rays": (_i, [_f, _f, _f, _f, _f, _i, _i, _i64, _i, _f, _f]),
    "nerfmi_siren_saved_floats": (C.c_size_t, [_i64]),
    "nerfmi_siren_backward_workspace_floats": (C.c_size_t, [_i64]),
    "nerfmi_siren_forward_rays_train": (_i, [_f, _f, _f, _f, _f, _i, _i, _i64, _f, _f, _f]),
    "nerfmi_siren_forward_points_train": (_i, [_f, _f, _f, _f, _f, _i64, _i64, _f, _f, _f]),
    "nerfmi_siren_backward": (_i, [_f, _f, _f, _f, _i64, _i64, C.POINTER(C.c_void_p), _f, _f]),
    "nerfmi_render_rays_workspace_floats": (C.c_size_t, [_i, _i, _i, _i]),
    "nerfmi_render_rays_fused": (_i, [_i, _f, _f, _f, _f, _f, _i, _i, _i, _i, _fl, _fl, _i, _i, C.c_uint64, C.c_uint64, _f,
                                      _f, _f, _f, _f, _f, _f, _f]),
    "nerfmi_profile_start": (_i, []),
    "nerfmi_profile_stop": (_i, []),
    "nerfmi_profile_report": (_i64, [C.c_char_p, C.c_size_t]),
    "nerfmi_siren_forward_rays_train_fast": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i64, _f, _f, _f]),
    "nerfmi_siren_backward_fast": (_i, [_f, _f, _f, _f, _f, _i64, _i64, C.POINTER(C.c_void_p), _f, _f, _f, _f]),
    "nerfmi_siren_backward_cond": (_i, [_f, _f, _f, _f, _i64, C.POINTER(C.c_void_p), _f, _f, _f, _f]),
    "nerfmi_eg3d_pack_planes": (_i, [_f, _i, _i, _i, _i, _f, _f]),
    "nerfmi_eg3d_decoder_floats": (C.c_size_t, []),
    "nerfmi_eg3d_pack_decoder": (_i, [_f, _f, _f, _f, _fl, _f, _f]),
    "nerfmi_eg3d_sample_planes": (_i, [_f, _i, _i, _i, _f, _i64, _fl, _f, _f]),
    "nerfmi_eg3d_run_model": (_i, [_f, _i, _i, _i, _f, _f, _i64, _fl, _f, _f, _f]),
    "nerfmi_eg3d_run_model_rays": (_i, [_f, _i, _i, _i, _f, _f, _f, _f, _i64, _i, _fl, _f, _f, _f]),
    "nerfmi_eg3d_sample_stratified": (_i, [_f, _f, _fl, _fl, _f, _i64, _i, _i, _f, _f]),
    "nerfmi_eg3d_sample_stratified_philox": (_i, [_f, _f, _fl, _fl, C.c_uint64, C.c_uint64, _i64, _i, _i, _f, _f]),
    "nerfmi_eg3d_minmax": (_i, [_f, _i64, _f, _f]),
    "nerfmi_eg3d_march": (_i, [_f, _f, _f, _f, _i64, _i, _i, _f, _f, _f, _f, _f]),
    "nerfmi_eg3d_sample_importance": (_i, [_f, _f, _f, _i64, _i, _i, _f, _f]),
    "nerfmi_eg3d_sample_importance_philox": (_i, [_f, _f, C.c_uint64, C.c_uint64, _i64, _i, _i, _f, _f]),
    "nerfmi_eg3d_unify": (_i, [_f, _f, _f, _f, _f, _f, _i64, _i, _i, _f, _f, _f, _f, _f]),
    "nerfmi_eg3d_march_backward": (_i, [_f, _f, _f, _f, _f, _f, _f, _i64, _i, _i, _i, _f, _f, _f]),
    "nerfmi_eg3d_unify_backward": (_i, [_f, _f, _f, _i64, _i, _i, _f, _f, _f, _f, _f]),
    "nerfmi_eg3d_backward_aux_floats": (C.c_size_t, [_i64]),
    "nerfmi_eg3d_run_model_rays_backward": (_i, [_f, _i, _i, _i, _f, _f, _f, _f, _i64, _i, _fl, _f, _f, _f, _f, _f]),
    "nerfmi_eg3d_wgrad_partial_floats": (C.c_size_t, []),
    "nerfmi_eg3d_decoder_wgrad": (_i, [_f, _i64, _fl, _i, _f, _f, _f, _f, _f, _f]),
    "nerfmi_eg3d_unpack_planes": (_i, [_f, _i, _i, _i, _i, _f, _f]),
    "nerfmi_eg3d_ray_sampler": (_i, [_f, _f, _i, _i, _f, _f, _f]),
    "nerfmi_eg3d_ray_limits_box": (_i, [_f, _f, _i64, _fl, _f, _f, _f]),
    "nerfmi_composite": (_i, [_f, _i, _f, _f, _f, _fl, _i, _i, _i, _f, _f, _f, _f, _f]),
    "nerfmi_composite_backward": (_i, [_f, _f, _f, _f, _fl, _i, _i, _i, _f, _f, _f, _f, _f]),
    "nerfmi_sample_pdf": (_i, [_f, _f, _f, _i, _i, _i, _f, _f, _f, _f]),
    "nerfmi_search_lerp": (_i, [_f, _f, _f, _i, _i, _i, _f, _f, _f]),
    "nerfmi_searchsorted": (_i, [_f, _f, _i, _i, _i, _i, _i, _f, _f]),
    "nerfmi_merge_sorted": (_i, [_f, _f, _i, _i, _i, _f, _f]),
    "nerfmi_ray_directions": (_i, [_i, _i, C.c_double, _f, _f]),
    "nerfmi_get_rays": (_i, [_f, _f, _i64, _f, _f, _f]),
    "nerfmi_ndc_rays": (_i, [_i, _i, C.c_double, C.c_double, _f, _f, _i64, _f, _f, _f]),
    "nerfmi_generate_rays": (_i, [_f, _i, _i, _i, C.c_double, _f, _i64, _i, C.c_double, C.c_double, _f, _f]),
    "nerfmi_create_samples": (_i, [_i, C.c_double, C.c_double, C.c_double, C.c_double, _f, _f]),
    "nerfmi_mse_loss": (_i, [_f, _f, _f, _i64, _fl, _f, _f, _f, _f]),
    "nerfmi_adam_step": (_i, [_f, _f, _f, _f, _i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i64,
                             C.c_double, _f]),
    "nerfmi_importance_resample": (_i, [_f, _f, _f, _i, _i, _i, _f, _f, _f]),
}

_lib = None


class NerfmiError(RuntimeError):
    pass


def load_shared(path=None):
    """ctypes.CDLL of libnerfmi.so bound to the SAME HIP runtime PyTorch uses.

    PyTorch-ROCm ships its own libamdhip64.so / libhsa-runtime64.so and loads them only when torch.cuda initialises.  If
    libnerfmi.so is dlopen-ed BEFORE that, the loader resolves its `libamdhip64.so.7` dependency to /opt/rocm's copy, the
    process ends up with two HIP runtimes, and the second one to open the device finds none: the library's first launch
    fails with "no ROCm-capable device is detected" (seen in round 3: build() loaded the library, then smoke() ran in the
    same process).  Loading torch's copy first (by path, RTLD_GLOBAL) makes the later dependency lookup hit the already
    loaded SONAME whatever the order.  Without a bundled runtime (or without torch) the system copy is the only one."""
    try:
        import torch
        tl = os.path.join(os.path.dirname(torch.__file__), "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            p = os.path.join(tl, name)
            if os.path.exists(p):
                C.CDLL(p, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError):
        pass
    return C.CDLL(path or LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "nerf_siren_amd has no CPU/PyTorch fallback.")
        l = load_shared()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().nerfmi_last_error().decode("utf-8", "replace")
        raise NerfmiError(f"{what or 'nerfmi'} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
