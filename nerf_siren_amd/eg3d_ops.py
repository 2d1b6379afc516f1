"""Tensor-level wrappers over the EG3D part of the C ABI (include/nerfmi.h)."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr
from .ops import _req, _stream


def pack_planes(planes):
    """(N,3,32,H,W) NCHW -> channels-last (N*3,H,W,32) image used by every EG3D kernel."""
    planes = _req(planes, "planes", (None, 3, 32, None, None))
    n, _, c, h, w = planes.shape
    out = torch.empty((n * 3, h, w, c), device=planes.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_pack_planes(ptr(planes), n * 3, c, h, w, ptr(out), _stream(planes)), "eg3d_pack_planes")
    return out


def pack_decoder(w0, b0, w1, b1, lr_mul=1.0):
    w0, b0 = _req(w0.detach(), "net.0.weight", (64, 32)), _req(b0.detach(), "net.0.bias", (64,))
    w1, b1 = _req(w1.detach(), "net.2.weight", (4, 64)), _req(b1.detach(), "net.2.bias", (4,))
    out = torch.empty(_lib.lib().nerfmi_eg3d_decoder_floats(), device=w0.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_pack_decoder(ptr(w0), ptr(b0), ptr(w1), ptr(b1), float(lr_mul), ptr(out), _stream(w0)),
          "eg3d_pack_decoder")
    return out


def sample_planes(planes_hwc, n, coords, box_warp):
    coords = _req(coords, "coordinates", (n, None, 3))
    p = coords.shape[1]
    h, w = planes_hwc.shape[1], planes_hwc.shape[2]
    out = torch.empty((n, 3, p, 32), device=coords.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_sample_planes(ptr(planes_hwc), n, h, w, ptr(coords), p, float(box_warp), ptr(out),
                                               _stream(coords)), "eg3d_sample_planes")
    return out


def run_model(planes_hwc, n, dec, coords, box_warp):
    coords = _req(coords, "coordinates", (n, None, 3))
    p = coords.shape[1]
    h, w = planes_hwc.shape[1], planes_hwc.shape[2]
    rgb = torch.empty((n, p, 3), device=coords.device, dtype=torch.float32)
    sigma = torch.empty((n, p, 1), device=coords.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_run_model(ptr(planes_hwc), n, h, w, ptr(dec), ptr(coords), p, float(box_warp), ptr(rgb),
                                           ptr(sigma), _stream(coords)), "eg3d_run_model")
    return rgb, sigma


def run_model_rays(planes_hwc, n, dec, ray_o, ray_d, depths, box_warp):
    ray_o = _req(ray_o, "ray_origins", (n, None, 3))
    m = ray_o.shape[1]
    ray_d = _req(ray_d, "ray_directions", (n, m, 3))
    depths = _req(depths.reshape(n, m, -1), "depths")
    s = depths.shape[2]
    h, w = planes_hwc.shape[1], planes_hwc.shape[2]
    rgb = torch.empty((n, m, s, 3), device=ray_o.device, dtype=torch.float32)
    sigma = torch.empty((n, m, s, 1), device=ray_o.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_run_model_rays(ptr(planes_hwc), n, h, w, ptr(dec), ptr(ray_o), ptr(ray_d), ptr(depths),
                                                m, s, float(box_warp), ptr(rgb), ptr(sigma), _stream(ray_o)),
          "eg3d_run_model_rays")
    return rgb, sigma


def sample_stratified(n_rays, n_samples, rand, ray_start, ray_end, disparity=False, device=None):
    """rand: the injected rand_like draw (n_rays, n_samples), or a (seed, offset) Philox key (ops.next_draw_key): the draw is
    then made INSIDE the kernel (segment 0 of that stream) -- no aten distribution launch, no tensor of draws."""
    philox = rand if isinstance(rand, tuple) else None
    per_ray = isinstance(ray_start, torch.Tensor)
    if philox is None:
        rand = _req(rand.reshape(n_rays, n_samples), "rand")
        device = rand.device
    elif per_ray:
        device = ray_start.device
    out = torch.empty((n_rays, n_samples), device=device, dtype=torch.float32)
    st = _req(ray_start.reshape(n_rays), "ray_start") if per_ray else None
    en = _req(ray_end.reshape(n_rays), "ray_end") if per_ray else None
    a, b = (0.0, 0.0) if per_ray else (float(ray_start), float(ray_end))
    disp = 0 if per_ray else int(bool(disparity))
    if philox is None:
        check(_lib.lib().nerfmi_eg3d_sample_stratified(ptr(st), ptr(en), a, b, ptr(rand), n_rays, n_samples, disp, ptr(out),
                                                       _stream(out)), "eg3d_sample_stratified")
    else:
        check(_lib.lib().nerfmi_eg3d_sample_stratified_philox(ptr(st), ptr(en), a, b, int(philox[0]) & (2 ** 64 - 1),
                                                              int(philox[1]) & (2 ** 64 - 1), n_rays, n_samples, disp,
                                                              ptr(out), _stream(out)), "eg3d_sample_stratified_philox")
    return out


def minmax(x):
    x = _req(x.reshape(-1), "depths")
    out = torch.empty(2, device=x.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_minmax(ptr(x), x.numel(), ptr(out), _stream(x)), "eg3d_minmax")
    return out


def march(colors, densities, depths, white_back=False, mm=None):
    """colors (R,S,3), densities (R,S), depths (R,S) -> rgb (R,3), depth (R), weights (R,S-1), weight_sum (R)."""
    colors = _req(colors, "colors", (None, None, 3))
    r, s = colors.shape[0], colors.shape[1]
    densities = _req(densities.reshape(r, s), "densities")
    depths = _req(depths.reshape(r, s), "depths")
    if mm is None:
        mm = minmax(depths)
    dev = colors.device
    rgb = torch.empty((r, 3), device=dev, dtype=torch.float32)
    depth = torch.empty((r,), device=dev, dtype=torch.float32)
    w = torch.empty((r, s - 1), device=dev, dtype=torch.float32)
    ws = torch.empty((r,), device=dev, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_march(ptr(colors), ptr(densities), ptr(depths), ptr(mm), r, s, int(bool(white_back)),
                                       ptr(rgb), ptr(depth), ptr(w), ptr(ws), _stream(colors)), "eg3d_march")
    return rgb, depth, w, ws


def sample_importance(depths, weights, u, n_importance=None):
    """u: the injected torch.rand draw (R, F), or a (seed, offset) Philox key + n_importance: drawn inside the kernel
    (segment 2 of that stream)."""
    depths = _req(depths, "z_vals", (None, None))
    r, s = depths.shape
    weights = _req(weights.reshape(r, s - 1), "weights")
    if isinstance(u, tuple):
        f = int(n_importance)
        out = torch.empty((r, f), device=depths.device, dtype=torch.float32)
        check(_lib.lib().nerfmi_eg3d_sample_importance_philox(ptr(depths), ptr(weights), int(u[0]) & (2 ** 64 - 1),
                                                              int(u[1]) & (2 ** 64 - 1), r, s, f, ptr(out), _stream(depths)),
              "eg3d_sample_importance_philox")
        return out
    u = _req(u, "u", (r, None))
    f = u.shape[1]
    out = torch.empty((r, f), device=depths.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_sample_importance(ptr(depths), ptr(weights), ptr(u), r, s, f, ptr(out), _stream(depths)),
          "eg3d_sample_importance")
    return out


def unify(d1, c1, s1, d2, c2, s2, want_idx=False):
    d1 = _req(d1, "depths1", (None, None))
    r, n1 = d1.shape
    d2 = _req(d2.reshape(r, -1), "depths2")
    n2 = d2.shape[1]
    c1, s1 = _req(c1.reshape(r, n1, 3), "colors1"), _req(s1.reshape(r, n1), "densities1")
    c2, s2 = _req(c2.reshape(r, n2, 3), "colors2"), _req(s2.reshape(r, n2), "densities2")
    dev = d1.device
    d = torch.empty((r, n1 + n2), device=dev, dtype=torch.float32)
    c = torch.empty((r, n1 + n2, 3), device=dev, dtype=torch.float32)
    s = torch.empty((r, n1 + n2), device=dev, dtype=torch.float32)
    idx = torch.empty((r, n1 + n2), device=dev, dtype=torch.int32) if want_idx else None
    check(_lib.lib().nerfmi_eg3d_unify(ptr(d1), ptr(c1), ptr(s1), ptr(d2), ptr(c2), ptr(s2), r, n1, n2, ptr(d), ptr(c),
                                       ptr(s), ptr(idx), _stream(d1)), "eg3d_unify")
    return (d, c, s, idx) if want_idx else (d, c, s)


def march_backward(colors, densities, depths, mm, g_rgb, g_depth, g_wsum, white_back, into=None):
    """-> d_colors (R,S,3), d_densities (R,S); `into` = (d_colors, d_densities) to accumulate into."""
    r, s = colors.shape[0], colors.shape[1]
    dev = colors.device
    acc = into is not None
    dc, ds = into if acc else (torch.empty((r, s, 3), device=dev, dtype=torch.float32),
                               torch.empty((r, s), device=dev, dtype=torch.float32))
    g_rgb = _req(g_rgb.reshape(r, 3), "g_rgb") if g_rgb is not None else None
    g_depth = _req(g_depth.reshape(r), "g_depth") if g_depth is not None else None
    g_wsum = _req(g_wsum.reshape(r), "g_wsum") if g_wsum is not None else None
    check(_lib.lib().nerfmi_eg3d_march_backward(ptr(colors), ptr(densities), ptr(depths), ptr(mm), ptr(g_rgb),
                                                ptr(g_depth), ptr(g_wsum), r, s, int(bool(white_back)), int(acc),
                                                ptr(dc), ptr(ds), _stream(colors)), "eg3d_march_backward")
    return dc, ds


def unify_backward(idx, g_c, g_s, n1, n2):
    r = idx.shape[0]
    dev = idx.device
    dc1 = torch.empty((r, n1, 3), device=dev, dtype=torch.float32)
    ds1 = torch.empty((r, n1), device=dev, dtype=torch.float32)
    dc2 = torch.empty((r, n2, 3), device=dev, dtype=torch.float32)
    ds2 = torch.empty((r, n2), device=dev, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_unify_backward(ptr(idx), ptr(g_c), ptr(g_s), r, n1, n2, ptr(dc1), ptr(ds1), ptr(dc2),
                                                ptr(ds2), _stream(idx)), "eg3d_unify_backward")
    return dc1, ds1, dc2, ds2


def run_model_rays_backward(planes_hwc, n, dec, ray_o, ray_d, depths, box_warp, d_rgb, d_sigma, gplanes_hwc):
    m = ray_o.shape[1]
    depths = depths.reshape(n, m, -1)
    s = depths.shape[2]
    h, w = planes_hwc.shape[1], planes_hwc.shape[2]
    npts = n * m * s
    aux = torch.empty(_lib.lib().nerfmi_eg3d_backward_aux_floats(npts), device=ray_o.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_run_model_rays_backward(ptr(planes_hwc), n, h, w, ptr(dec), ptr(ray_o), ptr(ray_d),
                                                         ptr(depths), m, s, float(box_warp),
                                                         ptr(_req(d_rgb.reshape(npts, 3), "d_rgb")),
                                                         ptr(_req(d_sigma.reshape(npts), "d_sigma")), ptr(gplanes_hwc),
                                                         ptr(aux), _stream(ray_o)), "eg3d_run_model_rays_backward")
    return aux, npts


def decoder_wgrad(aux, npts, lr_mul, grads, accumulate):
    partial = torch.empty(_lib.lib().nerfmi_eg3d_wgrad_partial_floats(), device=aux.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_decoder_wgrad(ptr(aux), npts, float(lr_mul), int(bool(accumulate)), ptr(partial),
                                               ptr(grads[0]), ptr(grads[1]), ptr(grads[2]), ptr(grads[3]), _stream(aux)),
          "eg3d_decoder_wgrad")


def unpack_planes(g_hwc, n):
    _, h, w, c = g_hwc.shape
    out = torch.empty((n, 3, c, h, w), device=g_hwc.device, dtype=torch.float32)
    check(_lib.lib().nerfmi_eg3d_unpack_planes(ptr(g_hwc), n * 3, c, h, w, ptr(out), _stream(g_hwc)), "eg3d_unpack_planes")
    return out


def ray_sampler(cam2world, intrinsics, resolution):
    cam2world = _req(cam2world, "cam2world_matrix", (None, 4, 4))
    n = cam2world.shape[0]
    intrinsics = _req(intrinsics, "intrinsics", (n, 3, 3))
    o = torch.empty((n, resolution * resolution, 3), device=cam2world.device, dtype=torch.float32)
    d = torch.empty_like(o)
    check(_lib.lib().nerfmi_eg3d_ray_sampler(ptr(cam2world), ptr(intrinsics), n, int(resolution), ptr(o), ptr(d),
                                             _stream(cam2world)), "eg3d_ray_sampler")
    return o, d


def ray_limits_box(rays_o, rays_d, box_side_length):
    shp = rays_o.shape
    o = _req(rays_o.detach().reshape(-1, 3), "rays_o")
    d = _req(rays_d.detach().reshape(-1, 3), "rays_d")
    tmin = torch.empty(o.shape[0], device=o.device, dtype=torch.float32)
    tmax = torch.empty_like(tmin)
    check(_lib.lib().nerfmi_eg3d_ray_limits_box(ptr(o), ptr(d), o.shape[0], float(box_side_length), ptr(tmin), ptr(tmax),
                                                _stream(o)), "eg3d_ray_limits_box")
    return tmin.reshape(*shp[:-1], 1), tmax.reshape(*shp[:-1], 1)
