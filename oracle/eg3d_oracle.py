"""CPU oracle for the EG3D tri-plane importance renderer -- TEST INFRASTRUCTURE, NOT PRODUCT.

numpy restatement of
    volumetric_rendering/renderer.py:23-256      generate_planes, project_onto_planes, sample_from_planes,
                                                 ImportanceRenderer (forward, run_model, unify_samples,
                                                 sample_stratified, sample_importance, sample_pdf)
    volumetric_rendering/ray_marcher.py:25-57    MipRayMarcher2.run_forward
    volumetric_rendering/ray_sampler.py:24-63    RaySampler.forward
    volumetric_rendering/math_utils.py:46-118    get_ray_limits_box, linspace
    eg3d_training/triplane.py:144-167            OSGDecoder
    eg3d_training/networks_stylegan2.py:96-127   FullyConnectedLayer (linear + bias branch)
Parity status: PINNED by tests/golden/g9_eg3d*.npz (outputs of the reference itself,
tools/make_golden.py).  Same arithmetic conventions as oracle/nerf_oracle.py.
"""
from __future__ import annotations

import numpy as np

from .nerf_oracle import F32, F64, linspace01, searchsorted

T_FLOOR = F32(1e-10)


def linspace(start: float, end: float, n: int) -> np.ndarray:
    """torch.linspace(start, end, n) fp32 on CPU (symmetric fill, fma form)."""
    start, end = F32(start), F32(end)
    step = F32((F64(end) - F64(start)) / (n - 1))         # computed in fp64 then stored fp32 (ATen: scalar_t step)
    step = F32((end - start) / F32(n - 1))
    i = np.arange(n)
    lo = (F64(start) + F64(step) * i).astype(F32)
    hi = (F64(end) - F64(step) * (n - 1 - i)).astype(F32)
    return np.where(i < n // 2, lo, hi).astype(F32)


# ---------------------------------------------------------------------------- a12 stratified
def sample_stratified(n_rays_shape, ray_start, ray_end, S, rand, disparity=False):
    """renderer.py:172-195 with float ray_start/ray_end. rand (N,M,S,1) = the rand_like draw."""
    N, M = n_rays_shape
    rand = np.asarray(rand, F32)
    if disparity:
        d = np.broadcast_to(linspace01(S).reshape(1, 1, S, 1), (N, M, S, 1)).copy()
        delta = F32(1 / (S - 1))
        d = (d + (rand * delta).astype(F32)).astype(F32)
        a = (F32(1.0 / ray_start) * (F32(1) - d).astype(F32)).astype(F32)
        b = (F32(1.0 / ray_end) * d).astype(F32)
        return (F32(1) / (a + b).astype(F32)).astype(F32)
    d = np.broadcast_to(linspace(ray_start, ray_end, S).reshape(1, 1, S, 1), (N, M, S, 1)).copy()
    delta = F32((ray_end - ray_start) / (S - 1))
    return (d + (rand * delta).astype(F32)).astype(F32)


def sample_stratified_tensor(ray_start, ray_end, S, rand):
    """renderer.py:186-189 (per-ray start/end, math_utils.linspace). ray_start/end (N,M,1)."""
    steps = (np.arange(S, dtype=F32) / F32(S - 1)).astype(F32)
    span = (ray_end - ray_start).astype(F32)
    d = (ray_start[None] + (steps[:, None, None, None] * span[None]).astype(F32)).astype(F32)      # (S,N,M,1)
    d = np.transpose(d, (1, 2, 0, 3))
    delta = (span / F32(S - 1)).astype(F32)
    return (d + (np.asarray(rand, F32) * delta[..., None]).astype(F32)).astype(F32)


# ---------------------------------------------------------------------------- a9 tri-plane sampling
PLANE_SEL = ((0, 1), (0, 2), (2, 0))   # coordinates @ inv(plane_axes) [..., :2] (renderer.py:23-53)


def grid_sample_bilinear(plane, gx, gy):
    """F.grid_sample(bilinear, zeros, align_corners=False) for one (C,H,W) plane; gx,gy (P,) -> (P,C)."""
    C, H, W = plane.shape
    ix = (((gx + F32(1)) * F32(W)).astype(F32) - F32(1)).astype(F32) / F32(2)
    iy = (((gy + F32(1)) * F32(H)).astype(F32) - F32(1)).astype(F32) / F32(2)
    ix, iy = ix.astype(F32), iy.astype(F32)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    wx1 = (ix - x0).astype(F32)
    wx0 = (x1 - ix).astype(F32)
    wy1 = (iy - y0).astype(F32)
    wy0 = (y1 - iy).astype(F32)
    out = np.zeros((gx.shape[0], C), F32)

    def tap(xi, yi, w):
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        xc = np.clip(xi, 0, W - 1).astype(np.int64)
        yc = np.clip(yi, 0, H - 1).astype(np.int64)
        v = plane[:, yc, xc].T                                    # (P,C)
        return np.where(ok[:, None], (v * w[:, None]).astype(F32), F32(0))

    out = tap(x0, y0, (wx0 * wy0).astype(F32))
    out = (out + tap(x1, y0, (wx1 * wy0).astype(F32))).astype(F32)
    out = (out + tap(x0, y1, (wx0 * wy1).astype(F32))).astype(F32)
    out = (out + tap(x1, y1, (wx1 * wy1).astype(F32))).astype(F32)
    return out


def sample_from_planes(planes, coords, box_warp):
    """renderer.py:55-65. planes (N,3,C,H,W), coords (N,P,3) -> (N,3,P,C)."""
    N = planes.shape[0]
    c = (F32(2 / box_warp) * np.asarray(coords, F32)).astype(F32)
    out = np.zeros((N, 3, coords.shape[1], planes.shape[2]), F32)
    for n in range(N):
        for pl, (a, b) in enumerate(PLANE_SEL):
            out[n, pl] = grid_sample_bilinear(planes[n, pl], c[n, :, a], c[n, :, b])
    return out


# ---------------------------------------------------------------------------- a10 decoder
def softplus(x):
    x = np.asarray(x, F32)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20).astype(F64))).astype(F32)).astype(F32)


def osg_decoder(p, feats, lr_mul=1.0):
    """OSGDecoder.forward (triplane.py:155-167). feats (N,3,P,32); p: net.0.weight/bias, net.2.weight/bias."""
    x = (feats.astype(F32).sum(1) / F32(3)).astype(F32)          # mean over planes
    N, P, C = x.shape
    x = x.reshape(N * P, C)
    for k, fin in (("net.0", 32), ("net.2", 64)):
        w = (p[k + ".weight"] * F32(lr_mul / np.sqrt(fin))).astype(F32)
        b = (p[k + ".bias"] * F32(lr_mul)).astype(F32) if lr_mul != 1 else p[k + ".bias"]
        x = (x @ w.T + b).astype(F32)
        if k == "net.0":
            x = softplus(x)
    x = x.reshape(N, P, -1)
    sig = (F32(1) / (F32(1) + np.exp(-x[..., 1:].astype(F64)))).astype(F32)
    rgb = ((sig * F32(1 + 2 * 0.001)).astype(F32) - F32(0.001)).astype(F32)
    return rgb, x[..., 0:1]


# ---------------------------------------------------------------------------- a11 marcher
def mip_march(colors, densities, depths, white_back=False):
    """MipRayMarcher2.run_forward (ray_marcher.py:25-57). colors (N,M,S,3), densities/depths (N,M,S,1)."""
    colors, densities, depths = (np.asarray(a, F32) for a in (colors, densities, depths))
    deltas = (depths[:, :, 1:] - depths[:, :, :-1]).astype(F32)
    cmid = ((colors[:, :, :-1] + colors[:, :, 1:]).astype(F32) / F32(2)).astype(F32)
    dmid = ((densities[:, :, :-1] + densities[:, :, 1:]).astype(F32) / F32(2)).astype(F32)
    zmid = ((depths[:, :, :-1] + depths[:, :, 1:]).astype(F32) / F32(2)).astype(F32)
    dmid = softplus((dmid - F32(1)).astype(F32))
    dd = (dmid * deltas).astype(F32)
    alpha = (F32(1) - np.exp(-dd.astype(F64)).astype(F32)).astype(F32)
    a = ((F32(1) - alpha).astype(F32) + T_FLOOR).astype(F32)
    ones = np.ones_like(a[:, :, :1], dtype=F64)
    T = np.cumprod(np.concatenate([ones, a[:, :, :-1].astype(F64)], -2), -2).astype(F32)
    w = (alpha * T).astype(F32)
    rgb = (w * cmid).astype(F32).astype(F64).sum(-2).astype(F32)
    wt = w.astype(F64).sum(2).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        depth = ((w * zmid).astype(F32).astype(F64).sum(-2).astype(F32) / wt).astype(F32)
    depth = np.where(np.isnan(depth), F32(np.inf), depth)
    depth = np.clip(depth, depths.min(), depths.max()).astype(F32)
    if white_back:
        rgb = ((rgb + F32(1)).astype(F32) - wt).astype(F32)
    return rgb, depth, w


# ---------------------------------------------------------------------------- a12 importance
def sample_importance(z_vals, weights, n_importance, u):
    """renderer.py:197-256. z_vals (N,M,S,1), weights (N,M,S-1,1), u (N*M, F) -> (N,M,F,1)."""
    N, M, S, _ = z_vals.shape
    z = z_vals.reshape(N * M, S)
    w = weights.reshape(N * M, -1)                                     # S-1
    neg = np.full((N * M, 1), -np.inf, F32)
    wp = np.concatenate([neg, w, neg], 1)
    mx = np.maximum(wp[:, :-1], wp[:, 1:])                             # max_pool1d(2,1,pad 1) -> S
    av = ((mx[:, :-1] + mx[:, 1:]).astype(F32) / F32(2)).astype(F32)   # avg_pool1d(2,1) -> S-1
    av = (av + F32(0.01)).astype(F32)
    zmid = (F32(0.5) * (z[:, :-1] + z[:, 1:]).astype(F32)).astype(F32)  # S-1 bins
    wi = av[:, 1:-1]                                                   # S-3 weights
    nw = wi.shape[1]
    ww = (wi + F32(1e-5)).astype(F32)
    pdf = (ww / ww.astype(F64).sum(-1, keepdims=True).astype(F32)).astype(F32)
    cdf = np.concatenate([np.zeros((N * M, 1), F32), np.cumsum(pdf.astype(F64), -1).astype(F32)], -1)   # nw+1
    u = np.asarray(u, F32)
    inds = searchsorted(cdf, u, "right")
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, nw)
    cb, ca = np.take_along_axis(cdf, below, 1), np.take_along_axis(cdf, above, 1)
    bb, ba = np.take_along_axis(zmid, below, 1), np.take_along_axis(zmid, above, 1)
    denom = (ca - cb).astype(F32)
    denom = np.where(denom < F32(1e-5), F32(1), denom)
    s = (bb + (((u - cb).astype(F32) / denom).astype(F32) * (ba - bb).astype(F32)).astype(F32)).astype(F32)
    return s.reshape(N, M, n_importance, 1), dict(cdf=cdf, inds=inds)


def unify_samples(d1, c1, s1, d2, c2, s2):
    """renderer.py:160-170."""
    d = np.concatenate([d1, d2], -2)
    c = np.concatenate([c1, c2], -2)
    s = np.concatenate([s1, s2], -2)
    idx = np.argsort(d, axis=-2, kind="stable")
    return (np.take_along_axis(d, idx, -2), np.take_along_axis(c, np.broadcast_to(idx, c.shape), -2),
            np.take_along_axis(s, idx, -2))


def run_model(planes, dec, coords, box_warp, lr_mul=1.0):
    return osg_decoder(dec, sample_from_planes(planes, coords, box_warp), lr_mul)


def importance_renderer(planes, dec, ray_o, ray_d, opts, rand_strat, u, lr_mul=1.0):
    """ImportanceRenderer.forward (renderer.py:88-142), fixed ray_start/ray_end."""
    N, M, _ = ray_o.shape
    S, F = opts["depth_resolution"], opts["depth_resolution_importance"]
    dc = sample_stratified((N, M), opts["ray_start"], opts["ray_end"], S, rand_strat,
                           opts.get("disparity_space_sampling", False))
    coords = (ray_o[:, :, None, :] + (dc * ray_d[:, :, None, :]).astype(F32)).astype(F32).reshape(N, -1, 3)
    rgb, sig = run_model(planes, dec, coords, opts["box_warp"], lr_mul)
    cc, sc = rgb.reshape(N, M, S, 3), sig.reshape(N, M, S, 1)
    wb = opts.get("white_back", False)
    rgb_c, depth_c, w_c = mip_march(cc, sc, dc, wb)
    df, aux = sample_importance(dc, w_c, F, u)
    coords = (ray_o[:, :, None, :] + (df * ray_d[:, :, None, :]).astype(F32)).astype(F32).reshape(N, -1, 3)
    rgb, sig = run_model(planes, dec, coords, opts["box_warp"], lr_mul)
    cf, sf = rgb.reshape(N, M, F, 3), sig.reshape(N, M, F, 1)
    da, ca, sa = unify_samples(dc, cc, sc, df, cf, sf)
    rgb_f, depth_f, w_f = mip_march(ca, sa, da, wb)
    return (rgb_c, depth_c, w_c.astype(F64).sum(2).astype(F32), rgb_f, depth_f, w_f.astype(F64).sum(2).astype(F32),
            dict(depths_coarse=dc, depths_fine=df, all_depths=da, **aux))


# ---------------------------------------------------------------------------- a13 / a14
def ray_sampler(cam2world, intrinsics, res):
    """RaySampler.forward (ray_sampler.py:24-63)."""
    cam2world, intrinsics = np.asarray(cam2world, F32), np.asarray(intrinsics, F32)
    N = cam2world.shape[0]
    fx, fy, cx, cy, sk = (intrinsics[:, 0, 0], intrinsics[:, 1, 1], intrinsics[:, 0, 2], intrinsics[:, 1, 2],
                          intrinsics[:, 0, 1])
    ar = (np.arange(res, dtype=F32) * F32(1.0 / res)).astype(F32) + F32(0.5 / res)
    ar = ar.astype(F32)
    yy, xx = np.meshgrid(ar, ar, indexing="ij")                # uv.flip(0): x fastest
    x_cam = np.broadcast_to(xx.reshape(1, -1), (N, res * res))
    y_cam = np.broadcast_to(yy.reshape(1, -1), (N, res * res))
    c = lambda a: a[:, None]
    t1 = (x_cam - c(cx)).astype(F32)
    t2 = ((c(cy) * c(sk)).astype(F32) / c(fy)).astype(F32)
    t3 = ((c(sk) * y_cam).astype(F32) / c(fy)).astype(F32)
    x_lift = ((((t1 + t2).astype(F32) - t3).astype(F32)) / c(fx)).astype(F32)
    y_lift = ((y_cam - c(cy)).astype(F32) / c(fy)).astype(F32)
    pts = np.stack([x_lift, y_lift, np.ones_like(x_lift), np.ones_like(x_lift)], -1).astype(F32)   # (N,M,4)
    world = np.einsum("nij,nmj->nmi", cam2world, pts).astype(F32)[:, :, :3]
    loc = cam2world[:, :3, 3]
    d = (world - loc[:, None, :]).astype(F32)
    nrm = np.maximum(np.sqrt((d.astype(F64) ** 2).sum(-1, keepdims=True)).astype(F32), F32(1e-12))
    d = (d / nrm).astype(F32)
    o = np.broadcast_to(loc[:, None, :], d.shape).copy()
    return o, d


def get_ray_limits_box(rays_o, rays_d, box_side_length):
    """math_utils.py:46-98."""
    shp = rays_o.shape
    o = np.asarray(rays_o, F32).reshape(-1, 3)
    d = np.asarray(rays_d, F32).reshape(-1, 3)
    h = F32(box_side_length / 2)
    bounds = np.array([[-h, -h, -h], [h, h, h]], F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (F32(1) / d).astype(F32)
        sign = (inv < 0).astype(np.int64)
        valid = np.ones(o.shape[0], bool)
        ax = lambda s, k: bounds[s[:, k], k]
        tmin = ((ax(sign, 0) - o[:, 0]).astype(F32) * inv[:, 0]).astype(F32)
        tmax = ((ax(1 - sign, 0) - o[:, 0]).astype(F32) * inv[:, 0]).astype(F32)
        tymin = ((ax(sign, 1) - o[:, 1]).astype(F32) * inv[:, 1]).astype(F32)
        tymax = ((ax(1 - sign, 1) - o[:, 1]).astype(F32) * inv[:, 1]).astype(F32)
        valid[(tmin > tymax) | (tymin > tmax)] = False
        tmin = np.maximum(tmin, tymin)
        tmax = np.minimum(tmax, tymax)
        tzmin = ((ax(sign, 2) - o[:, 2]).astype(F32) * inv[:, 2]).astype(F32)
        tzmax = ((ax(1 - sign, 2) - o[:, 2]).astype(F32) * inv[:, 2]).astype(F32)
        valid[(tmin > tzmax) | (tzmin > tmax)] = False
        tmin = np.maximum(tmin, tzmin)
        tmax = np.minimum(tmax, tzmax)
    tmin = np.where(valid, tmin, F32(-1)).astype(F32)
    tmax = np.where(valid, tmax, F32(-2)).astype(F32)
    return tmin.reshape(*shp[:-1], 1), tmax.reshape(*shp[:-1], 1)
