"""CPU oracle for the render_rays hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A numpy restatement of the reference algorithm (Freedomcls/nerf-siren):
    models/rendering.py:22-67    sample_pdf
    models/rendering.py:70-262   render_rays (+ nested inference :105-190)
    models/nerf.py:4-38          Embedding
    models/nerf.py:41-124        NeRF
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (nerf_siren_amd/) never does.

Parity status: PINNED.  tests/golden/*.npz hold outputs of the reference itself
(imported from /root/reference by tools/make_golden.py in the build container);
tests/test_oracle_golden.py checks every function below against them.

Arithmetic specification ("nerfmi arithmetic").  The reference runs op-by-op in
fp32 on torch's CPU backend; what is observable there and restated here:
  * every elementwise op rounds to fp32 once (no fused multiply-add);
  * torch.linspace(0,1,n)[i] = i<n/2 ? fl(step*i) : fl(1 - step*(n-1-i)) with
    fp32 step = 1/(n-1) and a single rounding (fma form)       [probed, exact];
  * torch.cumsum / torch.cumprod accumulate in fp64 and store each prefix
    rounded to fp32                                             [probed, exact];
  * torch.sum over a row has an ISA-dependent order: restated as fp64
    accumulation rounded once (differs from the reference by <= 1 ulp);
  * torch.searchsorted(cdf, u, right=True) = #{k : cdf[k] <= u}.
The HIP kernels implement exactly this specification, so kernel == oracle is
checked much tighter (bit-exact indices/depths given the same inputs) than
oracle == reference (1e-6 .. 1e-4, see the tests).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F64 = np.float64

EPS_PDF = F32(1e-5)        # rendering.py:22 (eps), :37, :63
T_FLOOR = F32(1e-10)       # rendering.py:175
DELTA_INF = F32(1e10)      # rendering.py:163


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def linspace01(n: int) -> np.ndarray:
    """torch.linspace(0, 1, n) on CPU, fp32 (rendering.py:207, :44)."""
    if n == 1:
        return np.zeros(1, F32)
    step = F32(F32(1.0) / F32(n - 1))
    i = np.arange(n)
    lo = (F64(step) * i).astype(F32)                 # fma(step, i, 0)
    hi = (1.0 - F64(step) * (n - 1 - i)).astype(F32)  # fma(-step, n-1-i, 1)
    return np.where(i < n // 2, lo, hi).astype(F32)


def searchsorted(a: np.ndarray, v: np.ndarray, side: str = "right") -> np.ndarray:
    """Row-wise searchsorted with row broadcasting, int64 result.

    torchsearchsorted/src/torchsearchsorted/searchsorted.py:20-53 and
    test/utils.py:4-15 (numpy_searchsorted); rendering.py:54 uses right=True.
    """
    a = np.asarray(a)
    v = np.asarray(v)
    nrow = max(a.shape[0], v.shape[0])
    out = np.empty((nrow, v.shape[1]), np.int64)
    for r in range(nrow):
        ra = a[r if a.shape[0] > 1 else 0]
        rv = v[r if v.shape[0] > 1 else 0]
        out[r] = np.searchsorted(ra, rv, side=side)
    return out


def embed(x: np.ndarray, n_freqs: int) -> np.ndarray:
    """Embedding.forward (nerf.py:21-38): [x, sin(2^k x), cos(2^k x)]_{k<n_freqs}.

    2^k * x is exact in fp32; sin/cos are evaluated in fp64 and rounded once
    (the correctly rounded value any <=1-ulp libm is compared against).
    """
    x = np.asarray(x, F32)
    out = [x]
    for k in range(n_freqs):
        arg = (x * F32(2.0 ** k)).astype(F32).astype(F64)
        out.append(np.sin(arg).astype(F32))
        out.append(np.cos(arg).astype(F32))
    return np.concatenate(out, -1)


# ----------------------------------------------------------------------------
# NeRF MLP (nerf.py:41-124), parameters keyed exactly like the state_dict
# ----------------------------------------------------------------------------
def _lin(p, name, x):
    return (x @ p[name + ".weight"].T + p[name + ".bias"]).astype(F32)


def nerf_forward(p: dict, x: np.ndarray, sigma_only: bool = False, keep: bool = False):
    """NeRF.forward (nerf.py:83-124). x: (B,63+27) or (B,63) when sigma_only.

    Returns out (B,4)=[rgb,sigma] or (B,1); with keep=True also the activation
    cache the manual backward needs.
    """
    x = np.asarray(x, F32)
    in_xyz = x[:, :63]
    h = in_xyz
    acts = []                      # inputs of each xyz_encoding layer
    pres = []                      # pre-activations (tests: which units sit on the ReLU kink)
    for i in range(8):
        if i == 4:                 # skips=[4], nerf.py:108-109
            h = np.concatenate([in_xyz, h], -1)
        acts.append(h)
        pre = _lin(p, f"xyz_encoding_{i+1}.0", h)
        pres.append(pre)
        h = np.maximum(pre, F32(0))
    sigma = _lin(p, "sigma", h)
    if sigma_only:
        return (sigma, dict(acts=acts, h8=h, pres=pres)) if keep else sigma
    final = _lin(p, "xyz_encoding_final", h)
    dir_in = np.concatenate([final, x[:, 63:90]], -1)       # nerf.py:118
    dir_pre = _lin(p, "dir_encoding.0", dir_in)
    dir_h = np.maximum(dir_pre, F32(0))
    rgb_pre = _lin(p, "rgb.0", dir_h)
    rgb = (F32(1) / (F32(1) + np.exp(-rgb_pre.astype(F64)))).astype(F32)
    out = np.concatenate([rgb, sigma], -1).astype(F32)
    if keep:
        return out, dict(acts=acts, h8=h, dir_in=dir_in, dir_h=dir_h, rgb=rgb, pres=pres, dir_pre=dir_pre)
    return out


def nerf_backward(p: dict, cache: dict, grad_out: np.ndarray, sigma_only: bool = False, masks: dict | None = None) -> dict:
    """Manual backward of nerf_forward w.r.t. every parameter (autograd of
    nerf.py:83-124).  grad_out: (B,4) [d rgb, d sigma] or (B,1).
    masks (tests only): {'h': [8 x (B,256) bool], 'dir': (B,128) bool} -- the ReLU derivative masks (output > 0) to use
    INSTEAD of the oracle's own: a pre-activation within rounding of 0 lands on either side of the kink under a different
    fp32 summation order, and the parity tests exempt exactly those units by evaluating both sides with the same masks."""
    g = {}
    mask_h = [None] * 8 if masks is None else list(masks["h"])
    mask_dir = None if masks is None else masks.get("dir")
    B = grad_out.shape[0]
    h8 = cache["h8"]
    if sigma_only:
        d_sigma = grad_out.reshape(B, 1)
        d_h8 = np.zeros_like(h8)
    else:
        d_rgb = grad_out[:, :3]
        d_sigma = grad_out[:, 3:4]
        rgb = cache["rgb"]
        d_pre = (d_rgb * rgb * (F32(1) - rgb)).astype(F32)
        g["rgb.0.weight"] = d_pre.T @ cache["dir_h"]
        g["rgb.0.bias"] = d_pre.sum(0)
        d_dir_h = d_pre @ p["rgb.0.weight"]
        d_dir_h = d_dir_h * ((cache["dir_h"] > 0) if mask_dir is None else mask_dir)
        g["dir_encoding.0.weight"] = d_dir_h.T @ cache["dir_in"]
        g["dir_encoding.0.bias"] = d_dir_h.sum(0)
        d_final = (d_dir_h @ p["dir_encoding.0.weight"])[:, :256]
        g["xyz_encoding_final.weight"] = d_final.T @ h8
        g["xyz_encoding_final.bias"] = d_final.sum(0)
        d_h8 = d_final @ p["xyz_encoding_final.weight"]
    g["sigma.weight"] = d_sigma.T @ h8
    g["sigma.bias"] = d_sigma.sum(0)
    d_h = d_h8 + d_sigma @ p["sigma.weight"]
    h_out = h8
    for i in reversed(range(8)):
        name = f"xyz_encoding_{i+1}.0"
        d_pre = (d_h * ((h_out > 0) if mask_h[i] is None else mask_h[i])).astype(F32)
        x_in = cache["acts"][i]
        g[name + ".weight"] = d_pre.T @ x_in
        g[name + ".bias"] = d_pre.sum(0)
        if i == 0:
            break
        d_in = d_pre @ p[name + ".weight"]
        if i == 4:
            d_in = d_in[:, 63:]
            h_out = x_in[:, 63:]
        else:
            h_out = x_in
        d_h = d_in
    return {k: np.asarray(v, F32) for k, v in g.items()}


# ----------------------------------------------------------------------------
# a2  stratified sampler (rendering.py:207-225)
# ----------------------------------------------------------------------------
def sample_z(rays: np.ndarray, n_samples: int, use_disp: bool = False,
             perturb: float = 0.0, perturb_rand: np.ndarray | None = None) -> np.ndarray:
    """z_vals (N,S). perturb_rand = the torch.rand(N,S) draw of rendering.py:221
    (before the multiplication by `perturb`)."""
    rays = np.asarray(rays, F32)
    near, far = rays[:, 6:7], rays[:, 7:8]
    t = linspace01(n_samples)[None, :]
    omt = (F32(1) - t).astype(F32)
    if not use_disp:
        z = ((near * omt).astype(F32) + (far * t).astype(F32)).astype(F32)
    else:
        inv_n = (F32(1) / near).astype(F32)
        inv_f = (F32(1) / far).astype(F32)
        z = (F32(1) / ((inv_n * omt).astype(F32) + (inv_f * t).astype(F32)).astype(F32)).astype(F32)
    z = np.broadcast_to(z, (rays.shape[0], n_samples)).copy()
    if perturb > 0:
        mid = (F32(0.5) * (z[:, :-1] + z[:, 1:]).astype(F32)).astype(F32)
        upper = np.concatenate([mid, z[:, -1:]], -1)
        lower = np.concatenate([z[:, :1], mid], -1)
        pr = (F32(perturb) * np.asarray(perturb_rand, F32)).astype(F32)
        z = (lower + ((upper - lower).astype(F32) * pr).astype(F32)).astype(F32)
    return z


def points(rays: np.ndarray, z: np.ndarray) -> np.ndarray:
    """xyz = o + d*z (rendering.py:224-225, :249-250), (N,S,3)."""
    o, d = rays[:, None, 0:3], rays[:, None, 3:6]
    return (o + (d * z[:, :, None]).astype(F32)).astype(F32)


# ----------------------------------------------------------------------------
# a8  compositing (rendering.py:162-190) forward + backward
# ----------------------------------------------------------------------------
def ray_norm(d: np.ndarray) -> np.ndarray:
    """torch.norm(dir_, dim=-1) (rendering.py:168): fp32, sequential x,y,z."""
    d = np.asarray(d, F32)
    s = ((d[:, 0] * d[:, 0]).astype(F32) + (d[:, 1] * d[:, 1]).astype(F32)).astype(F32)
    s = (s + (d[:, 2] * d[:, 2]).astype(F32)).astype(F32)
    return np.sqrt(s).astype(F32)


def composite(sigma, rgb, z, rays_d, noise=None, noise_std: float = 0.0,
              white_back: bool = False, keep: bool = False):
    """inference() tail, rendering.py:162-190.

    sigma (N,P), rgb (N,P,3) or None (weights_only), z (N,P), rays_d (N,3),
    noise = the randn(N,P) draw of :170 (before * noise_std) or None.
    Returns dict(weights, opacity[, rgb, depth]).
    """
    sigma = np.asarray(sigma, F32)
    z = np.asarray(z, F32)
    N, P = sigma.shape
    delta = np.empty((N, P), F32)
    delta[:, :-1] = (z[:, 1:] - z[:, :-1]).astype(F32)
    delta[:, -1] = DELTA_INF
    delta = (delta * ray_norm(rays_d)[:, None]).astype(F32)
    s = sigma
    if noise is not None:
        s = (sigma + (np.asarray(noise, F32) * F32(noise_std)).astype(F32)).astype(F32)
    relu_s = np.maximum(s, F32(0))
    e = np.exp((-(delta * relu_s).astype(F32)).astype(F64)).astype(F32)
    alpha = (F32(1) - e).astype(F32)
    a = ((F32(1) - alpha).astype(F32) + T_FLOOR).astype(F32)     # 1-alpha+1e-10
    T64 = np.cumprod(np.concatenate([np.ones((N, 1), F64), a[:, :-1].astype(F64)], -1), -1)
    T = T64.astype(F32)                                          # exclusive product
    w = (alpha * T).astype(F32)
    out = {"weights": w, "opacity": w.astype(F64).sum(-1).astype(F32)}
    if rgb is not None:
        rgb = np.asarray(rgb, F32)
        c = (w[:, :, None] * rgb).astype(F32).astype(F64).sum(1).astype(F32)
        depth = (w * z).astype(F32).astype(F64).sum(-1).astype(F32)
        if white_back:
            c = ((c + F32(1)).astype(F32) - out["opacity"][:, None]).astype(F32)
        out["rgb"] = c
        out["depth"] = depth
    if keep:
        out["_cache"] = dict(delta=delta, s=s, alpha=alpha, a=a, T=T, rgb=rgb, z=z)
    return out


def composite_backward(cache, w, g_rgb, g_depth, g_opacity, white_back: bool):
    """d loss / d (sigma, rgb) of composite() -- SURVEY section 8a backward
    contract. g_rgb (N,3), g_depth (N,), g_opacity (N,). fp64 internally."""
    alpha = cache["alpha"].astype(F64)
    a = cache["a"].astype(F64)
    T = cache["T"].astype(F64)
    rgb = cache["rgb"].astype(F64)
    z = cache["z"].astype(F64)
    w = w.astype(F64)
    g_rgb = g_rgb.astype(F64)
    v = (rgb * g_rgb[:, None, :]).sum(-1) + z * g_depth[:, None].astype(F64) + g_opacity[:, None].astype(F64)
    if white_back:
        v = v - g_rgb.sum(-1)[:, None]
    d_rgb = w[:, :, None] * g_rgb[:, None, :]
    wv = w * v
    suffix = np.cumsum(wv[:, ::-1], -1)[:, ::-1] - wv        # sum_{j>i} w_j v_j
    d_alpha = T * v - suffix / a
    d_s = d_alpha * cache["delta"].astype(F64) * (1.0 - alpha) * (cache["s"] > 0)
    return d_s.astype(F32), d_rgb.astype(F32)


# ----------------------------------------------------------------------------
# a3  sample_pdf (rendering.py:22-67)
# ----------------------------------------------------------------------------
def build_cdf(weights: np.ndarray) -> np.ndarray:
    """rendering.py:36-40. weights (N,nw) -> cdf (N,nw+1) with leading 0."""
    w = (np.asarray(weights, F32) + EPS_PDF).astype(F32)
    tot = w.astype(F64).sum(-1, keepdims=True).astype(F32)
    pdf = (w / tot).astype(F32)
    cdf = np.cumsum(pdf.astype(F64), -1).astype(F32)
    return np.concatenate([np.zeros_like(cdf[:, :1]), cdf], -1)


def search_lerp(bins: np.ndarray, cdf: np.ndarray, u: np.ndarray):
    """rendering.py:54-66 given (bins, cdf, u). Returns (inds int64, samples)."""
    bins = np.asarray(bins, F32)
    cdf = np.asarray(cdf, F32)
    u = np.asarray(u, F32)
    nw = cdf.shape[1] - 1                                   # N_samples_
    inds = searchsorted(cdf, u, "right")
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, nw)
    cdf_b = np.take_along_axis(cdf, below, 1)
    cdf_a = np.take_along_axis(cdf, above, 1)
    bin_b = np.take_along_axis(bins, below, 1)
    bin_a = np.take_along_axis(bins, above, 1)
    denom = (cdf_a - cdf_b).astype(F32)
    denom = np.where(denom < EPS_PDF, F32(1), denom)
    tt = ((u - cdf_b).astype(F32) / denom).astype(F32)
    samples = (bin_b + (tt * (bin_a - bin_b).astype(F32)).astype(F32)).astype(F32)
    return inds, samples


def sample_pdf(bins, weights, n_importance: int, det: bool = False, u=None):
    """sample_pdf (rendering.py:22-67). u = the torch.rand(N,F) draw when not det."""
    cdf = build_cdf(weights)
    if det:
        u = np.broadcast_to(linspace01(n_importance)[None, :], (cdf.shape[0], n_importance))
    inds, samples = search_lerp(bins, cdf, u)
    return samples, dict(cdf=cdf, inds=inds, u=np.asarray(u, F32))


def midpoints(z: np.ndarray) -> np.ndarray:
    """z_vals_mid (rendering.py:242)."""
    return (F32(0.5) * (z[:, :-1] + z[:, 1:]).astype(F32)).astype(F32)


# ----------------------------------------------------------------------------
# a1  render_rays (rendering.py:70-262) forward, and the training backward
# ----------------------------------------------------------------------------
def _field(p, rays, z, sigma_only, keep):
    """inference() head, rendering.py:131-159: embed + NeRF per point."""
    N, P = z.shape
    xyz = points(rays, z).reshape(-1, 3)
    x = embed(xyz, 10)
    if not sigma_only:
        d_emb = embed(rays[:, 3:6], 4)
        x = np.concatenate([x, np.repeat(d_emb, P, 0)], -1)
    r = nerf_forward(p, x, sigma_only, keep)
    out, cache = r if keep else (r, None)
    if sigma_only:
        return out.reshape(N, P), None, cache
    out = out.reshape(N, P, 4)
    return out[..., 3], out[..., :3], cache


def render_rays(params, rays, N_samples=64, use_disp=False, perturb=0.0, noise_std=1.0,
                N_importance=0, white_back=False, test_time=False, rng=None, keep=False):
    """render_rays (rendering.py:70-262). params = [coarse, fine] dicts.

    rng (dict, all optional): perturb_rand (N,S) = draw #1 (:221),
    noise_coarse (N,S) = draw #2 (:170), u (N,F) = draw #3 (:47),
    noise_fine (N,S+F) = draw #4; z_fine (N,S+F) overrides the merged depths
    (test hook: sample_pdf is ill-conditioned in ~zero-weight bins, so stage
    tests condition on the reference's depths).  Returns the result dict (+ '_aux').
    """
    rng = rng or {}
    rays = np.asarray(rays, F32)
    d = rays[:, 3:6]
    z = sample_z(rays, N_samples, use_disp, perturb, rng.get("perturb_rand"))
    aux = {"z_coarse": z}
    sig, rgb, mc = _field(params[0], rays, z, test_time, keep)
    cc = composite(sig, rgb, z, d, rng.get("noise_coarse"), noise_std, white_back, keep)
    res = {"opacity_coarse": cc["opacity"]}
    if not test_time:
        res["rgb_coarse"] = cc["rgb"]
        res["depth_coarse"] = cc["depth"]
    aux.update(weights_coarse=cc["weights"], sigma_coarse=sig, rgb_raw_coarse=rgb)
    if N_importance > 0:
        zmid = midpoints(z)
        z_new, pa = sample_pdf(zmid, cc["weights"][:, 1:-1], N_importance,
                               det=(perturb == 0), u=rng.get("u"))
        z_all = np.sort(np.concatenate([z, z_new], -1), -1)
        if "z_fine" in rng:          # test hook: condition on the reference's own merged depths
            z_all = np.asarray(rng["z_fine"], F32)
        sig_f, rgb_f, mf = _field(params[1], rays, z_all, False, keep)
        cf = composite(sig_f, rgb_f, z_all, d, rng.get("noise_fine"), noise_std, white_back, keep)
        res["rgb_fine"] = cf["rgb"]
        res["depth_fine"] = cf["depth"]
        res["opacity_fine"] = cf["opacity"]
        aux.update(z_new=z_new, z_fine=z_all, cdf=pa["cdf"], inds=pa["inds"],
                   weights_fine=cf["weights"], sigma_fine=sig_f, rgb_raw_fine=rgb_f)
        if keep:
            aux["_fine"] = (mf, cf)
    if keep:
        aux["_coarse"] = (mc, cc)
    res["_aux"] = aux
    return res


def render_rays_backward(params, res, grads, white_back=False, masks=None):
    """Parameter gradients of a training-mode render_rays() call made with
    keep=True.  grads: dict name -> dL/d(output) for any of rgb_/depth_/opacity_
    {coarse,fine}.  Returns [g_coarse, g_fine] (fine None when absent).
    No gradient crosses sample_pdf (rendering.py:54 cdf.detach(), :244 .detach()).
    masks (tests only): [coarse, fine] ReLU masks for nerf_backward (see there), or None."""
    aux = res["_aux"]
    out = []
    for ti, tag in enumerate(("coarse", "fine")):
        key = "_" + tag
        if key not in aux:
            out.append(None)
            continue
        mcache, cres = aux[key]
        N = cres["weights"].shape[0]
        zero = np.zeros(N, F32)
        g_rgb = grads.get("rgb_" + tag, np.zeros((N, 3), F32))
        g_dep = grads.get("depth_" + tag, zero)
        g_op = grads.get("opacity_" + tag, zero)
        d_s, d_rgb = composite_backward(cres["_cache"], cres["weights"], g_rgb, g_dep, g_op, white_back)
        g_out = np.concatenate([d_rgb, d_s[:, :, None]], -1).reshape(-1, 4)
        out.append(nerf_backward(params[0 if tag == "coarse" else 1], mcache, g_out,
                                 masks=None if masks is None else masks[ti]))
    return out


# ----------------------------------------------------------------------------
# a7  FiLM-SIREN field: FiLMLayer (nerf.py:142-151) and
#     SemanticNeRF.forward_with_frequencies_phase_shifts (nerf.py:201-216)
# ----------------------------------------------------------------------------
def film_layer(w, b, x, freq, phase, keep: bool = False):
    """FiLMLayer.forward: sin(freq * (x W^T + b) + phase); freq/phase (Bz,H) broadcast over points."""
    y = (x @ w.T + b).astype(F32)
    arg = ((freq[:, None, :] * y).astype(F32) + phase[:, None, :]).astype(F32)
    out = np.sin(arg.astype(F64)).astype(F32)
    return (out, arg, y) if keep == "pre" else ((out, arg) if keep else out)


def siren_forward(p: dict, inp, frequencies, phase_shifts, ray_directions, sigma_only: bool = False,
                  keep: bool = False):
    """inp (Bz,Np,3), frequencies/phase_shifts (Bz, 9*256), ray_directions (Bz,Np,3) -> (Bz,Np,4) [rgb,sigma].
    keep=True also returns the cache siren_backward needs."""
    inp = np.asarray(inp, F32)
    H = 256
    fr = ((np.asarray(frequencies, F32) * F32(15)).astype(F32) + F32(30)).astype(F32)       # nerf.py:202
    ph = np.asarray(phase_shifts, F32)
    x = (inp * F32(2.0 / 51.0)).astype(F32)                                                   # UniformBoxWarp(51), :134-140,:193
    xs, args, pres = [], [], []
    for i in range(8):
        xs.append(x)
        x, a, y = film_layer(p[f"network.{i}.layer.weight"], p[f"network.{i}.layer.bias"], x,
                             fr[:, i * H:(i + 1) * H], ph[:, i * H:(i + 1) * H], keep="pre")
        args.append(a)
        pres.append(y)
    sigma = (x @ p["final_layer.weight"].T + p["final_layer.bias"]).astype(F32)
    if sigma_only:
        return (sigma, dict(xs=xs, args=args, h=x, fr=fr)) if keep else sigma
    cin = np.concatenate([np.asarray(ray_directions, F32), x], -1)                            # :213
    c, carg, cpre = film_layer(p["color_layer_sine.layer.weight"], p["color_layer_sine.layer.bias"], cin, fr[:, -H:],
                               ph[:, -H:], keep="pre")
    pre = (c @ p["color_layer_linear.0.weight"].T + p["color_layer_linear.0.bias"]).astype(F32)
    rgb = (F32(1) / (F32(1) + np.exp(-pre.astype(F64)))).astype(F32)
    out = np.concatenate([rgb, sigma], -1).astype(F32)
    if keep:
        return out, dict(xs=xs, args=args, pres=pres, h=x, fr=fr, cin=cin, c=c, carg=carg, cpre=cpre, rgb=rgb)
    return out


def siren_backward(p: dict, cache: dict, grad_out: np.ndarray, cond: bool = False):
    """Manual backward of siren_forward (autograd of nerf.py:142-151, :201-216) w.r.t. the 22 parameters.
    grad_out (Bz,Np,4) = [d rgb, d sigma].  d/dz sin(fr*z + ph) = fr * cos(fr*z + ph).
    cond=True: -> (parameter gradients, d frequencies (Bz, 9*256), d phase_shifts (Bz, 9*256)): with fr = 15 f + 30 (:202)
    d/d ph = cos(arg), d/d f = 15 z cos(arg), each summed over the points of its conditioning row."""
    H = 256
    fr = cache["fr"]
    g = {}
    d_f = np.zeros(fr.shape, F64)
    d_p = np.zeros(fr.shape, F64)

    def cond_grads(sl, d_act, arg, pre):
        gc = d_act.astype(F64) * np.cos(arg.astype(F64))                                      # (Bz, Np, H)
        d_p[:, sl] = gc.sum(1)
        d_f[:, sl] = 15.0 * (gc * pre.astype(F64)).sum(1)

    def flat(a):
        return a.reshape(-1, a.shape[-1])

    d_rgb, d_sigma = grad_out[..., :3], grad_out[..., 3:4]
    rgb = cache["rgb"]
    d_pre = (d_rgb * rgb * (F32(1) - rgb)).astype(F32)                                       # sigmoid, :214
    g["color_layer_linear.0.weight"] = flat(d_pre).T @ flat(cache["c"])
    g["color_layer_linear.0.bias"] = flat(d_pre).sum(0)
    d_c = d_pre @ p["color_layer_linear.0.weight"]
    dz = (d_c * fr[:, None, -H:] * np.cos(cache["carg"].astype(F64)).astype(F32)).astype(F32)
    cond_grads(slice(8 * H, 9 * H), d_c, cache["carg"], cache["cpre"])
    g["color_layer_sine.layer.weight"] = flat(dz).T @ flat(cache["cin"])                      # columns [dir 3 | hidden 256], :213
    g["color_layer_sine.layer.bias"] = flat(dz).sum(0)
    d_h = (dz @ p["color_layer_sine.layer.weight"])[..., 3:] + d_sigma @ p["final_layer.weight"]
    g["final_layer.weight"] = flat(d_sigma).T @ flat(cache["h"])
    g["final_layer.bias"] = flat(d_sigma).sum(0)
    for i in reversed(range(8)):
        dz = (d_h * fr[:, None, i * H:(i + 1) * H] * np.cos(cache["args"][i].astype(F64)).astype(F32)).astype(F32)
        cond_grads(slice(i * H, (i + 1) * H), d_h, cache["args"][i], cache["pres"][i])
        g[f"network.{i}.layer.weight"] = flat(dz).T @ flat(cache["xs"][i])
        g[f"network.{i}.layer.bias"] = flat(dz).sum(0)
        if i:
            d_h = dz @ p[f"network.{i}.layer.weight"]
    g = {k: np.asarray(v, F32) for k, v in g.items()}
    return (g, d_f.astype(F32), d_p.astype(F32)) if cond else g


# ----------------------------------------------------------------------------
# f2  the step after the path: losses.py:10-20 (MSELoss), metrics.py:4-13 (mse / psnr),
#     utils/__init__.py:20 -> torch.optim.Adam (torch/optim/adam.py _single_tensor_adam, amsgrad=False)
# ----------------------------------------------------------------------------
def mse_loss(rgb_coarse, rgb_fine, targets, grad_out=1.0):
    """-> dict(loss, mse_coarse, mse_fine, psnr, g_coarse, g_fine); either prediction may be None.
    nn.MSELoss(mean): mean((x-t)^2) (fp64 accumulate here); backward (2/numel)*(x-t)*grad_out, fp32 op by op."""
    t = np.asarray(targets, F32)
    n = t.size
    norm = F32(2.0 / n)
    out = {"g_coarse": None, "g_fine": None, "mse_coarse": F32(0), "mse_fine": F32(0)}
    for key, x in (("coarse", rgb_coarse), ("fine", rgb_fine)):
        if x is None:
            continue
        d = (np.asarray(x, F32) - t).astype(F32)
        out["mse_" + key] = F32((d * d).astype(F32).astype(F64).sum() / n)
        out["g_" + key] = ((norm * d).astype(F32) * F32(grad_out)).astype(F32)
    if rgb_coarse is not None and rgb_fine is not None:
        out["loss"] = F32(out["mse_coarse"] + out["mse_fine"])
    else:
        out["loss"] = out["mse_coarse"] if rgb_coarse is not None else out["mse_fine"]
    out["psnr"] = F32(-10.0) * np.log10(out["mse_fine"] if rgb_fine is not None else out["mse_coarse"], dtype=F32)
    return out


def adam_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """One torch.optim.Adam update of fp32 arrays; returns (p, m, v).  The Python-float scalars are formed in
    double and rounded once, every tensor op is one fp32 rounding (torch fuses lerp's multiply-add: <= 1 ulp)."""
    p, g, m, v = (np.asarray(a, F32) for a in (p, g, m, v))
    b1, b2 = betas
    if weight_decay != 0:
        g = (g + (F32(weight_decay) * p).astype(F32)).astype(F32)
    m = (m + (F32(1.0 - b1) * (g - m).astype(F32)).astype(F32)).astype(F32)
    v = ((v * F32(b2)).astype(F32) + ((F32(1.0 - b2) * g).astype(F32) * g).astype(F32)).astype(F32)
    step_size = lr / (1.0 - b1 ** step)
    bc2_sqrt = (1.0 - b2 ** step) ** 0.5
    denom = ((np.sqrt(v).astype(F32) / F32(bc2_sqrt)).astype(F32) + F32(eps)).astype(F32)
    p = (p + ((F32(-step_size) * m).astype(F32) / denom).astype(F32)).astype(F32)
    return p, m, v


# ----------------------------------------------------------------------------
# f1  the step before the path: datasets/ray_utils.py:5-93 and the (N,8) packing of
#     datasets/blender.py:60-69 / datasets/llff.py:234-250.
#     PARITY UNPINNED for this block: ray_utils.py imports kornia at module level and kornia is absent, so the
#     reference's functions cannot be executed here; they are restated line by line (create_meshgrid(H, W, False)
#     = integer pixel coordinates, i = column, j = row) and tests check the restatement against the same
#     formulas evaluated with torch CPU ops.
# ----------------------------------------------------------------------------
def ray_directions(H: int, W: int, focal: float) -> np.ndarray:
    """get_ray_directions: (H, W, 3) = ((i - W/2)/focal, -(j - H/2)/focal, -1)."""
    i = np.broadcast_to(np.arange(W, dtype=F32)[None, :], (H, W))
    j = np.broadcast_to(np.arange(H, dtype=F32)[:, None], (H, W))
    f = F32(focal)
    dx = ((i - F32(W / 2)).astype(F32) / f).astype(F32)
    dy = ((-(j - F32(H / 2)).astype(F32)) / f).astype(F32)
    return np.stack([dx, dy, -np.ones_like(dx)], -1).astype(F32)


def get_rays(directions, c2w):
    """get_rays: directions (...,3), c2w (3,4) -> rays_o (n,3), rays_d (n,3) (normalised)."""
    d = np.asarray(directions, F32).reshape(-1, 3)
    c = np.asarray(c2w, F32)
    r = np.stack([(((d[:, 0] * c[k, 0]).astype(F32) + (d[:, 1] * c[k, 1]).astype(F32)).astype(F32)
                   + (d[:, 2] * c[k, 2]).astype(F32)).astype(F32) for k in range(3)], -1)
    r = (r / ray_norm(r)[:, None]).astype(F32)
    return np.broadcast_to(c[:, 3], r.shape).astype(F32), r


def ndc_rays(H: int, W: int, focal: float, near: float, rays_o, rays_d):
    """get_ndc_rays, fp32 op by op; Python-float constants formed in double and rounded once."""
    o = np.asarray(rays_o, F32)
    d = np.asarray(rays_d, F32)
    t = ((-(F32(near) + o[:, 2]).astype(F32)) / d[:, 2]).astype(F32)
    o = (o + (t[:, None] * d).astype(F32)).astype(F32)
    ox_oz = (o[:, 0] / o[:, 2]).astype(F32)
    oy_oz = (o[:, 1] / o[:, 2]).astype(F32)
    cw, ch = F32(-1.0 / (W / (2.0 * focal))), F32(-1.0 / (H / (2.0 * focal)))
    o0 = (cw * ox_oz).astype(F32)
    o1 = (ch * oy_oz).astype(F32)
    o2 = (F32(1) + ((F32(1) / o[:, 2]).astype(F32) * F32(2.0 * near)).astype(F32)).astype(F32)   # scalar / tensor = reciprocal * scalar
    d0 = (cw * ((d[:, 0] / d[:, 2]).astype(F32) - ox_oz).astype(F32)).astype(F32)
    d1 = (ch * ((d[:, 1] / d[:, 2]).astype(F32) - oy_oz).astype(F32)).astype(F32)
    d2 = (F32(1) - o2).astype(F32)
    return np.stack([o0, o1, o2], -1).astype(F32), np.stack([d0, d1, d2], -1).astype(F32)


def generate_rays(c2w, H: int, W: int, focal: float, pixel_index=None, ndc: bool = False, near: float = 2.0,
                  far: float = 6.0) -> np.ndarray:
    """(n,8) [o, d, near, far] rows for pixel_index = image*H*W + row*W + column (default: all pixels of all images)."""
    c2w = np.asarray(c2w, F32).reshape(-1, 3, 4)
    if pixel_index is None:
        pixel_index = np.arange(c2w.shape[0] * H * W, dtype=np.int64)
    pixel_index = np.asarray(pixel_index, np.int64)
    dirs = ray_directions(H, W, focal).reshape(-1, 3)
    img, pix = pixel_index // (H * W), pixel_index % (H * W)
    out = np.empty((pixel_index.size, 8), F32)
    for k in np.unique(img):
        sel = img == k
        o, d = get_rays(dirs[pix[sel]], c2w[k])
        if ndc:
            o, d = ndc_rays(H, W, focal, 1.0, o, d)
        out[sel, 0:3], out[sel, 3:6] = o, d
    out[:, 6], out[:, 7] = (0.0, 1.0) if ndc else (near, far)
    return out


# ----------------------------------------------------------------------------
# f3  dense field query: extract_color_mesh.py:117-140, extract_mesh.ipynb cells 4 and 7 (restated; the scripts
#     themselves need mcubes/open3d/datasets and cannot run here -- the field evaluation they call is pinned above)
# ----------------------------------------------------------------------------
def grid_points(N, x_range, y_range, z_range):
    x = np.linspace(x_range[0], x_range[1], N)
    y = np.linspace(y_range[0], y_range[1], N)
    z = np.linspace(z_range[0], z_range[1], N)
    return np.stack(np.meshgrid(x, y, z), -1).reshape(-1, 3).astype(F32)


def create_samples(N=256, voxel_origin=(0, 0, 0), cube_length=2.0):
    """extract_color_mesh_eg3d.py:72-94 (the DeepSDF grid helper as this fork runs it).  NB the y and x columns are
    built with FLOAT division -- (idx.float() / N) % N, ((idx.float() / N) / N) % N -- so they are not integer voxel
    indices but vary continuously with the flat index; restated as is (fp32 arithmetic of the torch ops).
    Returns samples (1, N^3, 3) with columns scaled as in :88-90, voxel_origin (3,) float64, voxel_size."""
    origin = np.array(voxel_origin, F64) - cube_length / 2
    voxel_size = cube_length / (N - 1)
    idx = np.arange(N ** 3, dtype=np.int64)
    fN = F32(N)
    s = np.zeros((N ** 3, 3), F32)
    s[:, 2] = (idx % N).astype(F32)
    q = (idx.astype(F32) / fN).astype(F32)
    s[:, 1] = np.fmod(q, fN)
    s[:, 0] = np.fmod((q / fN).astype(F32), fN)
    s[:, 0] = (s[:, 0] * F32(voxel_size)).astype(F32) + F32(origin[2])
    s[:, 1] = (s[:, 1] * F32(voxel_size)).astype(F32) + F32(origin[1])
    s[:, 2] = (s[:, 2] * F32(voxel_size)).astype(F32) + F32(origin[0])
    return s[None], origin, voxel_size


def query_field(p: dict, xyz, dirs=None, sigma_only=False):
    xyz = np.asarray(xyz, F32).reshape(-1, 3)
    dirs = np.zeros_like(xyz) if dirs is None else np.asarray(dirs, F32).reshape(-1, 3)
    x = np.concatenate([embed(xyz, 10), embed(dirs, 4)], -1)
    return nerf_forward(p, x[:, :63] if sigma_only else x, sigma_only=sigma_only)


def pack_vol(rgbsigma, N, extent):
    rgbsigma = np.asarray(rgbsigma, F32).reshape(-1, 4)
    sigma = np.maximum(rgbsigma[:, -1], 0)
    a = 1 - np.exp(-(extent) / N * sigma)
    a = a.flatten()
    rgb = (rgbsigma[:, :3] * 255).astype(np.uint32)
    i = np.where(a > 0)[0]
    rgb = rgb[i]
    a = a[i]
    s = rgb.dot(np.array([1 << 24, 1 << 16, 1 << 8])) + (a * 255).astype(np.uint32)
    return np.stack([i, s], -1).astype(np.uint32).flatten()


# ----------------------------------------------------------------------------
# Arithmetic of the opt-in split-bf16 kernels (csrc/bf16x3_core.h), restated for CPU tests
# ----------------------------------------------------------------------------
def bf16_round(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 (v_cvt_pk_bf16_f32)."""
    b = np.asarray(x, F32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(F32)


def split3(x):
    """x = p0 + p1 + p2 with three bf16 terms (each residual is exact in fp32)."""
    x = np.asarray(x, F32)
    p0 = bf16_round(x)
    r1 = (x - p0).astype(F32)
    p1 = bf16_round(r1)
    r2 = (r1 - p1).astype(F32)
    return p0, p1, bf16_round(r2)


# ----------------------------------------------------------------------------
# perf-mode random draws: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
# SC'11; the Random123 known-answer vectors are in tests/test_oracle_golden.py) as csrc/rays.hip render_draws_kernel
# uses it.  Not part of the reference (which draws from torch's mt19937 on CPU): parity tests inject draws instead.
# ----------------------------------------------------------------------------
def philox4x32_10(counter, key):
    """counter (..., 4) uint32, key (..., 2) uint32 -> (..., 4) uint32."""
    c = [np.asarray(counter[..., i], np.uint64) for i in range(4)]
    k = [np.asarray(key[..., i], np.uint64) for i in range(2)]
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k[0]) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k[1]) & m32, p0 & m32]
        k = [(k[0] + np.uint64(0x9E3779B9)) & m32, (k[1] + np.uint64(0xBB67AE85)) & m32]
    return np.stack(c, -1).astype(np.uint32)


def render_draws(seed: int, offset: int, sizes):
    """The four segments of nerfmi_render_draws: sizes = (n_perturb, n_noise_coarse, n_u, n_noise_fine) floats;
    segments 0, 2 uniform [0,1), segments 1, 3 N(0,1) by Box-Muller.  Returns four fp32 arrays."""
    out = []
    for seg, n in enumerate(sizes):
        q = (n + 3) // 4
        i = np.arange(q, dtype=np.uint64)
        ctr = np.stack([i & np.uint64(0xFFFFFFFF), np.uint64(seg) | ((i >> np.uint64(32)) << np.uint64(2)),
                        np.full(q, offset & 0xFFFFFFFF, np.uint64), np.full(q, (offset >> 32) & 0xFFFFFFFF, np.uint64)], -1)
        key = np.stack([np.full(q, seed & 0xFFFFFFFF, np.uint64), np.full(q, (seed >> 32) & 0xFFFFFFFF, np.uint64)], -1)
        x = philox4x32_10(ctr.astype(np.uint32), key.astype(np.uint32))
        if seg & 1:
            u1 = ((x[:, 0::2] >> 8).astype(F32) + F32(1)) * F32(2.0 ** -24)
            u2 = (x[:, 1::2] >> 8).astype(F32) * F32(2.0 ** -24)
            r = np.sqrt(F32(-2.0) * np.log(u1)).astype(F32)
            a = (F32(6.283185307179586) * u2).astype(F32)
            v = np.stack([r * np.cos(a), r * np.sin(a)], -1).reshape(q, 4).astype(F32)
        else:
            v = ((x >> 8).astype(F32) * F32(2.0 ** -24)).astype(F32)
        out.append(v.reshape(-1)[:n])
    return out
