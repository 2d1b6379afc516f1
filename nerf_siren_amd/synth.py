"""Deterministic synthetic workloads (no dataset, no checkpoint exists offline).

Everything here is integer-hash based (splitmix64) so that the numbers are
identical in the build container, on the GPU box and in tools/make_golden.py,
independent of any library RNG stream.

Ray geometry follows the reference's Blender pipeline:
  datasets/ray_utils.py:5-24   get_ray_directions  [(i-W/2)/f, -(j-H/2)/f, -1]
  datasets/ray_utils.py:27-50  get_rays            d = dirs @ c2w[:, :3].T, normalised
  datasets/blender.py:36-37    near 2.0 / far 6.0,  :66-69 rays = [o, d, near, far]
with lego's public camera_angle_x = 0.6911112 (focal 555.555 at 400x400) and
cameras on the radius-4.0311 upper hemisphere.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_uniform(shape, seed: int) -> np.ndarray:
    """U[0,1) fp32 with 24 random bits per element, exact by construction."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
    bits = _splitmix64(idx) >> np.uint64(40)
    return (bits.astype(np.float32) / np.float32(1 << 24)).reshape(shape)


def hash_normal(shape, seed: int) -> np.ndarray:
    """N(0,1) fp32 via Box-Muller on two hash_uniform streams."""
    u1 = hash_uniform(shape, seed * 2 + 1).astype(np.float64)
    u2 = hash_uniform(shape, seed * 2 + 2).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


# ---------------------------------------------------------------------------
# NeRF parameters, state_dict layout of models/nerf.py:61-81
# ---------------------------------------------------------------------------
NERF_SHAPES = (
    [("xyz_encoding_1.0", 256, 63)]
    + [(f"xyz_encoding_{i}.0", 256, 256) for i in (2, 3, 4)]
    + [("xyz_encoding_5.0", 256, 319)]
    + [(f"xyz_encoding_{i}.0", 256, 256) for i in (6, 7, 8)]
    + [("xyz_encoding_final", 256, 256), ("dir_encoding.0", 128, 283),
       ("sigma", 1, 256), ("rgb.0", 3, 128)]
)


def nerf_params(seed: int = 0, structured: bool = True, sigma_bias: float = -1.0) -> dict:
    """nn.Linear-style U(+-1/sqrt(fan_in)) parameters for one NeRF().

    structured=True rescales the density head so that the field is not the
    near-degenerate default-init one (fixture-model caveat):
    both transparent and opaque rays, peaked and flat weight profiles occur.
    """
    p = {}
    for li, (name, fo, fi) in enumerate(NERF_SHAPES):
        bound = np.float32(1.0 / np.sqrt(fi))
        w = (hash_uniform((fo, fi), seed * 1000 + 2 * li) * 2 - 1) * bound
        b = (hash_uniform((fo,), seed * 1000 + 2 * li + 1) * 2 - 1) * bound
        p[name + ".weight"] = w.astype(np.float32)
        p[name + ".bias"] = b.astype(np.float32)
    if structured:
        for i in range(1, 9):
            p[f"xyz_encoding_{i}.0.weight"] = (p[f"xyz_encoding_{i}.0.weight"] * np.float32(1.6)).astype(np.float32)
        p["sigma.weight"] = (p["sigma.weight"] * np.float32(40.0)).astype(np.float32)
        p["sigma.bias"] = (p["sigma.bias"] * 0 + np.float32(sigma_bias)).astype(np.float32)
        p["rgb.0.weight"] = (p["rgb.0.weight"] * np.float32(12.0)).astype(np.float32)
    return p


# ---------------------------------------------------------------------------
# rays
# ---------------------------------------------------------------------------
LEGO_ANGLE_X = 0.6911112
LEGO_RADIUS = 4.0311


def _look_at_c2w(elev, azim, radius):
    """Camera-to-world (3,4), camera looks at the origin, -z forward, +y up."""
    pos = radius * np.array([np.cos(elev) * np.cos(azim), np.cos(elev) * np.sin(azim), np.sin(elev)])
    fwd = -pos / np.linalg.norm(pos)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    up2 = np.cross(right, fwd)
    c2w = np.stack([right, up2, -fwd, pos], 1)
    return c2w.astype(np.float32)


def blender_rays(n_rays: int, seed: int = 0, img_wh=(400, 400), n_views: int = 100) -> np.ndarray:
    """(n_rays, 8) fp32 rows [o, d, near, far], uniform random pixels over
    n_views lego-like cameras (unit-norm d, near 2 / far 6)."""
    W, H = img_wh
    focal = np.float32(0.5 * W / np.tan(0.5 * LEGO_ANGLE_X))
    uv = hash_uniform((n_views, 2), seed * 7919 + 11)
    elev = uv[:, 0].astype(np.float64) * np.deg2rad(60.0)
    azim = uv[:, 1].astype(np.float64) * 2 * np.pi
    c2ws = np.stack([_look_at_c2w(e, a, LEGO_RADIUS) for e, a in zip(elev, azim)])
    pick = hash_uniform((n_rays, 3), seed * 7919 + 13)
    v = np.minimum((pick[:, 0] * n_views).astype(np.int64), n_views - 1)
    i = np.minimum((pick[:, 1] * W).astype(np.int64), W - 1).astype(np.float32)
    j = np.minimum((pick[:, 2] * H).astype(np.int64), H - 1).astype(np.float32)
    dirs = np.stack([(i - np.float32(W / 2)) / focal, -(j - np.float32(H / 2)) / focal,
                     -np.ones_like(i)], -1).astype(np.float32)
    c2w = c2ws[v]
    d = np.einsum("nk,njk->nj", dirs, c2w[:, :, :3]).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    o = c2w[:, :, 3]
    near = np.full((n_rays, 1), 2.0, np.float32)
    far = np.full((n_rays, 1), 6.0, np.float32)
    return np.concatenate([o, d, near, far], -1).astype(np.float32)


def ndc_rays(n_rays: int, seed: int = 0, img_wh=(504, 378)) -> np.ndarray:
    """LLFF-style NDC rows (datasets/ray_utils.py:53-93): near 0 / far 1,
    directions NOT unit-norm. focal = 0.809*W (fern)."""
    W, H = img_wh
    focal = np.float32(0.809 * W)
    pick = hash_uniform((n_rays, 5), seed * 7919 + 17)
    i = np.minimum((pick[:, 0] * W).astype(np.int64), W - 1).astype(np.float32)
    j = np.minimum((pick[:, 1] * H).astype(np.int64), H - 1).astype(np.float32)
    d = np.stack([(i - np.float32(W / 2)) / focal, -(j - np.float32(H / 2)) / focal,
                  -np.ones_like(i)], -1).astype(np.float32)
    o = ((pick[:, 2:5] - 0.5) * np.float32(0.6)).astype(np.float32)
    near = np.float32(1.0)
    t = -(near + o[:, 2]) / d[:, 2]
    o = o + t[:, None] * d
    ox_oz = o[:, 0] / o[:, 2]
    oy_oz = o[:, 1] / o[:, 2]
    o0 = -1.0 / (W / (2.0 * focal)) * ox_oz
    o1 = -1.0 / (H / (2.0 * focal)) * oy_oz
    o2 = 1.0 + 2.0 * near / o[:, 2]
    d0 = -1.0 / (W / (2.0 * focal)) * (d[:, 0] / d[:, 2] - ox_oz)
    d1 = -1.0 / (H / (2.0 * focal)) * (d[:, 1] / d[:, 2] - oy_oz)
    d2 = 1 - o2
    rays = np.stack([o0, o1, o2, d0, d1, d2, np.zeros_like(o0), np.ones_like(o0)], -1)
    return rays.astype(np.float32)


# ---------------------------------------------------------------------------
# FiLM-SIREN parameters, state_dict layout of models/nerf.py:159-191 (SemanticNeRF)
# ---------------------------------------------------------------------------
SIREN_SHAPES = (
    [("network.0.layer", 256, 3)]
    + [(f"network.{i}.layer", 256, 256) for i in range(1, 8)]
    + [("final_layer", 1, 256), ("color_layer_sine.layer", 256, 259), ("color_layer_linear.0", 3, 256)]
)


def siren_params(seed: int = 0) -> dict:
    """Parameters distributed like SemanticNeRF's init (nerf.py:126-132, :153-157, :187-191):
    weights U(+-sqrt(6/in)/25) (first layer U(+-1/in)), biases nn.Linear default U(+-1/sqrt(in))."""
    p = {}
    for li, (name, fo, fi) in enumerate(SIREN_SHAPES):
        wb = (1.0 / fi) if li == 0 else (np.sqrt(6.0 / fi) / 25.0)
        w = (hash_uniform((fo, fi), seed * 1000 + 500 + 2 * li) * 2 - 1) * np.float32(wb)
        b = (hash_uniform((fo,), seed * 1000 + 501 + 2 * li) * 2 - 1) * np.float32(1.0 / np.sqrt(fi))
        p[name + ".weight"] = w.astype(np.float32)
        p[name + ".bias"] = b.astype(np.float32)
    return p


# ---------------------------------------------------------------------------
# EG3D tri-plane renderer inputs
# ---------------------------------------------------------------------------
def osg_params(seed: int = 0) -> dict:
    """OSGDecoder parameters (eg3d_training/triplane.py:144-153; FullyConnectedLayer init randn / lr_mul, bias 0
    -- biases made non-zero here so that they are exercised)."""
    return {"net.0.weight": hash_normal((64, 32), seed * 10 + 1), "net.0.bias": (hash_normal((64,), seed * 10 + 2) * 0.1).astype(np.float32),
            "net.2.weight": hash_normal((4, 64), seed * 10 + 3), "net.2.bias": (hash_normal((4,), seed * 10 + 4) * 0.1).astype(np.float32)}


def triplanes(seed: int = 0, res: int = 64, channels: int = 32, n: int = 1) -> np.ndarray:
    """(n, 3, channels, res, res) feature planes, N(0,1)."""
    return hash_normal((n, 3, channels, res, res), 7000 + seed)


EG3D_OPTIONS = dict(depth_resolution=64, depth_resolution_importance=64, ray_start=0.1, ray_end=10.0, box_warp=15.0,
                    white_back=False, clamp_mode="softplus", disparity_space_sampling=False)   # eg3d_renderer.py:30-36


def eg3d_rays(n_rays: int, seed: int = 0, radius: float = 2.7):
    """Rays from cameras on a sphere of avg_camera_radius 2.7 looking at the origin (eg3d_renderer.py:30)."""
    r = blender_rays(n_rays, seed)
    o = (r[:, 0:3] * np.float32(radius / LEGO_RADIUS)).astype(np.float32)
    return o, r[:, 3:6].copy()


# ---------------------------------------------------------------------------
# PSNR-parity protocol (tools/make_psnr_golden.py <-> tests): deterministic batches and random draws
# ---------------------------------------------------------------------------
def view_rays(res: int, view_seed: int) -> np.ndarray:
    """All res*res rays (res*res, 8) of one lego-like camera (same geometry as blender_rays), row-major pixels."""
    uv = hash_uniform((1, 2), view_seed * 7919 + 11)
    c2w = _look_at_c2w(float(uv[0, 0]) * np.deg2rad(60.0), float(uv[0, 1]) * 2 * np.pi, LEGO_RADIUS)
    focal = np.float32(0.5 * res / np.tan(0.5 * LEGO_ANGLE_X))
    j, i = np.meshgrid(np.arange(res, dtype=np.float32), np.arange(res, dtype=np.float32), indexing="ij")
    dirs = np.stack([(i - res / 2) / focal, -(j - res / 2) / focal, -np.ones_like(i)], -1).reshape(-1, 3).astype(np.float32)
    d = (dirs @ c2w[:, :3].T).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    o = np.broadcast_to(c2w[:, 3], d.shape)
    nf = np.tile(np.array([[2.0, 6.0]], np.float32), (d.shape[0], 1))
    return np.concatenate([o, d, nf], -1).astype(np.float32)


def psnr_rays(g) -> tuple:
    """(train rays, validation rays) of a PSNR fixture: stored (g15) or regenerated from the view seeds (g19)."""
    if "rays" in g:
        return g["rays"], g["val_rays"]
    rays = np.concatenate([view_rays(int(g["cfg_res"]), 300 + v) for v in range(int(g["cfg_n_train_views"]))], 0)
    return rays, view_rays(int(g["cfg_val_res"]), 399)


def psnr_batch_indices(step: int, n_total: int, batch: int) -> np.ndarray:
    return np.minimum((hash_uniform((batch,), 80000 + step) * n_total).astype(np.int64), n_total - 1)


def psnr_step_rng(step: int, batch: int, S: int, F: int) -> dict:
    return {"perturb_rand": hash_uniform((batch, S), 90000 + 4 * step), "u": hash_uniform((batch, F), 90001 + 4 * step)}
