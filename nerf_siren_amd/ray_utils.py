"""Device-side counterpart of datasets/ray_utils.py: same names and argument meaning, tensors
live on the GPU, no kornia.  `generate_rays` is the fused form a data loader should call: (poses, focal, pixel
indices) -> the (N, 8) [o, d, near, far] rows render_rays consumes, with no host ray buffer and no per-step H2D copy."""
from __future__ import annotations

import torch

from . import ops


def get_ray_directions(H, W, focal, device=None):
    """ray_utils.py:5-24 -> directions (H, W, 3) in camera coordinates."""
    return ops.ray_directions(H, W, focal, device or torch.device("cuda", torch.cuda.current_device()))


def get_rays(directions, c2w):
    """ray_utils.py:27-50 -> rays_o (H*W, 3), rays_d (H*W, 3) normalised, world coordinates."""
    return ops.get_rays(directions, c2w)


def get_ndc_rays(H, W, focal, near, rays_o, rays_d):
    """ray_utils.py:53-93."""
    if isinstance(near, torch.Tensor):
        raise NotImplementedError("per-ray near planes are not supported (the reference's loaders pass the float 1.0)")
    return ops.get_ndc_rays(H, W, focal, near, rays_o, rays_d)


def generate_rays(c2w, H, W, focal, pixel_index=None, ndc=False, near=2.0, far=6.0):
    """Fused get_ray_directions + get_rays (+ get_ndc_rays) + [o, d, near, far] packing (blender.py:60-69,
    llff.py:234-250).  c2w (n_images, 3, 4); pixel_index int64 = image*H*W + row*W + column (None: every pixel)."""
    return ops.generate_rays(c2w, H, W, focal, pixel_index, ndc, near, far)
