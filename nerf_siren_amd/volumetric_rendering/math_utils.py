"""volumetric_rendering/math_utils.py:46-118 on the GPU kernels."""
import torch

from .. import eg3d_ops


def get_ray_limits_box(rays_o: torch.Tensor, rays_d: torch.Tensor, box_side_length):
    """Ray / axis-aligned box slab test (math_utils.py:46-98): (...,3),(...,3) -> (...,1),(...,1); misses (-1,-2)."""
    return eg3d_ops.ray_limits_box(rays_o, rays_d, box_side_length)


def linspace(start: torch.Tensor, stop: torch.Tensor, num: int):
    """math_utils.py:101-118: [num, *start.shape], start + i/(num-1) * (stop - start).
    (A 3-op elementwise helper: the stratified sampler fuses it, this form is kept for API parity.)"""
    r = start.numel()
    zeros = torch.zeros((r, num), device=start.device, dtype=torch.float32)
    out = eg3d_ops.sample_stratified(r, num, zeros, start.reshape(r), stop.reshape(r))
    return out.t().reshape(num, *start.shape)
