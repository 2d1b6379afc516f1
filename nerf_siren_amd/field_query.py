"""Dense field queries: the same MLP kernel without ray structure.

    query_field   NeRF()(cat[Embedding(xyz), Embedding(dir)]) for arbitrary points -- the chunk loop of
                  extract_color_mesh.py:128-137 / extract_mesh.ipynb cell 4 as ONE launch per chunk: a point is a
                  "ray" with origin xyz, direction dir and a single sample at depth 0 (xyz + dir*0 == xyz exactly),
                  so the fused embed+MLP kernel runs unchanged and reads 36 B per point instead of a 360 B
                  pre-embedded row
    sigma_grid    the N^3 occupancy grid of extract_color_mesh.py:117-140 (np.meshgrid 'xy' ordering, zero
                  directions, sigma clamped at 0)
    pack_vol      the sparse `.vol` records of extract_mesh.ipynb cell 7 (uint32 pairs [voxel index, r<<24|g<<16|b<<8|a])
    create_samples / eg3d_sigma_grid
                  the EG3D "neural volume" of BASELINE configs[4]: the N^3 sample grid of
                  extract_color_mesh_eg3d.py:72-94 and the slab-wise run_model query + flip of :177-195

marching cubes / mesh colouring (PyMCubes, open3d) stay on the host and out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def query_field(model, xyz, dirs=None, sigma_only=False, chunk=1 << 22):
    """xyz (B,3) fp32 on the GPU, dirs (B,3) or None (= zeros, 'sigma is independent of direction',
    extract_color_mesh.py:124) -> (B,4) [rgb, sigma] or (B,1) sigma."""
    from .rendering import _field_infer          # honours set_math('bf16x3')
    xyz = xyz.reshape(-1, 3).float().contiguous()
    B = xyz.shape[0]
    outs = []
    for i in range(0, B, chunk):
        x = xyz[i:i + chunk]
        n = x.shape[0]
        rays = torch.zeros((n, 8), device=x.device, dtype=torch.float32)
        rays[:, 0:3] = x
        if dirs is not None:
            rays[:, 3:6] = dirs.reshape(-1, 3)[i:i + chunk]
        z = torch.zeros((n, 1), device=x.device, dtype=torch.float32)
        out = _field_infer(model, rays, z, sigma_only)
        outs.append(out.reshape(n, -1))
    return torch.cat(outs, 0) if len(outs) != 1 else outs[0]


def grid_points(N, x_range, y_range, z_range, device):
    """xyz_ of extract_color_mesh.py:117-122: np.linspace in fp64, np.meshgrid (default 'xy' indexing), stacked and
    rounded to fp32 -> (N^3, 3); flat index = (iy*N + ix)*N + iz."""
    x = torch.linspace(x_range[0], x_range[1], N, dtype=torch.float64, device=device)
    y = torch.linspace(y_range[0], y_range[1], N, dtype=torch.float64, device=device)
    z = torch.linspace(z_range[0], z_range[1], N, dtype=torch.float64, device=device)
    gy, gx, gz = torch.meshgrid(y, x, z, indexing="ij")          # == np.meshgrid(x, y, z) ('xy')
    return torch.stack([gx, gy, gz], -1).reshape(-1, 3).float()


def sigma_grid(model, N, x_range, y_range, z_range, chunk=1 << 22, return_rgbsigma=False):
    """(N,N,N) occupancy grid max(sigma, 0) (extract_color_mesh.py:139-140), on the GPU."""
    dev = next(model.parameters()).device
    pts = grid_points(N, x_range, y_range, z_range, dev)
    with torch.no_grad():
        rgbsigma = query_field(model, pts, None, sigma_only=not return_rgbsigma, chunk=chunk)
    sigma = torch.clamp_min(rgbsigma[:, -1], 0).reshape(N, N, N)
    return (sigma, rgbsigma) if return_rgbsigma else sigma


def pack_vol(rgbsigma, N, extent) -> np.ndarray:
    """extract_mesh.ipynb cell 7: a = 1 - exp(-extent/N * relu(sigma)); keep a > 0; records uint32
    [index, (r<<24)|(g<<16)|(b<<8)|a] with r,g,b = trunc(rgb*255), a = trunc(a*255).  Returns the flat uint32
    array whose bytes are the `.vol` file (extent = xmax - xmin)."""
    rgbsigma = rgbsigma.reshape(-1, 4).float()
    sigma = torch.clamp_min(rgbsigma[:, 3], 0)
    a = 1 - torch.exp(sigma * (-float(extent) / N))                  # fp32, like numpy on a float32 array
    idx = torch.nonzero(a > 0).reshape(-1)
    rgb = (rgbsigma[idx, :3] * 255).to(torch.int64)                  # astype(uint32) of non-negative values
    s = (rgb[:, 0] << 24) + (rgb[:, 1] << 16) + (rgb[:, 2] << 8) + (a[idx] * 255).to(torch.int64)
    res = torch.stack([idx, s], -1).reshape(-1)
    return (res & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)


def create_samples(N=256, voxel_origin=(0, 0, 0), cube_length=2.0, device=None):
    """extract_color_mesh_eg3d.py:72-94 on the device -> (samples (1, N^3, 3), voxel_origin (3,) float64 numpy,
    voxel_size).  As in the reference the y and x columns come from FLOAT divisions of the flat index
    ((idx.float() / N) % N, ((idx.float() / N) / N) % N), not from integer voxel indices."""
    origin = np.array(voxel_origin, np.float64) - cube_length / 2
    voxel_size = cube_length / (N - 1)
    s = torch.empty((N ** 3, 3), device=device, dtype=torch.float32)
    ops.check(ops._lib.lib().nerfmi_create_samples(int(N), float(origin[0]), float(origin[1]), float(origin[2]),
                                                   float(voxel_size), ops.ptr(s), ops._stream(s)), "create_samples")
    return s.unsqueeze(0), origin, voxel_size


def eg3d_sigma_grid(renderer, planes, decoder, options, N=256, cube_length=3.0, max_batch=1000000):
    """The query loop of extract_color_mesh_eg3d.py:177-195: create_samples(N, [0,0,0], cube_length) -> run_model in
    slabs of max_batch points -> sigma reshaped (N,N,N) and flipped along axis 0.  planes (1,3,32,H,W)."""
    samples, _, _ = create_samples(N, (0, 0, 0), cube_length, device=planes.device)
    sigmas = torch.empty((1, samples.shape[1], 1), device=planes.device)
    with torch.no_grad():
        for head in range(0, samples.shape[1], max_batch):
            sigmas[:, head:head + max_batch] = renderer.run_model(planes, decoder, samples[:, head:head + max_batch], None,
                                                                  options)['sigma']
    return torch.flip(sigmas.reshape(N, N, N), (0,))
