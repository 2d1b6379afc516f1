"""volumetric_rendering/ray_sampler.py:24-63."""
import torch

from .. import eg3d_ops


class RaySampler(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ray_origins_h, self.ray_directions, self.depths, self.image_coords, self.rendering_options = \
            None, None, None, None, None

    def forward(self, cam2world_matrix, intrinsics, resolution):
        """cam2world_matrix (N,4,4), intrinsics (N,3,3) normalised to a unit image, resolution int ->
        ray_origins (N, res*res, 3), ray_dirs (N, res*res, 3) (x fastest, pixel centres)."""
        return eg3d_ops.ray_sampler(cam2world_matrix, intrinsics, resolution)
