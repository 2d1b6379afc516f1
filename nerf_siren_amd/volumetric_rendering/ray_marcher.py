"""volumetric_rendering/ray_marcher.py:20-62 (MipRayMarcher2)."""
import torch

from .. import eg3d_ops


class MipRayMarcher2(torch.nn.Module):
    def __init__(self):
        super().__init__()

    def run_forward(self, colors, densities, depths, rendering_options):
        """colors (N,M,S,3), densities (N,M,S,1), depths (N,M,S,1) ->
        composite_rgb (N,M,3), composite_depth (N,M,1), weights (N,M,S-1,1)."""
        if rendering_options['clamp_mode'] != 'softplus':
            assert False, "MipRayMarcher only supports `clamp_mode`=`softplus`!"       # ray_marcher.py:35
        n, m, s = colors.shape[0], colors.shape[1], colors.shape[2]
        rgb, depth, w, _ = eg3d_ops.march(colors.reshape(n * m, s, 3), densities.reshape(n * m, s),
                                          depths.reshape(n * m, s), rendering_options.get('white_back', False))
        return rgb.view(n, m, 3), depth.view(n, m, 1), w.view(n, m, s - 1, 1)

    def forward(self, colors, densities, depths, rendering_options):
        composite_rgb, composite_depth, weights = self.run_forward(colors, densities, depths, rendering_options)
        return composite_rgb, composite_depth, weights
