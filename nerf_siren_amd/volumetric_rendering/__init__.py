"""EG3D tri-plane importance renderer with the reference's module layout
(volumetric_rendering/{renderer,ray_marcher,ray_sampler,math_utils}.py) on the gfx950 kernels."""
from .renderer import ImportanceRenderer, generate_planes, project_onto_planes, sample_from_planes  # noqa: F401
from .ray_marcher import MipRayMarcher2  # noqa: F401
from .ray_sampler import RaySampler  # noqa: F401
from . import math_utils  # noqa: F401
