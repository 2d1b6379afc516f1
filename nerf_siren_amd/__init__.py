"""nerf_siren_amd -- MI355X-native volumetric-rendering hot path of Freedomcls/nerf-siren.

Public API mirrors the reference: render_rays, sample_pdf (models/rendering.py),
Embedding, NeRF (models/nerf.py), searchsorted (torchsearchsorted).  All compute is
in libnerfmi.so (hand-written HIP for gfx950) behind the C ABI of include/nerfmi.h.
"""
__version__ = "0.1.0"


def __getattr__(name):
    # torch-dependent modules load lazily so that `nerf_siren_amd.synth` stays numpy-only
    if name in ("render_rays", "sample_pdf", "set_math", "get_math"):
        from . import rendering
        return getattr(rendering, name)
    if name in ("Embedding", "NeRF", "SemanticNeRF", "FiLMLayer", "SirenField"):
        from . import nerf
        return getattr(nerf, name)
    if name in ("OSGDecoder", "FullyConnectedLayer"):
        from . import eg3d
        return getattr(eg3d, name)
    if name in ("ImportanceRenderer", "MipRayMarcher2", "RaySampler"):
        from . import volumetric_rendering
        return getattr(volumetric_rendering, name)
    if name in ("FusedAdam", "FusedMSELoss"):
        from . import training
        return getattr(training, name)
    if name in ("get_ray_directions", "get_rays", "get_ndc_rays", "generate_rays"):
        from . import ray_utils
        return getattr(ray_utils, name)
    if name == "searchsorted":
        from . import ops
        return ops.searchsorted
    raise AttributeError(name)
