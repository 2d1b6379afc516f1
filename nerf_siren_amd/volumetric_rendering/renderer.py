"""volumetric_rendering/renderer.py:23-256 (ImportanceRenderer and helpers) on the gfx950 kernels.

Training: when grad is enabled and the planes or the decoder require grad, the whole forward is ONE autograd
node (`_RenderFn`) whose backward is marcher(fine) -> unify^-1 -> marcher(coarse) -> decoder/tri-plane scatter
(csrc/eg3d_bwd.hip); depths carry no gradient, as in the reference (renderer.py:201 no_grad).
rendering_options may carry two extra keys to inject the reference's random draws (parity tests):
'rng_stratified' (N,M,S,1) = the rand_like of sample_stratified, 'rng_importance' (N*M,F) = the rand of sample_pdf;
'aux' (a dict) receives the sampled depths of an inference call.
"""
import weakref

import torch

from .. import eg3d_ops
from .ray_marcher import MipRayMarcher2
from . import math_utils


# the reference calls a bare exit() here (renderer.py:139), which would take the caller's process down: an exception instead
_NO_IMPORTANCE = ("ImportanceRenderer needs depth_resolution_importance > 0 (volumetric_rendering/renderer.py:122-139 has "
                  "no coarse-only return)")


def generate_planes():
    """renderer.py:23-37."""
    return torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                         [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                         [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], dtype=torch.float32)


def project_onto_planes(planes, coordinates):
    """renderer.py:39-53: (N,M,3) -> (N*n_planes, M, 2).  With the reference's plane axes the projection is a
    coordinate selection: (x,y), (x,z), (z,x) -- which is what the fused kernel applies."""
    x, y, z = coordinates[..., 0], coordinates[..., 1], coordinates[..., 2]
    sel = torch.stack([torch.stack([x, y], -1), torch.stack([x, z], -1), torch.stack([z, x], -1)], 1)
    return sel.reshape(coordinates.shape[0] * 3, coordinates.shape[1], 2)


def sample_from_planes(plane_axes, plane_features, coordinates, mode='bilinear', padding_mode='zeros', box_warp=None):
    """renderer.py:55-65: plane_features (N,3,C,H,W), coordinates (N,M,3) -> (N,3,M,C)."""
    assert padding_mode == 'zeros'
    assert mode == 'bilinear'
    n = plane_features.shape[0]
    return eg3d_ops.sample_planes(eg3d_ops.pack_planes(plane_features), n, coordinates, box_warp)


class _RenderFn(torch.autograd.Function):
    """ImportanceRenderer.forward (fixed or 'auto' ray limits resolved by the caller) as one autograd node."""

    @staticmethod
    def forward(ctx, ren, decoder, opts, depths_coarse, u, ray_o, ray_d, planes, w0, b0, w1, b1):
        N, M, _ = ray_o.shape
        S = depths_coarse.shape[1]
        F = opts['depth_resolution_importance']
        wb = opts.get('white_back', False)
        planes_hwc = ren._pack(planes)
        dec = decoder.packed()
        # density_noise (renderer.py:149-150): sigma += randn_like(sigma) * density_noise behind both run_model calls -- additive,
        # so the backward is unchanged (d sigma passes through); opts['rng_density_noise'] = (coarse, fine) injects the draws
        dn = float(opts.get('density_noise', 0) or 0)
        inj = opts.get('rng_density_noise')
        colors_c, dens_c = eg3d_ops.run_model_rays(planes_hwc, N, dec, ray_o, ray_d, depths_coarse, opts['box_warp'])
        if dn > 0:
            dens_c = dens_c + (torch.randn_like(dens_c) if inj is None else inj[0].reshape(dens_c.shape).to(dens_c)) * dn
        cc, sc = colors_c.reshape(N * M, S, 3), dens_c.reshape(N * M, S)
        mm_c = eg3d_ops.minmax(depths_coarse)
        rgb_c, depth_c, w_c, wsum_c = eg3d_ops.march(cc, sc, depths_coarse, wb, mm_c)
        depths_fine = eg3d_ops.sample_importance(depths_coarse, w_c, u, F)
        colors_f, dens_f = eg3d_ops.run_model_rays(planes_hwc, N, dec, ray_o, ray_d, depths_fine, opts['box_warp'])
        if dn > 0:
            dens_f = dens_f + (torch.randn_like(dens_f) if inj is None else inj[1].reshape(dens_f.shape).to(dens_f)) * dn
        cf, sf = colors_f.reshape(N * M, F, 3), dens_f.reshape(N * M, F)
        all_d, all_c, all_s, idx = eg3d_ops.unify(depths_coarse, cc, sc, depths_fine, cf, sf, want_idx=True)
        mm_f = eg3d_ops.minmax(all_d)
        rgb_f, depth_f, _, wsum_f = eg3d_ops.march(all_c, all_s, all_d, wb, mm_f)
        ctx.save_for_backward(planes_hwc, dec, ray_o, ray_d, depths_coarse, depths_fine, cc, sc, all_d, all_c, all_s, idx,
                              mm_c, mm_f)
        ctx.cfg = (N, M, S, F, bool(wb), float(opts['box_warp']), float(decoder.lr_mul), tuple(planes.shape))
        v = lambda t, c: t.view(N, M, c)
        return v(rgb_c, 3), v(depth_c, 1), v(wsum_c, 1), v(rgb_f, 3), v(depth_f, 1), v(wsum_f, 1)

    @staticmethod
    def backward(ctx, g_rgb_c, g_depth_c, g_w_c, g_rgb_f, g_depth_f, g_w_f):
        (planes_hwc, dec, ray_o, ray_d, depths_c, depths_f, cc, sc, all_d, all_c, all_s, idx, mm_c, mm_f) = ctx.saved_tensors
        N, M, S, F, wb, box_warp, lr_mul, pshape = ctx.cfg
        z = lambda g: None if g is None else g.contiguous()
        # fine march -> inverse permutation -> coarse march (accumulating into the coarse slots)
        d_all_c, d_all_s = eg3d_ops.march_backward(all_c, all_s, all_d, mm_f, z(g_rgb_f), z(g_depth_f), z(g_w_f), wb)
        d_cc, d_sc, d_cf, d_sf = eg3d_ops.unify_backward(idx, d_all_c, d_all_s, S, F)
        eg3d_ops.march_backward(cc, sc, depths_c, mm_c, z(g_rgb_c), z(g_depth_c), z(g_w_c), wb, into=(d_cc, d_sc))
        gplanes_hwc = torch.zeros_like(planes_hwc)
        grads = [torch.empty(64, 32, device=dec.device), torch.empty(64, device=dec.device),
                 torch.empty(4, 64, device=dec.device), torch.empty(4, device=dec.device)]
        aux, npts = eg3d_ops.run_model_rays_backward(planes_hwc, N, dec, ray_o, ray_d, depths_c, box_warp, d_cc, d_sc,
                                                     gplanes_hwc)
        eg3d_ops.decoder_wgrad(aux, npts, lr_mul, grads, accumulate=False)
        aux, npts = eg3d_ops.run_model_rays_backward(planes_hwc, N, dec, ray_o, ray_d, depths_f, box_warp, d_cf, d_sf,
                                                     gplanes_hwc)
        eg3d_ops.decoder_wgrad(aux, npts, lr_mul, grads, accumulate=True)
        g_planes = eg3d_ops.unpack_planes(gplanes_hwc, N).view(pshape)
        return (None, None, None, None, None, None, None, g_planes, *grads)


class ImportanceRenderer(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ray_marcher = MipRayMarcher2()
        self.plane_axes = generate_planes()
        self._planes_src = None         # weak reference to the tensor OBJECT the cached image was packed from
        self._planes_ver = None
        self._planes_hwc = None

    def _pack(self, planes):
        """Channels-last image of `planes`, cached only while the SAME tensor object (weak reference) is passed again at
        the same version.  Never keyed on the address: in the reference planes is a fresh backbone.synthesis() output on
        every call (eg3d_training/triplane.py:61-65) and the caching allocator hands a freed activation's address --
        with _version 0 -- straight back to the next one."""
        src = self._planes_src() if self._planes_src is not None else None
        if src is not planes or self._planes_ver != planes._version or self._planes_hwc is None:
            self._planes_hwc = eg3d_ops.pack_planes(planes)
            self._planes_src = weakref.ref(planes)
            self._planes_ver = planes._version
        return self._planes_hwc

    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options):
        """-> rgb_coarse (N,M,3), depth_coarse (N,M,1), weights_coarse.sum(2) (N,M,1), rgb_final, depth_final,
        weights.sum(2)  (renderer.py:88-142)."""
        train = torch.is_grad_enabled() and (planes.requires_grad or any(p.requires_grad for p in decoder.parameters()))
        opts = rendering_options
        N, M, _ = ray_origins.shape
        S = opts['depth_resolution']
        dev = ray_origins.device
        # The two uniform draws of a forward (rand_like of renderer.py:172-195, rand of :230): the injected tensors (parity
        # tests) or ONE Philox key per call (seed / offset of torch's CUDA generator, which the call advances), from which the
        # sampling kernels draw in place -- segments 0 and 2 of that stream; no aten distribution launch on the path.
        key = None
        if opts.get('rng_stratified') is None or opts.get('rng_importance') is None:
            from .. import ops as _ops
            key = _ops.next_draw_key(dev)
        rs = opts.get('rng_stratified')
        rs = key if rs is None else rs.reshape(N * M, S)
        if opts['ray_start'] == opts['ray_end'] == 'auto':
            ray_start, ray_end = math_utils.get_ray_limits_box(ray_origins, ray_directions, box_side_length=opts['box_warp'])
            is_ray_valid = ray_end > ray_start
            if torch.any(is_ray_valid).item():                                   # renderer.py:93-95
                ray_start[~is_ray_valid] = ray_start[is_ray_valid].min()
                ray_end[~is_ray_valid] = ray_start[is_ray_valid].max()
            depths_coarse = eg3d_ops.sample_stratified(N * M, S, rs, ray_start, ray_end, device=dev)
        else:
            depths_coarse = eg3d_ops.sample_stratified(N * M, S, rs, opts['ray_start'], opts['ray_end'],
                                                       opts.get('disparity_space_sampling', False), device=dev)
        o, d = ray_origins.detach().contiguous(), ray_directions.detach().contiguous()
        if train:
            F_ = opts['depth_resolution_importance']
            if F_ <= 0:
                raise ValueError(_NO_IMPORTANCE)
            u_ = opts.get('rng_importance')
            u_ = key if u_ is None else u_.reshape(N * M, F_).contiguous()
            net = decoder.net
            return _RenderFn.apply(self, decoder, opts, depths_coarse, u_, o, d, planes, net[0].weight,
                                   net[0].bias, net[2].weight, net[2].bias)
        planes_hwc = self._pack(planes)
        dec = decoder.packed()
        wb = opts.get('white_back', False)

        colors_coarse, dens_coarse = eg3d_ops.run_model_rays(planes_hwc, N, dec, o, d, depths_coarse, opts['box_warp'])
        inj = opts.get('rng_density_noise')                                      # (coarse, fine) draws injected by parity tests
        if opts.get('density_noise', 0) > 0:                                     # renderer.py:149-150
            dens_coarse = dens_coarse + (torch.randn_like(dens_coarse) if inj is None
                                         else inj[0].reshape(dens_coarse.shape).to(dens_coarse)) * opts['density_noise']
        cc = colors_coarse.reshape(N * M, S, 3)
        sc = dens_coarse.reshape(N * M, S)
        rgb_coarse, depth_coarse, weights_coarse, wsum_coarse = eg3d_ops.march(cc, sc, depths_coarse, wb)

        F = opts['depth_resolution_importance']
        if F <= 0:
            raise ValueError(_NO_IMPORTANCE)
        u = opts.get('rng_importance')
        u = key if u is None else u.reshape(N * M, F)
        depths_fine = eg3d_ops.sample_importance(depths_coarse, weights_coarse, u, F)
        colors_fine, dens_fine = eg3d_ops.run_model_rays(planes_hwc, N, dec, o, d, depths_fine, opts['box_warp'])
        if opts.get('density_noise', 0) > 0:
            dens_fine = dens_fine + (torch.randn_like(dens_fine) if inj is None
                                     else inj[1].reshape(dens_fine.shape).to(dens_fine)) * opts['density_noise']
        all_d, all_c, all_s = eg3d_ops.unify(depths_coarse, cc, sc, depths_fine, colors_fine.reshape(N * M, F, 3),
                                             dens_fine.reshape(N * M, F))
        rgb_final, depth_final, _, wsum = eg3d_ops.march(all_c, all_s, all_d, wb)
        if isinstance(opts.get('aux'), dict):        # test hook: the sampled depths (parity tests condition on them)
            opts['aux'].update(depths_coarse=depths_coarse, depths_fine=depths_fine, all_depths=all_d)
        v = lambda t, c: t.view(N, M, c)
        return (v(rgb_coarse, 3), v(depth_coarse, 1), v(wsum_coarse, 1), v(rgb_final, 3), v(depth_final, 1), v(wsum, 1))

    def run_model(self, planes, decoder, sample_coordinates, sample_directions, options):
        """renderer.py:144-151 -> {'rgb': (N,M,3), 'sigma': (N,M,1)}."""
        n = planes.shape[0]
        rgb, sigma = eg3d_ops.run_model(self._pack(planes), n, decoder.packed(), sample_coordinates, options['box_warp'])
        if options.get('density_noise', 0) > 0:
            sigma = sigma + torch.randn_like(sigma) * options['density_noise']
        return {'rgb': rgb, 'sigma': sigma}

    def unify_samples(self, depths1, colors1, densities1, depths2, colors2, densities2):
        """renderer.py:160-170; (N,M,S,.) tensors."""
        n, m = depths1.shape[0], depths1.shape[1]
        s1, s2 = depths1.shape[2], depths2.shape[2]
        d, c, s = eg3d_ops.unify(depths1.reshape(n * m, s1), colors1.reshape(n * m, s1, 3), densities1.reshape(n * m, s1),
                                 depths2.reshape(n * m, s2), colors2.reshape(n * m, s2, 3), densities2.reshape(n * m, s2))
        return d.view(n, m, s1 + s2, 1), c.view(n, m, s1 + s2, 3), s.view(n, m, s1 + s2, 1)

    sort_samples = None  # the reference's sort_samples (renderer.py:153-158) has no caller

    def sample_stratified(self, ray_origins, ray_start, ray_end, depth_resolution, disparity_space_sampling=False,
                          rand=None):
        """renderer.py:172-195 -> (N,M,S,1)."""
        N, M, _ = ray_origins.shape
        if rand is None:
            from .. import ops as _ops
            rand = _ops.next_draw_key(ray_origins.device)
        return eg3d_ops.sample_stratified(N * M, depth_resolution, rand, ray_start, ray_end, disparity_space_sampling,
                                          device=ray_origins.device).view(N, M, depth_resolution, 1)

    def sample_importance(self, z_vals, weights, N_importance, u=None):
        """renderer.py:197-215 -> (N,M,N_importance,1)."""
        n, m, s, _ = z_vals.shape
        if u is None:
            from .. import ops as _ops
            u = _ops.next_draw_key(z_vals.device)
        out = eg3d_ops.sample_importance(z_vals.reshape(n * m, s), weights.reshape(n * m, s - 1), u, N_importance)
        return out.view(n, m, N_importance, 1)
