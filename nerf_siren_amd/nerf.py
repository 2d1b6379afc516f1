"""Field modules with the reference's API and state_dict layout (models/nerf.py).

`Embedding` and `NeRF` keep the constructor signatures, attribute names and
parameter names of models/nerf.py:4-38 and :41-124, so checkpoints written by the
reference load unchanged (utils/__init__.py:56-86 keys `xyz_encoding_1.0.weight`
...).  The math runs in the HIP kernels of libnerfmi.so; torch.nn only stores the
parameters.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class Embedding(nn.Module):
    """Embedding(in_channels, N_freqs, logscale=True) -- models/nerf.py:4-38.
    x (B,3) -> (B, 3*(2*N_freqs+1)) = [x, sin(2^k x), cos(2^k x)]_k."""

    def __init__(self, in_channels, N_freqs, logscale=True):
        super().__init__()
        if in_channels != 3 or not logscale:
            raise NotImplementedError("the gfx950 kernels implement the reference's configuration: "
                                      "in_channels=3, logscale=True (system.py:181-182)")
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.out_channels = in_channels * (2 * N_freqs + 1)
        self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs)

    def forward(self, x):
        return ops.embed(x, self.N_freqs)


class NeRF(nn.Module):
    """NeRF(D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4]) --
    models/nerf.py:41-124.  forward(x, sigma_only=False): x (B,90)|(B,63) ->
    (B,4) [rgb, sigma] | (B,1)."""

    def __init__(self, D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4]):
        super().__init__()
        if (D, W, in_channels_xyz, in_channels_dir, list(skips)) != (8, 256, 63, 27, [4]):
            raise NotImplementedError("the gfx950 MLP kernel is specialised for the reference's only "
                                      "configuration: D=8, W=256, 63+27 inputs, skips=[4]")
        self.D, self.W = D, W
        self.in_channels_xyz, self.in_channels_dir, self.skips = in_channels_xyz, in_channels_dir, skips
        for i in range(D):
            if i == 0:
                layer = nn.Linear(in_channels_xyz, W)
            elif i in skips:
                layer = nn.Linear(W + in_channels_xyz, W)
            else:
                layer = nn.Linear(W, W)
            setattr(self, f"xyz_encoding_{i+1}", nn.Sequential(layer, nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), nn.ReLU(True))
        self.sigma = nn.Linear(W, 1)
        self.rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
        self._packed = None
        self._packed_key = None
        self._fast = None

    # -- parameters in C-ABI order -------------------------------------------------
    def param_list(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in ops.PARAM_ORDER]

    def mark_parameters_changed(self):
        """Called by writers that change parameter memory without going through torch (training.FusedAdam)."""
        self._epoch = getattr(self, "_epoch", 0) + 1

    grad_numel = ops.PARAM_NUMEL

    @staticmethod
    def grad_views(flat):
        """The 24 gradient tensors as views of one flat buffer, in param_list() order."""
        return ops.flat_views(flat)

    def packed(self):
        """Fragment-order weight blob (csrc/mlp_layout.h), re-packed when any
        parameter changed (optimizer steps bump tensor._version)."""
        ps = self.param_list()
        key = (getattr(self, "_epoch", 0),) + tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or key != self._packed_key or self._packed.device != ps[0].device:
            self._packed = ops.nerf_pack(ps, self._packed if (self._packed is not None
                                                               and self._packed.device == ps[0].device) else None)
            self._packed_key = key
        return self._packed

    def packed_fast(self):
        """bf16x3 split image for the opt-in fast inference math (rendering.set_math('bf16x3'))."""
        pk = self.packed()
        if getattr(self, "_fast_key", None) is not self._packed_key or self._fast is None:
            self._fast = ops.nerf_pack_fast(pk)
            self._fast_key = self._packed_key
        return self._fast

    def forward(self, x, sigma_only=False):
        if torch.is_grad_enabled():
            if x.requires_grad:
                raise NotImplementedError("gradients w.r.t. the embedded inputs are not implemented (the embeddings have "
                                          "no parameters; the reference's training path never needs them)")
            if any(p.requires_grad for p in self.parameters()):
                from .rendering import EmbeddedField
                return EmbeddedField.apply(self, x, bool(sigma_only), *self.param_list())
        return ops.nerf_forward_embedded(self.packed(), x, sigma_only)


# ---------------------------------------------------------------------------------------------
# FiLM-SIREN field (models/nerf.py:126-216)
# ---------------------------------------------------------------------------------------------
def frequency_init(freq):
    """models/nerf.py:126-132."""
    def init(m):
        with torch.no_grad():
            if isinstance(m, nn.Linear):
                num_input = m.weight.size(-1)
                b = (6 / num_input) ** 0.5 / freq
                m.weight.uniform_(-b, b)
    return init


def first_layer_film_sine_init(m):
    """models/nerf.py:153-157."""
    with torch.no_grad():
        if isinstance(m, nn.Linear):
            num_input = m.weight.size(-1)
            m.weight.uniform_(-1 / num_input, 1 / num_input)


class UniformBoxWarp(nn.Module):
    """models/nerf.py:134-140 (applied inside the kernel; kept for attribute parity)."""

    def __init__(self, sidelength):
        super().__init__()
        self.scale_factor = 2 / sidelength


class FiLMLayer(nn.Module):
    """FiLMLayer(input_dim, hidden_dim) -- models/nerf.py:142-151: sin(freq * Linear(x) + phase_shift).
    Parameter holder: the arithmetic of all nine layers runs in one fused HIP kernel."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.layer = nn.Linear(input_dim, hidden_dim)


class SemanticNeRF(nn.Module):
    """SemanticNeRF(input_dim=2, z_dim=100, hidden_dim=256, output_dim=1, device=None) --
    models/nerf.py:159-216, the pi-GAN style FiLM-SIREN field ("TALLSIREN" + UniformBoxWarp(51)).

    Same parameter names/shapes/init as the reference (529 156 parameters).
    forward_with_frequencies_phase_shifts(input (Bz,Np,3), frequencies (Bz,9*256), phase_shifts (Bz,9*256),
    ray_directions (Bz,Np,3)) -> (Bz,Np,4) [rgb, sigma].
    """

    def __init__(self, input_dim=2, z_dim=100, hidden_dim=256, output_dim=1, device=None):
        super().__init__()
        if hidden_dim != 256:
            raise NotImplementedError("the gfx950 SIREN kernel is specialised for hidden_dim=256 (the reference default)")
        self.device = device
        self.input_dim, self.z_dim, self.hidden_dim, self.output_dim = input_dim, z_dim, hidden_dim, output_dim
        self.network = nn.ModuleList([FiLMLayer(3, hidden_dim)] + [FiLMLayer(hidden_dim, hidden_dim) for _ in range(7)])
        self.final_layer = nn.Linear(hidden_dim, 1)
        self.color_layer_sine = FiLMLayer(hidden_dim + 3, hidden_dim)
        self.color_layer_linear = nn.Sequential(nn.Linear(hidden_dim, 3))
        self.network.apply(frequency_init(25))
        self.final_layer.apply(frequency_init(25))
        self.color_layer_sine.apply(frequency_init(25))
        self.color_layer_linear.apply(frequency_init(25))
        self.network[0].apply(first_layer_film_sine_init)
        self.gridwarper = UniformBoxWarp(51)
        self._packed = None
        self._packed_key = None

    def param_list(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in ops.SIREN_PARAM_ORDER]

    def mark_parameters_changed(self):
        """Called by writers that change parameter memory without going through torch (training.FusedAdam)."""
        self._epoch = getattr(self, "_epoch", 0) + 1

    grad_numel = ops.SIREN_PARAM_NUMEL

    @staticmethod
    def grad_views(flat):
        """The 22 gradient tensors as views of one flat buffer, in param_list() order."""
        return ops.siren_flat_views(flat)

    def packed(self):
        ps = self.param_list()
        key = (getattr(self, "_epoch", 0),) + tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or key != self._packed_key or self._packed.device != ps[0].device:
            self._packed = ops.siren_pack(ps)
            self._packed_key = key
        return self._packed

    def packed_fast(self):
        """bf16x3 split image for the opt-in fast math (rendering.set_math('bf16x3'))."""
        pk = self.packed()
        if getattr(self, "_fast_key", None) is not self._packed_key or getattr(self, "_fast", None) is None:
            self._fast = ops.siren_pack_fast(pk)
            self._fast_key = self._packed_key
        return self._fast

    def forward(self, input, z, ray_directions, **kwargs):
        # the reference calls self.mapping_network, which it never defines (nerf.py:185 is commented out)
        raise AttributeError("'SemanticNeRF' object has no attribute 'mapping_network' (models/nerf.py:185, :198); "
                             "use forward_with_frequencies_phase_shifts")

    def forward_with_frequencies_phase_shifts(self, input, frequencies, phase_shifts, ray_directions, **kwargs):
        """nerf.py:201-216.  Differentiable w.r.t. the 22 parameters and w.r.t. the conditioning rows `frequencies` /
        `phase_shifts` (which the reference would get from its mapping network, nerf.py:185); gradients w.r.t. the points
        and the directions are not provided."""
        bz, npts = input.shape[0], input.shape[1]
        pts = input.reshape(-1, 3).contiguous()
        dirs = ray_directions.reshape(-1, 3).contiguous()
        freq, phase = frequencies.reshape(bz, -1).contiguous(), phase_shifts.reshape(bz, -1).contiguous()
        if torch.is_grad_enabled():
            if input.requires_grad or ray_directions.requires_grad:
                raise NotImplementedError("gradients w.r.t. the SIREN points / directions are not implemented")
            cond = freq.requires_grad or phase.requires_grad
            if cond and bz > 1:
                # the conditioning gradients come out of a launch that shares ONE row (csrc/siren_bwd.hip): one autograd
                # node per row; autograd sums the rows' parameter gradients
                from .rendering import SirenPoints
                pts3, dirs3 = pts.view(bz, npts, 3), dirs.view(bz, npts, 3)
                return torch.stack([SirenPoints.apply(self, pts3[r].contiguous(), dirs3[r].contiguous(), freq[r:r + 1],
                                                      phase[r:r + 1], npts, *self.param_list()) for r in range(bz)])
            if cond or any(p.requires_grad for p in self.parameters()):
                from .rendering import SirenPoints
                return SirenPoints.apply(self, pts, dirs, freq, phase, npts, *self.param_list()).view(bz, npts, 4)
        out = ops.siren_forward_points(self.packed(), pts, dirs, freq, phase, npts)
        return out.view(bz, npts, 4)


class SirenField(nn.Module):
    """Adapter that lets the FiLM-SIREN field be rendered by render_rays() (the
    reference's SemanticNeRF signature differs from NeRF.forward(x, sigma_only), so it cannot be passed to
    render_rays unchanged).  One conditioning row (frequencies, phase_shifts) for all rays."""

    def __init__(self, model: SemanticNeRF, frequencies=None, phase_shifts=None):
        super().__init__()
        self.model = model
        h = 9 * model.hidden_dim
        self.frequencies = nn.Parameter(torch.zeros(1, h) if frequencies is None else frequencies.reshape(1, h).clone(),
                                        requires_grad=False)
        self.phase_shifts = nn.Parameter(torch.zeros(1, h) if phase_shifts is None else phase_shifts.reshape(1, h).clone(),
                                         requires_grad=False)

    # the training protocol of rendering.FieldRender / training.FusedAdam / parallel.FlatGradAllReduce
    def param_list(self):
        return self.model.param_list()

    def cond_rows(self):
        """(2, 2304) [frequencies; phase_shifts] in one buffer (what nerfmi_render_rays_fused takes), rebuilt when either
        row changed."""
        key = (self.frequencies.data_ptr(), self.frequencies._version, self.phase_shifts.data_ptr(), self.phase_shifts._version)
        if getattr(self, "_cond_key", None) != key:
            self._cond_rows = torch.cat([self.frequencies.detach().reshape(1, -1), self.phase_shifts.detach().reshape(1, -1)]).contiguous()
            self._cond_key = key
        return self._cond_rows

    def cond_param_list(self):
        """[frequencies, phase_shifts] when either is trainable (`.requires_grad_(True)`), else []: render_rays then
        differentiates through the FiLM conditioning as the reference's autograd does (nerf.py:147-151)."""
        if self.frequencies.requires_grad or self.phase_shifts.requires_grad:
            return [self.frequencies, self.phase_shifts]
        return []

    def mark_parameters_changed(self):
        self.model.mark_parameters_changed()

    grad_numel = ops.SIREN_PARAM_NUMEL

    @staticmethod
    def grad_views(flat):
        return ops.siren_flat_views(flat)

    def field_rays(self, rays, z, sigma_only=False):
        from .rendering import get_math
        if get_math() == "bf16x3":
            return ops.siren_forward_rays_fast(self.model.packed(), self.model.packed_fast(), rays, z, self.frequencies,
                                               self.phase_shifts, rays.shape[0], sigma_only)
        return ops.siren_forward_rays(self.model.packed(), rays, z, self.frequencies, self.phase_shifts,
                                      rays.shape[0], sigma_only)
