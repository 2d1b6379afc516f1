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

    # -- parameters in C-ABI order -------------------------------------------------
    def param_list(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in ops.PARAM_ORDER]

    def packed(self):
        """Fragment-order weight blob (csrc/mlp_layout.h), re-packed when any
        parameter changed (optimizer steps bump tensor._version)."""
        ps = self.param_list()
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or key != self._packed_key or self._packed.device != ps[0].device:
            self._packed = ops.nerf_pack(ps, self._packed if (self._packed is not None
                                                               and self._packed.device == ps[0].device) else None)
            self._packed_key = key
        return self._packed

    def forward(self, x, sigma_only=False):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .rendering import EmbeddedField
            return EmbeddedField.apply(self, x, bool(sigma_only), *self.param_list())
        return ops.nerf_forward_embedded(self.packed(), x, sigma_only)
