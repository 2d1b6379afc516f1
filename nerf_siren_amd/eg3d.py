"""eg3d_training/triplane.py:144-167 (OSGDecoder) and the linear branch of
eg3d_training/networks_stylegan2.py:96-127 (FullyConnectedLayer) as parameter holders for the fused kernels."""
import numpy as np
import torch

from . import eg3d_ops


class FullyConnectedLayer(torch.nn.Module):
    """FullyConnectedLayer(in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0)."""

    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        if activation != 'linear' or not bias:
            raise NotImplementedError("only the linear+bias branch (torch.addmm, networks_stylegan2.py:122-123) is on the path")
        self.in_features, self.out_features, self.activation = in_features, out_features, activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init)))
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def extra_repr(self):
        return f'in_features={self.in_features:d}, out_features={self.out_features:d}, activation={self.activation:s}'


class OSGDecoder(torch.nn.Module):
    """OSGDecoder(n_features, options) with options = {'decoder_lr_mul', 'decoder_output_dim'} (triplane.py:144-153).
    forward(sampled_features (N,3,M,32), ray_directions) -> {'rgb': (N,M,3), 'sigma': (N,M,1)}."""

    def __init__(self, n_features, options):
        super().__init__()
        if n_features != 32 or options['decoder_output_dim'] != 3:
            raise NotImplementedError("the fused decoder is compiled for 32 features -> 64 hidden -> 1+3 outputs")
        self.hidden_dim = 64
        self.lr_mul = options['decoder_lr_mul']
        self.net = torch.nn.Sequential(
            FullyConnectedLayer(n_features, self.hidden_dim, lr_multiplier=options['decoder_lr_mul']),
            torch.nn.Softplus(),
            FullyConnectedLayer(self.hidden_dim, 1 + options['decoder_output_dim'], lr_multiplier=options['decoder_lr_mul']))
        self._packed, self._key = None, None

    def packed(self):
        ps = [self.net[0].weight, self.net[0].bias, self.net[2].weight, self.net[2].bias]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if key != self._key:
            self._packed = eg3d_ops.pack_decoder(*ps, lr_mul=self.lr_mul)
            self._key = key
        return self._packed

    def forward(self, sampled_features, ray_directions):
        raise NotImplementedError("the decoder runs fused behind the tri-plane gather (ImportanceRenderer.run_model); "
                                  "a stand-alone call on materialised features is not provided")
