import sys, numpy as np, torch
sys.path.insert(0, '.')
from nerf_siren_amd import synth, SemanticNeRF, ops
dev = torch.device('cuda:0')
p = synth.siren_params(3)
m = SemanticNeRF(); m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}); m = m.to(dev)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
freq, phase = synth.hash_normal((1, 2304), 301), synth.hash_normal((1, 2304), 302)
for n in (37, 128, 1024):
    rays = synth.blender_rays(n, 33)
    z = torch.rand(n, 64, device=dev) * 4 + 2
    a = ops.siren_forward_rays(m.packed(), T(rays), z, T(freq), T(phase), n)
    b = ops.siren_forward_rays_fast(m.packed(), m.packed_fast(), T(rays), z, T(freq), T(phase), n)
    bs = ops.siren_forward_rays_fast(m.packed(), m.packed_fast(), T(rays), z, T(freq), T(phase), n, sigma_only=True)
    print(n, 'nan frac full', float(torch.isnan(b).float().mean()), 'sigma-only', float(torch.isnan(bs).float().mean()),
          'max diff where finite', float((a - b)[~torch.isnan(b)].abs().max()) if (~torch.isnan(b)).any() else None)
    bad = torch.isnan(b).any(1).nonzero().flatten()
    print('  first bad points', bad[:20].tolist())
