#!/usr/bin/env python
"""Experiment: per-workgroup shader-clock duration of siren_dw_kernel, grouped by task (needs a -DNERFMI_TIMING build).
usage (GPU box): NERFMI_LIB=<lib.so> python tools/exp_siren_dw_timing.py [c0,c1,c2]   (chunks per kind of the build)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nerf_siren_amd import _lib, ops, synth, SemanticNeRF

dev = torch.device("cuda:0")
N_RAYS, P = 1024, 128
m = SemanticNeRF()
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.siren_params(3).items()})
m = m.to(dev)
rays = torch.from_numpy(synth.blender_rays(N_RAYS, 1)).to(dev)
z = torch.rand(N_RAYS, P, device=dev) * 4 + 2
fr, ph = torch.randn(1, 2304, device=dev), torch.randn(1, 2304, device=dev)
out, saved = ops.siren_forward_rays_train(m.packed(), rays, z, fr, ph, N_RAYS)
g = torch.randn(N_RAYS * P, 4, device=dev)
for _ in range(5):
    ops.siren_backward(m.packed(), saved, g, fr, N_RAYS * P)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.siren_backward(m.packed(), saved, g, fr, N_RAYS * P)
e1.record()
torch.cuda.synchronize()
print("backward (chain + dW + reduce) ms:", e0.elapsed_time(e1) / 10)
buf = (C.c_ulonglong * 512)()
lib = C.CDLL(_lib.LIB_PATH)
assert lib.nerfmi_debug_timing_siren_dw(buf) == 0
t = np.array(buf, dtype=np.uint64).astype(np.int64)
kinds = [1] + [0] * 7 + [1, 0, 2, 2]            # siren_bwd.hip siren_plan task order
base = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [28, 10, 6]
off = 0
tot = {0: [], 1: [], 2: []}
for i, k in enumerate(kinds):
    n = base[k]
    seg = t[off:off + n]
    tot[k].append(float(np.median(seg)) * n)
    print(f"task {i:2d} kind {k} chunks {n:3d}: cycles median {int(np.median(seg)):9d} max {seg.max():9d}")
    off += n
print("workgroups", off, "kernel-limiting", t[:off].max())
print("total cycles per task by kind (median x chunks):", {k: int(np.mean(v)) for k, v in tot.items()})
