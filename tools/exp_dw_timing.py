#!/usr/bin/env python
"""Experiment: per-workgroup shader-clock duration of the split-bf16 dW kernel, grouped by task kind
(needs a -DNERFMI_TIMING build).  usage (GPU box): python tools/exp_dw_timing.py <lib.so>"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

N_RAYS, P = 1024, 128
lib = C.CDLL(sys.argv[1])
for f in ("nerfmi_nerf_packed_floats", "nerfmi_nerf_fast_bytes"):
    getattr(lib, f).restype = C.c_size_t
lib.nerfmi_nerf_saved_floats.restype = C.c_size_t
lib.nerfmi_nerf_saved_floats.argtypes = [C.c_int64]
lib.nerfmi_nerf_backward_workspace_floats.restype = C.c_size_t
lib.nerfmi_nerf_backward_workspace_floats.argtypes = [C.c_int64]
dev = torch.device("cuda:0")
npts = N_RAYS * P
packed = torch.randn(lib.nerfmi_nerf_packed_floats(), device=dev) * 0.05
fast = torch.empty(lib.nerfmi_nerf_fast_bytes(), dtype=torch.uint8, device=dev)
vp = C.c_void_p
lib.nerfmi_nerf_pack_fast.argtypes = [vp, vp, vp]
assert lib.nerfmi_nerf_pack_fast(packed.data_ptr(), fast.data_ptr(), None) == 0
saved = torch.rand(lib.nerfmi_nerf_saved_floats(npts), device=dev)
ws = torch.empty(lib.nerfmi_nerf_backward_workspace_floats(npts), device=dev)
gout = torch.randn(npts, 4, device=dev)
from nerf_siren_amd import ops
grads = ops.flat_views(torch.empty(ops.PARAM_NUMEL, device=dev))
arr = (C.c_void_p * 24)(*[g.data_ptr() for g in grads])
FP32 = len(sys.argv) > 3 and sys.argv[3] == "fp32"
if FP32:
    fn = lib.nerfmi_nerf_backward_rays
    fn.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_void_p), vp, vp]
    rays = torch.randn(N_RAYS, 8, device=dev)
    z = torch.rand(N_RAYS, P, device=dev)
    call = lambda: fn(packed.data_ptr(), rays.data_ptr(), z.data_ptr(), N_RAYS, P, saved.data_ptr(), gout.data_ptr(), arr, ws.data_ptr(), None)
else:
    fn = lib.nerfmi_nerf_backward_rays_fast
    fn.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_void_p), vp, vp]
    call = lambda: fn(packed.data_ptr(), fast.data_ptr(), N_RAYS, P, saved.data_ptr(), gout.data_ptr(), arr, ws.data_ptr(), None)
fn.restype = C.c_int
for _ in range(5):
    assert call() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    call()
e1.record()
torch.cuda.synchronize()
print("backward (chain + dW + reduce) ms:", e0.elapsed_time(e1) / 10)
buf = (C.c_ulonglong * 512)()
assert lib.nerfmi_debug_timing_dw(buf) == 0
t = np.array(buf, dtype=np.uint64).astype(np.int64)
# plan order (mlp_bwd.hip make_plan): kinds of the 14 tasks and the split-bf16 chunk table
kinds = [1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 2, 3, 4, 5]
base = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [25, 8, 14, 8, 8, 10]
off = 0
for i, k in enumerate(kinds):
    n = base[k]
    seg = t[off:off + n]
    print(f"task {i:2d} kind {k} chunks {n:3d}: cycles median {int(np.median(seg)):9d} max {seg.max():9d}")
    off += n
print("workgroups", off, "kernel-limiting", t[:off].max())
