#!/usr/bin/env python
"""Experiment: per-layer shader-clock stamps of the fp32 forward kernel (needs a -DNERFMI_TIMING build).
usage (GPU box): python tools/exp_timing.py <lib.so>"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

N_RAYS, P = 1024, 128
lib = C.CDLL(sys.argv[1])
FAST = len(sys.argv) > 2 and sys.argv[2] == "fast"
SAVE = len(sys.argv) > 2 and sys.argv[2] == "save"
lib.nerfmi_nerf_packed_floats.restype = C.c_size_t
dev = torch.device("cuda:0")
packed = torch.randn(lib.nerfmi_nerf_packed_floats(), device=dev) * 0.05
rays = torch.randn(N_RAYS, 8, device=dev)
z = torch.rand(N_RAYS, P, device=dev)
out = torch.empty(N_RAYS * P, 4, device=dev)
vp = C.c_void_p
fn = lib.nerfmi_nerf_forward_rays
fn.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
fn.restype = C.c_int
if FAST:
    lib.nerfmi_nerf_fast_bytes.restype = C.c_size_t
    fast = torch.empty(lib.nerfmi_nerf_fast_bytes(), dtype=torch.uint8, device=dev)
    lib.nerfmi_nerf_pack_fast.argtypes = [vp, vp, vp]
    assert lib.nerfmi_nerf_pack_fast(packed.data_ptr(), fast.data_ptr(), None) == 0
    ff = lib.nerfmi_nerf_forward_rays_fast
    ff.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    ff.restype = C.c_int
    call = lambda: ff(packed.data_ptr(), fast.data_ptr(), rays.data_ptr(), z.data_ptr(), N_RAYS, P, 0, out.data_ptr(), None, None)
elif SAVE:
    lib.nerfmi_nerf_saved_floats.restype = C.c_size_t
    lib.nerfmi_nerf_saved_floats.argtypes = [C.c_int64]
    saved = torch.empty(lib.nerfmi_nerf_saved_floats(N_RAYS * P), device=dev)
    call = lambda: fn(packed.data_ptr(), rays.data_ptr(), z.data_ptr(), N_RAYS, P, 0, out.data_ptr(), saved.data_ptr(), None)
else:
    call = lambda: fn(packed.data_ptr(), rays.data_ptr(), z.data_ptr(), N_RAYS, P, 0, out.data_ptr(), None, None)
for _ in range(30):
    assert call() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    call()
e1.record()
torch.cuda.synchronize()
print("kernel ms:", e0.elapsed_time(e1) / 20)
buf = (C.c_ulonglong * (64 * 16))()
assert (lib.nerfmi_debug_timing_fast if FAST else lib.nerfmi_debug_timing)(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)[:32]
names = ["start", "embed", "L1", "L2", "L3", "L4", "L5", "L6", "L7", "L8", "final", "dir", "heads"]
d = np.diff(t[:, :13], axis=1)
print(f"{'phase':8s} {'median':>9s} {'min':>9s} {'max':>9s}   ideal(MFMA cycles)")
ideal = {"L1": 64, "L2": 256, "L3": 256, "L4": 256, "L5": 320, "L6": 256, "L7": 256, "L8": 256, "final": 256, "dir": 144}
for i, n in enumerate(names[1:]):
    idl = ideal.get(n, 0) * (6 * 32 // 2 if FAST else 4 * 64)
    print(f"{n:8s} {int(np.median(d[:, i])):9d} {d[:, i].min():9d} {d[:, i].max():9d}   {idl}")
print("total median", int(np.median(t[:, 12] - t[:, 0])), "ideal", sum(ideal.values()) * (96 if FAST else 256))
print("start stamps (relative):", (t[:, 0] - t[:, 0].min())[:32])
