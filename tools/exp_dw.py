#!/usr/bin/env python
"""Experiment harness: time nerf_backward_rays' kernels for builds of libnerfmi with -D flags.
usage (GPU box): python tools/exp_dw.py <lib.so> [...]"""
import ctypes as C
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

N_RAYS, P = 1024, 128


def run(libpath):
    lib = C.CDLL(libpath)
    lib.nerfmi_nerf_packed_floats.restype = C.c_size_t
    lib.nerfmi_nerf_saved_floats.restype = C.c_size_t
    lib.nerfmi_nerf_saved_floats.argtypes = [C.c_int64]
    lib.nerfmi_nerf_backward_workspace_floats.restype = C.c_size_t
    lib.nerfmi_nerf_backward_workspace_floats.argtypes = [C.c_int64]
    dev = torch.device("cuda:0")
    npts = N_RAYS * P
    packed = torch.randn(lib.nerfmi_nerf_packed_floats(), device=dev) * 0.05
    saved = torch.randn(lib.nerfmi_nerf_saved_floats(npts), device=dev)
    ws = torch.empty(lib.nerfmi_nerf_backward_workspace_floats(npts), device=dev)
    gout = torch.randn(npts, 4, device=dev)
    rays = torch.randn(N_RAYS, 8, device=dev)
    z = torch.rand(N_RAYS, P, device=dev)
    from nerf_siren_amd import ops
    grads = ops.flat_views(torch.empty(ops.PARAM_NUMEL, device=dev))
    arr = (C.c_void_p * 24)(*[g.data_ptr() for g in grads])
    vp = C.c_void_p
    fn = lib.nerfmi_nerf_backward_rays
    fn.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_void_p), vp, vp]
    fn.restype = C.c_int

    def call():
        rc = fn(packed.data_ptr(), rays.data_ptr(), z.data_ptr(), N_RAYS, P, saved.data_ptr(), gout.data_ptr(), arr,
                ws.data_ptr(), None)
        assert rc == 0

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(p, f"{run(p):.3f} ms (chain + dW + reduce, fine pass)")
