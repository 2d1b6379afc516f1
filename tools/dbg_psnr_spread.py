#!/usr/bin/env python
"""How sensitive is the validation-PSNR trajectory of the long teacher-scene protocols to rounding alone?  Runs the same
protocol (same data, batches, draws, initial weights, schedule) with the three arithmetic variants of the HIP path
(torch Adam + elementwise loss, fused loss/Adam, fused + split-bf16 math) and prints the trajectories beside the
reference's.  usage (GPU box): python tools/dbg_psnr_spread.py g19s_psnr_spheres"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_gpu_parity as T

name = sys.argv[1] if len(sys.argv) > 1 else "g19s_psnr_spheres"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
dev = torch.device("cuda:0")
print("reference     ", np.round(g["psnr"], 3))
for impl in ("torch", "fused", "fused+bf16x3"):
    p = np.array(T._psnr_protocol(g, dev, impl))
    print(f"{impl:14s}", np.round(p, 3), "diff vs reference", np.round(p - g["psnr"], 3))
