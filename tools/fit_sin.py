#!/usr/bin/env python
"""Coefficients of csrc/mlp_core.h sin_pi(): sin(r) = r + r^3 P(r^2) on [-pi/2, pi/2], P of degree 3 (NT = 4 terms; NT = 5 is no
more accurate once evaluated in fp32), fitted by
iteratively re-weighted least squares (Lawson) on Chebyshev nodes; then the whole routine (magic-number rounding of
x/pi, two-term Cody-Waite reduction, Horner with fma) is emulated in fp32 and compared with fp64 sin."""
import numpy as np

r = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) * (np.pi / 2)
r = r[r > 0]
t = r * r
y = (np.sin(r) - r) / r ** 3
w = r ** 3
NT = 4
A = np.vander(t, NT, increasing=True)
for _ in range(60):
    c = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)[0]
    err = np.abs((A @ c - y) * r ** 3)
    w = w * (1 + 2 * err / err.max())
    w = w / w.max() * (r ** 3).max()
print("c3, c5, c7, ... =", [float(np.float32(v)) for v in c], " fp64 fit error", err.max())
c32 = [np.float32(v) for v in c]
F = np.float32
D = np.float64


def sin_pi32(x):
    x = F(x)
    magic = F(12582912.0)
    jm = F(D(x) * D(F(0.31830988618379067)) + D(magic))
    j = F(jm - magic)
    rr = F(D(x) - D(j) * D(F(3.14159274101257324)))
    rr = F(D(rr) - D(j) * D(F(-8.74227765734758577e-8)))
    r2 = F(rr * rr)
    p = c32[NT - 1]
    for k in range(NT - 2, -1, -1):
        p = F(D(p) * D(r2) + D(c32[k]))
    s = F(D(F(rr * r2)) * D(p) + D(rr))
    return np.where((jm.view(np.uint32) & 1) == 1, -s, s).astype(F)


for name, xs in (("N(0, 40^2)", (np.random.default_rng(0).standard_normal(2_000_000) * 40).astype(F)),
                 ("[-300, 300]", np.linspace(-300, 300, 3_000_001).astype(F))):
    e = np.abs(sin_pi32(xs) - np.sin(xs.astype(D)))
    print(f"{name}: max abs error {e.max():.3e} ({e.max() / 2 ** -24:.2f} ulp of 1), mean {e.mean():.3e}")
