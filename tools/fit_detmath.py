#!/usr/bin/env python3
"""Fit the polynomial coefficients used by vulkansift_amd/csrc/detmath.h.

detmath.h implements exp / exp2 / atan2 / sin / cos with +,-,*,fma and bit operations only, so
that the HIP kernels and the CPU oracle (in its "det" math mode) produce bit-identical results.
The reference leaves these functions to the GLSL implementation (precision is implementation-
defined, Vulkan spec "Precision and Operation of SPIR-V Instructions"), so any <= few-ulp
implementation is within its contract.

This script does a least-squares fit on Chebyshev nodes in float64, rounds the coefficients to
float32, then measures the float32 (fma-emulated) error against float64 libm. It prints C
initialisers; the values printed were pasted into detmath.h.
"""
import numpy as np

f32 = np.float32


def cheb_nodes(a, b, n):
    k = np.arange(n)
    x = np.cos(np.pi * (2 * k + 1) / (2 * n))
    return 0.5 * (a + b) + 0.5 * (b - a) * x


def fit(fn, a, b, deg, n=4000, w=None):
    x = cheb_nodes(a, b, n)
    y = fn(x)
    V = np.vander(x, deg + 1, increasing=True)
    if w is not None:
        ww = w(x)
        c, *_ = np.linalg.lstsq(V * ww[:, None], y * ww, rcond=None)
    else:
        c, *_ = np.linalg.lstsq(V, y, rcond=None)
    return c.astype(f32)


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def horner32(c, x):
    acc = np.full_like(x, c[-1], dtype=f32)
    for k in range(len(c) - 2, -1, -1):
        acc = fma32(acc, x, np.full_like(x, c[k], dtype=f32))
    return acc


def ulp_err(got, ref):
    ref32 = ref.astype(f32)
    u = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.max(np.abs(got.astype(np.float64) - ref) / u)


def show(name, c):
    print(f"// {name}")
    print("{ " + ", ".join(f"{float(v).hex()}f /*{float(v):.9e}*/" for v in c) + " }")


rng = np.random.default_rng(1)

# ---- exp(r) on |r| <= ln2/2 :  e^r = 1 + r + r^2 * P(r)
ln2h = 0.5 * np.log(2.0)
cE = fit(lambda r: np.where(np.abs(r) < 1e-8, 0.5, (np.exp(r) - 1 - r) / np.where(r == 0, 1, r * r)), -ln2h * 1.01, ln2h * 1.01, 4)
r = rng.uniform(-ln2h, ln2h, 2_000_000).astype(f32)
p = horner32(cE, r)
e = fma32(fma32(p, r, np.ones_like(r)) * 0 + p * 0 + p, r * r, r + f32(1))  # (1+r) + r^2*p  (approx check)
show("exp: e^r = 1 + r + r*r*P(r), P coefficients c0..c4", cE)
print("   exp core max ulp ~", ulp_err(e, np.exp(r.astype(np.float64))))

# ---- exp2(r) on |r| <= 0.5 :  2^r = 1 + r*Q(r)
cX = fit(lambda r: np.where(np.abs(r) < 1e-9, np.log(2.0), (np.exp2(r) - 1) / np.where(r == 0, 1, r)), -0.505, 0.505, 6)
r = rng.uniform(-0.5, 0.5, 2_000_000).astype(f32)
q = horner32(cX, r)
e2 = fma32(q, r, np.ones_like(r))
show("exp2: 2^r = 1 + r*Q(r), Q coefficients c0..c6", cX)
print("   exp2 core max ulp ~", ulp_err(e2, np.exp2(r.astype(np.float64))))

# ---- atan(a) on [0,1] : atan(a) = a + a^3 * P(a^2)
def atan_p(z):
    a = np.sqrt(z)
    return np.where(a < 1e-6, -1.0 / 3.0, (np.arctan(a) - a) / np.where(a == 0, 1, a ** 3))

for deg in (8, 9, 10):
    cA = fit(atan_p, 0.0, 1.0, deg)
    a = rng.uniform(0, 1, 2_000_000).astype(f32)
    z = (a * a).astype(f32)
    pa = horner32(cA, z)
    at = fma32((pa * z).astype(f32), a, a)
    print(f"   atan deg {deg}: max ulp ~", ulp_err(at, np.arctan(a.astype(np.float64))),
          " max abs", np.max(np.abs(at.astype(np.float64) - np.arctan(a.astype(np.float64)))))
    if deg == 10:
        show("atan: atan(a) = a + a*z*P(z), z=a*a, P coefficients c0..c10", cA)

# ---- sin / cos on |r| <= pi/4
cS = fit(lambda z: np.where(z < 1e-12, -1.0 / 6.0, (np.sin(np.sqrt(z)) - np.sqrt(z)) / np.where(z == 0, 1, np.sqrt(z) ** 3)), 0.0, (np.pi / 4 * 1.01) ** 2, 3)
cC = fit(lambda z: np.where(z < 1e-12, -0.5, (np.cos(np.sqrt(z)) - 1) / np.where(z == 0, 1, z)), 0.0, (np.pi / 4 * 1.01) ** 2, 4)
r = rng.uniform(-np.pi / 4, np.pi / 4, 2_000_000).astype(f32)
z = (r * r).astype(f32)
s = fma32((horner32(cS, z) * z).astype(f32), r, r)
c = fma32(horner32(cC, z), z, np.ones_like(z))
show("sin: sin(r) = r + r*z*S(z), z=r*r, S coefficients c0..c3", cS)
print("   sin core max ulp ~", ulp_err(s, np.sin(r.astype(np.float64))))
show("cos: cos(r) = 1 + z*C(z), C coefficients c0..c4", cC)
print("   cos core max ulp ~", ulp_err(c, np.cos(r.astype(np.float64))))
