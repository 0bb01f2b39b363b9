#!/usr/bin/env python3
"""Coefficients of g(z) = atan(sqrt z) / sqrt z on z in [0, 1] (degree 17, Chebyshev interpolation converted to the
monomial basis), used by cq_atan2 in csrc/cqpsk.hip: atan(t) = t g(t^2), |t| <= 1, max error 1.1e-15 in float64
(checked below against numpy on 2e6 points, Estrin evaluation as in the kernel)."""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P

deg = 17
k = np.arange(deg + 1)
x = np.cos(np.pi * (k + 0.5) / (deg + 1))
z = (x + 1) / 2
g = np.where(z > 0, np.arctan(np.sqrt(z)) / np.sqrt(np.where(z > 0, z, 1)), 1.0)
pu = C.cheb2poly(C.chebfit(x, g, deg))
lin, powp, c = np.array([-1.0, 2.0]), np.array([1.0]), np.zeros(deg + 1)
for ci in pu:
    c[:len(powp)] += ci * powp
    powp = P.polymul(powp, lin)


def estrin(c, z):
    lvl = [c[i] + c[i + 1] * z for i in range(0, len(c), 2)]
    p = z * z
    while len(lvl) > 1:
        lvl = [lvl[i] + lvl[i + 1] * p if i + 1 < len(lvl) else lvl[i] for i in range(0, len(lvl), 2)]
        p = p * p
    return lvl[0]


t = np.random.default_rng(1).random(2_000_000)
print("// max |t g(t^2) - atan t| on [0, 1]: %.2e" % np.max(np.abs(t * estrin(c, t * t) - np.arctan(t))))
print("static __device__ const double CQ_ATAN[18] = {")
for i in range(0, 18, 3):
    print("    " + ", ".join("%.17e" % v for v in c[i:i + 3]) + ",")
print("};")
