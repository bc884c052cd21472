#!/usr/bin/env python3
"""Generates tests/golden/manifolds.json — 100-digit golden vectors for ceres::Manifold::Plus / PlusJacobian of the
variable classes on the path: quaternion (left-multiplicative, full-angle exponential), R^n,
SphereManifold<3> (Householder construction), and the products used for control points, extrinsics and bias points.

Plus is evaluated with the mpmath retractions of make_golden.py (which share no code with the oracle or the kernels);
PlusJacobian = d Plus(x, delta) / d delta at delta = 0 by central differences with step 1e-20 at 100 digits.
Inputs are rounded to doubles first, so the expected values belong to exactly representable inputs.
Run:  python tests/golden/make_manifold_golden.py   (seconds)
"""
import json
import os
import sys

import mpmath as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import H, SplitMix64, plus_quat, plus_sphere, tofloat  # noqa: E402  (sets mp.dps = 100)

# kind ids of include/hyperslam_hip.h
CONSTANT, EUCLIDEAN, CONTROL_POINT, SE3, SPHERE3, BIAS_POINT = range(6)
TANGENT = {CONTROL_POINT: 6, SE3: 6, SPHERE3: 2, BIAS_POINT: 3}


def plus(kind, x, d):
    if kind == CONSTANT:
        return list(x)
    if kind == EUCLIDEAN:
        return [a + b for a, b in zip(x, d)]
    if kind in (CONTROL_POINT, SE3):
        out = plus_quat(x[:4], d[:3]) + [x[4 + i] + d[3 + i] for i in range(3)]
        return out + ([x[7]] if kind == CONTROL_POINT else [])
    if kind == SPHERE3:
        return plus_sphere(x, d)
    return [x[0] + d[0], x[1] + d[1], x[2] + d[2], x[3]]


def plus_jacobian(kind, x, tangent):
    cols = []
    for c in range(tangent):
        dp = [mp.mpf(0)] * tangent
        dm = list(dp)
        dp[c], dm[c] = H, -H
        a, b = plus(kind, x, dp), plus(kind, x, dm)
        cols.append([(u - v) / (2 * H) for u, v in zip(a, b)])
    return [[cols[c][r] for c in range(tangent)] for r in range(len(x))]  # ambient x tangent


def main():
    rng = SplitMix64(0x4D414E49)

    def u(lo, hi):
        return float(rng.uniform(lo=lo, hi=hi))

    def unit_quat():
        q = [u(-1, 1) for _ in range(4)]
        n = sum(c * c for c in q) ** 0.5
        return [c / n for c in q]  # unit up to double rounding, like a stored control point

    cases = []

    def add(kind, x, d):
        xm, dm = [mp.mpf(v) for v in x], [mp.mpf(v) for v in d]
        tangent = len(d)
        cases.append({"kind": kind, "ambient": len(x), "tangent": tangent, "x": x, "delta": d,
                      "plus": tofloat(plus(kind, xm, dm)), "jacobian": tofloat(plus_jacobian(kind, xm, tangent))})

    for scale in (0.0, 1e-9, 1e-3, 0.3, 2.5):  # |delta| from exactly zero to beyond pi/2
        for _ in range(2):
            add(CONTROL_POINT, unit_quat() + [u(-3, 3) for _ in range(3)] + [u(0, 10)], [scale * u(-1, 1) for _ in range(6)])
            add(SE3, unit_quat() + [u(-1, 1) for _ in range(3)], [scale * u(-1, 1) for _ in range(6)])
            g = [u(-1, 1), u(-1, 1), u(-1, 1)]
            n = sum(c * c for c in g) ** 0.5
            add(SPHERE3, [9.80665 * c / n for c in g], [scale * u(-1, 1) for _ in range(2)])
            add(BIAS_POINT, [u(-0.1, 0.1) for _ in range(3)] + [u(0, 10)], [scale * u(-1, 1) for _ in range(3)])
            add(EUCLIDEAN, [u(-5, 5) for _ in range(3)], [scale * u(-1, 1) for _ in range(3)])
    # SphereManifold branches: the default gravity (0, 0, -g) and +z take the sigma <= eps branch, pivot sign on either side of 0
    for x in ([0.0, 0.0, -9.80665], [0.0, 0.0, 9.80665], [1e-9, -2e-9, -9.80665], [0.3, -0.2, 1e-12], [0.6, 0.0, -0.8]):
        add(SPHERE3, x, [0.01, -0.02])
        add(SPHERE3, x, [1.2, 0.7])
    add(CONSTANT, [u(-1, 1) for _ in range(7)], [])
    with open(os.path.join(HERE, "manifolds.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_manifold_golden.py (mpmath, 100 digits)", "cases": cases}, f, separators=(",", ":"))
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
