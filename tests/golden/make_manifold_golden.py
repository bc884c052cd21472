#!/usr/bin/env python3
"""Generates tests/golden/manifolds.json — 100-digit golden vectors for ceres::Manifold::Plus / PlusJacobian / Minus / MinusJacobian of the
variable classes on the path: quaternion (left-multiplicative, full-angle exponential), R^n,
SphereManifold<3> (Householder construction), and the products used for control points, extrinsics and bias points.

Plus is evaluated with the mpmath retractions of make_golden.py (which share no code with the oracle or the kernels);
PlusJacobian = d Plus(x, delta) / d delta at delta = 0 by central differences with step 1e-20 at 100 digits.
Minus(y, x) with y = the double-rounded Plus(x, delta) is computed from the geometric definition — the tangent vector of the geodesic from
x to y in the coordinates of the tangent basis PlusJacobian(x) spans (quaternion: angle-axis of y x^-1; sphere: angle between x and y along the
normalised rejection of y from x) — and then CHECKED against the retraction: Plus(x, Minus(y, x)) must reproduce the direction of y to 1e-60 (on the sphere the geometric value is the first guess of a root search for exactly that property, see minus()).
MinusJacobian = d Minus(y, x) / dy at y = x by central differences in the ambient coordinates of y.
Inputs are rounded to doubles first, so the expected values belong to exactly representable inputs.
Run:  python tests/golden/make_manifold_golden.py   (seconds)
"""
import json
import os
import sys

import mpmath as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import H, SplitMix64, plus_quat, plus_sphere, qconj, qmul, tofloat  # noqa: E402  (sets mp.dps = 100)

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


def minus(kind, y, x):
    """Geometric definition of Manifold::Minus (not Ceres' formulas): see the module docstring."""
    def quat(yq, xq):
        n2 = sum(c * c for c in xq)
        r = qmul(yq, [c / n2 for c in qconj(xq)])  # y x^-1
        nv = mp.sqrt(r[0] ** 2 + r[1] ** 2 + r[2] ** 2)
        if nv == 0:
            return [mp.mpf(0)] * 3
        theta = mp.atan2(nv, r[3])  # rotation half-angle of the left factor: Plus uses [sin|d| d/|d| ; cos|d|] (x) x
        return [theta * c / nv for c in r[:3]]
    if kind == CONSTANT:
        return []
    if kind == EUCLIDEAN:
        return [a - b for a, b in zip(y, x)]
    if kind in (CONTROL_POINT, SE3):
        return quat(y[:4], x[:4]) + [y[4 + i] - x[4 + i] for i in range(3)]
    if kind == BIAS_POINT:
        return [y[i] - x[i] for i in range(3)]
    # sphere: geometric first guess (angle between x and y along the normalised rejection of y from x, in the coordinates of the
    # tangent basis PlusJacobian(x) / |x|), then refined so that Plus(x, delta) points exactly at y: Ceres' retraction is built on a
    # Householder reflection whose sigma <= eps branch (x within 1e-8 of the z axis) is not exactly orthogonal — there Plus(x, 0) is
    # 4e-9 away from x and "the inverse of Plus", which is what Minus is, differs from the geodesic definition by that much.
    nx = mp.sqrt(sum(c * c for c in x))
    J = plus_jacobian(SPHERE3, x, 2)
    basis = [[J[r][c] / nx for r in range(3)] for c in range(2)]
    xh = [c / nx for c in x]
    ny = mp.sqrt(sum(c * c for c in y))
    yh = [c / ny for c in y]
    along = sum(a * b for a, b in zip(yh, xh))
    rej = [a - along * b for a, b in zip(yh, xh)]
    nr = mp.sqrt(sum(c * c for c in rej))
    theta = mp.atan2(nr, along)
    guess = [theta * sum(rej[r] * basis[c][r] for r in range(3)) / nr if nr != 0 else mp.mpf(0) for c in range(2)]

    def miss(d0, d1):
        p = plus_sphere(x, [d0, d1])
        n = mp.sqrt(sum(c * c for c in p))
        return [sum((p[r] / n - yh[r]) * basis[c][r] for r in range(3)) for c in range(2)]
    if max(abs(m) for m in miss(*guess)) < mp.mpf(10) ** -80:
        return guess
    root = mp.findroot(miss, guess, tol=mp.mpf(10) ** -85, maxsteps=60)
    return [root[0], root[1]]


def minus_jacobian(kind, x, tangent):
    rows = [[mp.mpf(0)] * len(x) for _ in range(tangent)]
    for c in range(len(x)):
        yp, ym = list(x), list(x)
        yp[c] += H
        ym[c] -= H
        a, b = minus(kind, yp, x), minus(kind, ym, x)
        for r in range(tangent):
            rows[r][c] = (a[r] - b[r]) / (2 * H)
    return rows  # tangent x ambient


def direction(kind, v):
    """The part of an ambient point the manifold constrains up to scale, normalised (quaternion / sphere), the rest as is."""
    if kind in (CONTROL_POINT, SE3):
        n = mp.sqrt(sum(c * c for c in v[:4]))
        return [c / n for c in v[:4]] + list(v[4:])
    if kind == SPHERE3:
        n = mp.sqrt(sum(c * c for c in v))
        return [c / n for c in v]
    return list(v)


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
        y = tofloat(plus(kind, xm, dm))  # Minus is evaluated at the representable point y
        ym = [mp.mpf(v) for v in y]
        back = minus(kind, ym, xm)
        # definitional check: the retraction of x by Minus(y, x) points at y (norms of x and y differ by double rounding only)
        a, b = direction(kind, plus(kind, xm, back)), direction(kind, ym)
        assert max([abs(u - v) for u, v in zip(a, b)] + [mp.mpf(0)]) < mp.mpf(10) ** -60, (kind, x, d, max([abs(u - v) for u, v in zip(a, b)]))
        cases.append({"kind": kind, "ambient": len(x), "tangent": tangent, "x": x, "delta": d,
                      "plus": y, "jacobian": tofloat(plus_jacobian(kind, xm, tangent)),
                      "minus": tofloat(back), "minus_jacobian": tofloat(minus_jacobian(kind, xm, tangent))})

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
