#!/usr/bin/env python3
"""Generates tests/golden/factors.json — independent 100-digit restatement of the four factors (golden vectors).

The reference holds no golden vectors for this path and cannot be built here, so these vectors pin the
oracle and the HIP path against an implementation that shares NO code and NO derivation with them:
  * B-spline basis by the Cox-de Boor recursion on uniform knots (not the closed-form blending matrix),
  * rotations through mpmath quaternion exp / log at 100 digits,
  * angular velocity / acceleration by numerical time differentiation of R(t) (not the recursive formulas),
  * every Jacobian by central differences through the Ceres retractions with step 1e-20 at 100 digits
    (truncation error ~1e-40) — no analytic Jacobian formula appears in this file.
Run:  python tests/golden/make_golden.py   (rewrites factors.json deterministically; ~12 minutes: 230 cases, every block's Jacobian)
"""
import json
import os
import sys

import mpmath as mp

mp.mp.dps = 100  # nested differences (alpha, then Jacobians with step 1e-20) cost ~45 digits of cancellation
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from hyperslam_amd.synthetic import SplitMix64  # only the RNG (deterministic inputs)  # noqa: E402

H = mp.mpf(10) ** -20   # parameter step
HT = mp.mpf(10) ** -15  # time step


# ---- quaternions (x, y, z, w), Hamilton ---------------------------------------------------------------------------------
def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz]


def qconj(q):
    return [-q[0], -q[1], -q[2], q[3]]


def qexp(phi):
    t = mp.sqrt(sum(x * x for x in phi))
    if t == 0:
        return [mp.mpf(0)] * 3 + [mp.mpf(1)]
    s = mp.sin(t / 2) / t
    return [s * phi[0], s * phi[1], s * phi[2], mp.cos(t / 2)]


def qlog(q):
    q = list(q)
    if q[3] < 0:
        q = [-x for x in q]
    n = mp.sqrt(q[0] ** 2 + q[1] ** 2 + q[2] ** 2)
    if n == 0:
        return [mp.mpf(0)] * 3
    s = 2 * mp.atan2(n, q[3]) / n
    return [s * q[0], s * q[1], s * q[2]]


def qrot(q, v):
    r = qmul(qmul(q, list(v) + [mp.mpf(0)]), qconj(q))
    return r[:3]


def qnorm(q):
    n = mp.sqrt(sum(x * x for x in q))
    return [x / n for x in q]


# ---- uniform B-spline basis by Cox-de Boor -----------------------------------------------------------------------------
def bspline_basis(k, u):
    """Values at normalised time u in [0,1) of the k basis functions of order k that are non-zero on the segment."""
    # knots ..., -2, -1, 0, 1, 2, ... ; segment [0, 1); basis N_{j,k} with support [j, j+k), j = -(k-1) .. 0
    def N(j, order, x):
        if order == 1:
            return mp.mpf(1) if j <= x < j + 1 else mp.mpf(0)
        return (x - j) / (order - 1) * N(j, order - 1, x) + (j + order - x) / (order - 1) * N(j + 1, order - 1, x)
    return [N(j, k, u) for j in range(-(k - 1), 1)]


def cumulative(k, u):
    b = bspline_basis(k, u)
    return [sum(b[j:]) for j in range(k)]


# ---- spline value --------------------------------------------------------------------------------------------------------
def spline_pose(cps, k, t):
    """cps: k control points [q(4) p(3) stamp]; returns (q, p). Uses the stamps of the control points (uniform)."""
    i = (k - 1) // 2
    dt = cps[i + 1][7] - cps[i][7]
    u = (t - cps[i][7]) / dt
    lam = cumulative(k, u)
    q = cps[0][:4]
    p = list(cps[0][4:7])
    for j in range(1, k):
        d = qlog(qmul(qconj(cps[j - 1][:4]), cps[j][:4]))
        q = qmul(q, qexp([lam[j] * x for x in d]))
        for c in range(3):
            p[c] += lam[j] * (cps[j][4 + c] - cps[j - 1][4 + c])
    return qnorm(q), p


def body_rates(cps, k, t):
    """(w_b, alpha_b, v_w, a_w) by numerical differentiation in time."""
    def w_at(tt):
        qm, _ = spline_pose(cps, k, tt - HT)
        qp, _ = spline_pose(cps, k, tt + HT)
        q0, _ = spline_pose(cps, k, tt)
        d = qlog(qmul(qconj(qm), qp))  # Log(R(t-h)^T R(t+h)) = 2h w_b + O(h^3)
        return [x / (2 * HT) for x in d], q0
    w, _ = w_at(t)
    hh = mp.mpf(10) ** -12
    wp, _ = w_at(t + hh)
    wm, _ = w_at(t - hh)
    al = [(a - b) / (2 * hh) for a, b in zip(wp, wm)]
    _, pm = spline_pose(cps, k, t - hh)
    _, p0 = spline_pose(cps, k, t)
    _, pp = spline_pose(cps, k, t + hh)
    v = [(a - b) / (2 * hh) for a, b in zip(pp, pm)]
    a = [(x - 2 * y + z) / (hh * hh) for x, y, z in zip(pp, p0, pm)]
    return w, al, v, a


def r3_spline(cps, k, t):
    i = (k - 1) // 2
    dt = cps[i + 1][3] - cps[i][3]
    u = (t - cps[i][3]) / dt
    b = bspline_basis(k, u)
    return [sum(b[j] * cps[j][c] for j in range(k)) for c in range(3)]


# ---- factors (residual functions only) ---------------------------------------------------------------------------------
def to_sensor(q_wb, p_wb, T_bs, p_w):
    v = [p_w[c] - p_wb[c] for c in range(3)]
    vb = qrot(qconj(q_wb), v)
    vb = [vb[c] - T_bs[4 + c] for c in range(3)]
    return qrot(qconj(T_bs[:4]), vb)


def res_pixel(P):
    q, p = spline_pose(P["cps"], P["k"], P["stamp"])
    ps = to_sensor(q, p, P["T_bs"], P["landmark"])
    x, y = ps[0] / ps[2], ps[1] / ps[2]
    k1, k2, p1, p2 = P["distortion"]
    cx, cy, fx, fy = P["intrinsics"]
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return [cx + fx * xd - P["meas"][0], cy + fy * yd - P["meas"][1]]


def res_bearing(P):
    q, p = spline_pose(P["cps"], P["k"], P["stamp"])
    ps = to_sensor(q, p, P["T_bs"], P["landmark"])
    b = P["meas"]
    c = [ps[1] * b[2] - ps[2] * b[1], ps[2] * b[0] - ps[0] * b[2], ps[0] * b[1] - ps[1] * b[0]]
    return [mp.atan2(mp.sqrt(sum(x * x for x in c)), sum(ps[i] * b[i] for i in range(3)))]


def res_prior(P):
    q, p = spline_pose(P["cps"], P["k"], P["stamp"])
    T = P["T_bs"]
    q_ws = qmul(q, T[:4])
    Rt = qrot(q, T[4:7])
    rot = qlog(qmul(qconj(P["meas"][:4]), q_ws))
    return rot + [Rt[c] + p[c] - P["meas"][4 + c] for c in range(3)]


def align_matrix(c):
    return [[c[0], 0, 0], [c[3], c[1], 0], [c[4], c[5], c[2]]]


def matvec(M, v):
    return [sum(M[i][j] * v[j] for j in range(3)) for i in range(3)]


def cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def res_inertial(P):
    k = P["k"]
    q, _ = spline_pose(P["cps"], k, P["stamp"])
    w, al, _, a_w = body_rates(P["cps"], k, P["stamp"])
    T = P["T_bs"]
    g = P["gravity"]
    a_i = qrot(qconj(q), [a_w[c] - g[c] for c in range(3)])  # R_bw (a_w - g)
    Sg = [[P["S_g"][3 * c + r] for c in range(3)] for r in range(3)]  # column-major maps
    Xa = [[P["X_a"][3 * c + r] for c in range(3)] for r in range(3)]
    a_m = []
    for i in range(3):
        lever = [Xa[r][i] + T[4 + r] for r in range(3)]
        f = cross(w, cross(w, lever))
        f2 = cross(al, lever)
        a_m.append(a_i[i] + f[i] + f2[i])
    w_s = qrot(qconj(T[:4]), w)
    a_s = qrot(qconj(T[:4]), a_m)
    b_g = r3_spline(P["bias_g"], P["kb"], P["stamp"])
    b_a = r3_spline(P["bias_a"], P["kb"], P["stamp"])
    ang = matvec(align_matrix(P["i_g"]), w_s)
    sg = matvec(Sg, a_m)
    lin = matvec(align_matrix(P["i_a"]), a_s)
    pred = [ang[c] + sg[c] + b_g[c] for c in range(3)] + [lin[c] + b_a[c] for c in range(3)]
    return [pred[c] - P["meas"][c] for c in range(6)]


RES = {"pixel": res_pixel, "bearing": res_bearing, "prior": res_prior, "inertial": res_inertial}


# ---- Ceres retractions ------------------------------------------------------------------------------------
def plus_quat(q, d):
    n = mp.sqrt(sum(x * x for x in d))
    if n == 0:
        return list(q)
    s = mp.sin(n) / n
    return qmul([s * d[0], s * d[1], s * d[2], mp.cos(n)], q)


def plus_sphere(x, d):
    nd = mp.sqrt(d[0] ** 2 + d[1] ** 2)
    if nd == 0:
        return list(x)
    sigma = x[0] ** 2 + x[1] ** 2
    v = [x[0], x[1], mp.mpf(1)]
    beta = mp.mpf(0)
    if sigma <= mp.mpf(2) ** -52:
        if x[2] < 0:
            beta = mp.mpf(2)
    else:
        mu = mp.sqrt(x[2] ** 2 + sigma)
        vp = x[2] - mu if x[2] <= 0 else -sigma / (x[2] + mu)
        beta = 2 * vp * vp / (sigma + vp * vp)
        v[0] /= vp
        v[1] /= vp
    s = mp.sin(nd) / nd
    y = [s * d[0], s * d[1], mp.cos(nd)]
    nx = mp.sqrt(sum(c * c for c in x))
    vy = beta * sum(v[i] * y[i] for i in range(3))
    return [nx * (y[i] - v[i] * vy) for i in range(3)]


def perturbed(P, block, col, h):
    """Copy of the problem with local coordinate `col` of `block` moved by h through the block's Ceres manifold."""
    Q = {k: ([list(r) for r in v] if isinstance(v, list) and v and isinstance(v[0], list) else (list(v) if isinstance(v, list) else v))
         for k, v in P.items()}
    kind, idx = block
    d = [mp.mpf(0)] * 3
    if kind == "cp_rot":
        d[col] = h
        Q["cps"][idx][:4] = plus_quat(P["cps"][idx][:4], d)
    elif kind == "cp_trans":
        Q["cps"][idx][4 + col] += h
    elif kind == "landmark":
        Q["landmark"][col] += h
    elif kind == "bias_g":
        Q["bias_g"][idx][col] += h
    elif kind == "bias_a":
        Q["bias_a"][idx][col] += h
    elif kind == "gravity":
        dd = [mp.mpf(0)] * 2
        dd[col] = h
        Q["gravity"] = plus_sphere(P["gravity"], dd)
    elif kind == "T_bs_rot":  # sensor extrinsics: Product(EigenQuaternion, R3) manifold (sensors/sensor.cpp:26-29)
        d[col] = h
        Q["T_bs"][:4] = plus_quat(P["T_bs"][:4], d)
    elif kind == "T_bs_trans":
        Q["T_bs"][4 + col] += h
    elif kind in ("intrinsics", "distortion", "i_g", "i_a", "S_g", "X_a"):  # Euclidean sensor blocks
        Q[kind][col] += h
    return Q


def jacobian(ftype, P, block, ncols):
    cols = []
    for c in range(ncols):
        rp = RES[ftype](perturbed(P, block, c, H))
        rm = RES[ftype](perturbed(P, block, c, -H))
        cols.append([(a - b) / (2 * H) for a, b in zip(rp, rm)])
    return [[cols[c][r] for c in range(ncols)] for r in range(len(cols[0]))]  # n_res x ncols


# ---- case generation -----------------------------------------------------------------------------------------------------
def rand_quat(rng):
    return qnorm([mp.mpf(rng.uniform(lo=-1.0, hi=1.0)) for _ in range(4)])


def perp_unit(v, rng):
    """A unit vector orthogonal to v."""
    a = [mp.mpf(rng.uniform(lo=-1.0, hi=1.0)) for _ in range(3)]
    c = cross(v, a)
    n = mp.sqrt(sum(x * x for x in c))
    return [x / n for x in c]


def make_case(ftype, k, rng, variant=None):
    """variant: None | "u0" / "u1" (stamp at the very start / end of its segment) | "near_pi" (prior: relative rotation pi - 1e-3)
    | "small_angle" (bearing: 1e-5 rad off the measurement) | "axis" / "wide" (pixel: on the optical axis / strong distortion)
    | "identity" (inertial: I_g = I_a = I, S_g = X_a = 0, where the in-tree Jacobian is exact) | "cp_near_pi" (two consecutive control
    points pi - 1e-2 apart: Log branch of the cumulative spline) | "gravity_pivot[_neg]" (gravity on the SphereManifold pivot axis)."""
    dt = mp.mpf("0.1")
    P = {"k": k}
    q = rand_quat(rng)
    cps = []
    t_first = mp.mpf(rng.uniform(lo=0.0, hi=5.0))
    for j in range(k):
        step = [mp.mpf(rng.uniform(lo=-0.2, hi=0.2)) for _ in range(3)]
        if variant == "cp_near_pi" and j == (k - 1) // 2 + 1:  # Log branch: consecutive control points pi - 1e-2 apart
            axis = perp_unit([mp.mpf(3), mp.mpf(-1), mp.mpf(2)], rng)
            step = [(mp.pi - mp.mpf("1e-2")) * x for x in axis]
        q = qmul(q, qexp(step))
        cps.append(q + [mp.mpf(rng.uniform(lo=-1.0, hi=1.0)) for _ in range(3)] + [t_first + dt * j])
    P["cps"] = cps
    P["stamp"] = cps[(k - 1) // 2][7] + dt * mp.mpf(rng.uniform(lo=0.05, hi=0.95))
    if variant == "u0":
        P["stamp"] = cps[(k - 1) // 2][7] + dt * mp.mpf("1e-9")
    if variant == "u1":
        P["stamp"] = cps[(k - 1) // 2][7] + dt * (1 - mp.mpf("1e-9"))
    P["T_bs"] = rand_quat(rng) + [mp.mpf(rng.uniform(lo=-0.3, hi=0.3)) for _ in range(3)]
    if ftype in ("pixel", "bearing"):
        P["intrinsics"] = [mp.mpf("367.215"), mp.mpf("248.375"), mp.mpf("458.654"), mp.mpf("457.296")]
        P["distortion"] = [mp.mpf("-0.28340811"), mp.mpf("0.07395907"), mp.mpf("1.76187114e-05"), mp.mpf("0.00019359")]
        qw, pw = spline_pose(cps, k, P["stamp"])
        q_ws = qmul(qw, P["T_bs"][:4])
        p_ws = [a + b for a, b in zip(qrot(qw, P["T_bs"][4:7]), pw)]
        ps = [mp.mpf(rng.uniform(lo=-1.5, hi=1.5)), mp.mpf(rng.uniform(lo=-1.0, hi=1.0)), mp.mpf(rng.uniform(lo=2.0, hi=8.0))]
        if variant == "axis":
            ps = [mp.mpf("1e-9"), mp.mpf("-2e-9"), ps[2]]
        if variant == "wide":
            ps = [mp.mpf("0.9") * ps[2], mp.mpf("-0.55") * ps[2], ps[2]]
        P["landmark"] = [a + b for a, b in zip(qrot(q_ws, ps), p_ws)]
        if ftype == "pixel":
            P["meas"] = [mp.mpf(rng.uniform(lo=0.0, hi=752.0)), mp.mpf(rng.uniform(lo=0.0, hi=480.0))]
        else:
            b = [ps[i] + mp.mpf(rng.uniform(lo=-0.2, hi=0.2)) for i in range(3)]
            if variant == "small_angle":
                e = perp_unit(ps, rng)
                npz = mp.sqrt(sum(x * x for x in ps))
                b = [ps[i] + mp.mpf("1e-5") * npz * e[i] for i in range(3)]
            n = mp.sqrt(sum(x * x for x in b))
            P["meas"] = [x / n for x in b]
    elif ftype == "prior":
        P["meas"] = rand_quat(rng) + [mp.mpf(rng.uniform(lo=-1.0, hi=1.0)) for _ in range(3)]
        if variant == "near_pi":  # R_m = R_ws Exp(-(pi - 1e-3) axis): Log(R_m^T R_ws) has angle pi - 1e-3
            qw, _ = spline_pose(cps, k, P["stamp"])
            axis = perp_unit([mp.mpf(1), mp.mpf(2), mp.mpf(3)], rng)
            ang = mp.pi - mp.mpf("1e-3")
            P["meas"][:4] = qmul(qmul(qw, P["T_bs"][:4]), qexp([-ang * x for x in axis]))
    else:
        kb = 4
        P["kb"] = kb
        P["i_g"] = [mp.mpf(1) + mp.mpf(rng.uniform(lo=-0.1, hi=0.1)) for _ in range(3)] + [mp.mpf(rng.uniform(lo=-0.1, hi=0.1)) for _ in range(3)]
        P["i_a"] = [mp.mpf(1) + mp.mpf(rng.uniform(lo=-0.1, hi=0.1)) for _ in range(3)] + [mp.mpf(rng.uniform(lo=-0.1, hi=0.1)) for _ in range(3)]
        P["S_g"] = [mp.mpf(rng.uniform(lo=-0.01, hi=0.01)) for _ in range(9)]
        P["X_a"] = [mp.mpf(rng.uniform(lo=-0.05, hi=0.05)) for _ in range(9)]
        if variant == "identity":
            P["i_g"] = [mp.mpf(1)] * 3 + [mp.mpf(0)] * 3
            P["i_a"] = [mp.mpf(1)] * 3 + [mp.mpf(0)] * 3
            P["S_g"], P["X_a"] = [mp.mpf(0)] * 9, [mp.mpf(0)] * 9
        bdt = mp.mpf(1)
        bt0 = P["stamp"] - bdt * ((kb - 1) // 2) - mp.mpf(rng.uniform(lo=0.05, hi=0.95))
        for name in ("bias_g", "bias_a"):
            P[name] = [[mp.mpf(rng.uniform(lo=-0.5, hi=0.5)) for _ in range(3)] + [bt0 + bdt * j] for j in range(kb)]
        gq = rand_quat(rng)
        P["gravity"] = [mp.mpf("9.80665") * x for x in qrot(gq, [mp.mpf(1), mp.mpf(0), mp.mpf(0)])]
        if variant in ("gravity_pivot", "gravity_pivot_neg"):  # SphereManifold<3> Householder pivot: x = (0, 0, +-|g|)
            P["gravity"] = [mp.mpf(0), mp.mpf(0), mp.mpf("9.80665") * (1 if variant == "gravity_pivot" else -1)]
        P["meas"] = [mp.mpf(rng.uniform(lo=-1.0, hi=1.0)) for _ in range(6)]
    return P


def tofloat(x):
    if isinstance(x, list):
        return [tofloat(v) for v in x]
    if isinstance(x, mp.mpf):
        return float(x)
    return x


def main():
    rng = SplitMix64(0x48595045 ^ 0x601DE)
    cases = []
    # random cases + targeted edge cases per factor (SURVEY.md §8c asks for O(256) blocks)
    plan = []
    for ftype in ("pixel", "bearing", "prior", "inertial"):
        for k in (4, 6):
            plan += [(ftype, k, None)] * (27 if ftype != "inertial" else 12)
            edge = {"pixel": ["u0", "u1", "axis", "wide", "cp_near_pi"], "bearing": ["u0", "u1", "small_angle", "small_angle", "cp_near_pi"],
                    "prior": ["u0", "u1", "near_pi", "near_pi", "cp_near_pi"],
                    "inertial": ["u0", "u1", "identity", "identity", "cp_near_pi", "gravity_pivot", "gravity_pivot_neg"]}[ftype]
            plan += [(ftype, k, v) for v in edge]
    # `--order5` (round 4): a file of its own, factors_k5.json — order 5 is instantiated on the device since then; 9 random + 3 targeted cases
    # per factor (the segment of a stamp starts (k - 1) // 2 = 2 control points before it), own random stream
    order5 = "--order5" in sys.argv
    if order5:
        plan = []
        for ftype in ("pixel", "bearing", "prior", "inertial"):
            plan += [(ftype, 5, None)] * 9 + [(ftype, 5, v) for v in ("u0", "u1", "cp_near_pi")]
        rng = SplitMix64(0x48595045 ^ 0x0B5E5)
    # second block (own random stream, appended behind the first so that the cases above keep their values): the inertial factor has the
    # most parameter blocks and gets as many random cases per order as the other factors, 32 with its edge cases
    n_first = len(plan)
    if not order5:
        plan += [("inertial", k, None) for k in (4, 6) for _ in range(13)]
    rng2 = SplitMix64(0x48595045 ^ 0x1E127)
    only_second = "--second-block-only" in sys.argv  # (re-uses the first block of an existing factors.json: minutes instead of an hour)
    if only_second:
        with open(os.path.join(HERE, "factors.json")) as f:
            cases = json.load(f)["cases"][:n_first]
        assert len(cases) == n_first
    for rep, (ftype, k, variant) in enumerate(plan):
        if only_second and rep < n_first:
            continue
        if True:
            if True:
                P = make_case(ftype, k, rng if rep < n_first else rng2, variant)
                # inputs are rounded to doubles FIRST, so that the golden outputs belong to exactly representable inputs
                P = {key: (tofloat(v) if not isinstance(v, int) else v) for key, v in P.items()}
                Pm = {key: ([[mp.mpf(x) for x in r] for r in v] if isinstance(v, list) and isinstance(v[0], list)
                            else ([mp.mpf(x) for x in v] if isinstance(v, list) else (mp.mpf(v) if isinstance(v, float) else v)))
                      for key, v in P.items()}
                out = {"r": tofloat(RES[ftype](Pm)), "J_state": None}
                Js = [[] for _ in out["r"]]
                for j in range(k):
                    Jr = jacobian(ftype, Pm, ("cp_rot", j), 3)
                    Jt = jacobian(ftype, Pm, ("cp_trans", j), 3)
                    for r in range(len(out["r"])):
                        Js[r] += tofloat(Jr[r]) + tofloat(Jt[r])
                out["J_state"] = Js
                # sensor parameter blocks (constant in the solver; probed by the reference's tests, evaluator.hpp:38-65)
                Jr = jacobian(ftype, Pm, ("T_bs_rot", 0), 3)
                Jt = jacobian(ftype, Pm, ("T_bs_trans", 0), 3)
                out["J_extrinsics"] = [tofloat(Jr[r]) + tofloat(Jt[r]) for r in range(len(out["r"]))]
                if ftype == "pixel":
                    out["J_intrinsics"] = tofloat(jacobian(ftype, Pm, ("intrinsics", 0), 4))
                    out["J_distortion"] = tofloat(jacobian(ftype, Pm, ("distortion", 0), 4))
                if ftype in ("pixel", "bearing"):
                    out["J_landmark"] = tofloat(jacobian(ftype, Pm, ("landmark", 0), 3))
                if ftype == "inertial":
                    out["J_gyro_intrinsics"] = tofloat(jacobian(ftype, Pm, ("i_g", 0), 6))
                    out["J_acc_intrinsics"] = tofloat(jacobian(ftype, Pm, ("i_a", 0), 6))
                    out["J_gyro_sensitivity"] = tofloat(jacobian(ftype, Pm, ("S_g", 0), 9))
                    out["J_acc_offsets"] = tofloat(jacobian(ftype, Pm, ("X_a", 0), 9))
                    kb = P["kb"]
                    Jg, Ja = [[] for _ in range(6)], [[] for _ in range(6)]
                    for j in range(kb):
                        a = jacobian(ftype, Pm, ("bias_g", j), 3)
                        b = jacobian(ftype, Pm, ("bias_a", j), 3)
                        for r in range(6):
                            Jg[r] += tofloat(a[r])
                            Ja[r] += tofloat(b[r])
                    out["J_bias_g"], out["J_bias_a"] = Jg, Ja
                    out["J_gravity"] = tofloat(jacobian(ftype, Pm, ("gravity", 0), 2))
                    w, al, v, a = body_rates(Pm["cps"], k, Pm["stamp"])
                    out["w_b"], out["alpha_b"], out["v_w"], out["a_w"] = tofloat(w), tofloat(al), tofloat(v), tofloat(a)
                q, p = spline_pose(Pm["cps"], k, Pm["stamp"])
                out["pose"] = tofloat(q + p)
                cases.append({"type": ftype, "variant": variant, "inputs": P, "outputs": out})
                print(ftype, k, rep, variant, "ok", flush=True)
    with open(os.path.join(HERE, "factors_k5.json" if order5 else "factors.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (mpmath, 100 digits)", "cases": cases}, f, indent=None, separators=(",", ":"))
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
