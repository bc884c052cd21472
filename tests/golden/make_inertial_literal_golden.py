#!/usr/bin/env python3
"""Generates tests/golden/inertial_literal.json — 100-digit vectors of the inertial Jacobian AS WRITTEN upstream
(/root/reference/internal/hyper/optimizers/evaluators/inertial.cpp:131-198), at IMU parameters where it is NOT the derivative of the
prediction (I_g != I_a, S_g != 0, X_a != 0): the default mode of both libraries (HS_INERTIAL_AS_REFERENCE), which factors.json pins only
at the identity point.

What is transcribed and what is measured. The in-tree text composes three 6 x 6 blocks with the state's own Jacobians,
    J_state = J_value * dT/dx + J_velocity * dV/dx + J_acceleration * dA/dx                                    (inertial.cpp:134-153)
    J_value[lin, ang]        = I_g R_sb hat(a_b_i) R_bw                                                          (:136)
    J_velocity[ang, ang]     = I_g R_sb,   J_velocity[lin, ang] = -I_g R_sb (2 hat(w_b) hat(t_bs) - hat(t_bs) hat(w_b))   (:141-142)
    J_acceleration[lin, ang] = -I_g R_sb hat(t_bs),   J_acceleration[lin, lin] = I_a R_sb                        (:148,150)
    J_T_bs = [[I_g hat(w_s), 0], [I_g hat(a_s), I_a R_sb F_a]]                                                    (:157-160)
    J_X_a block i (linear rows) = (I_a R_sb).col(i) F_a.row(i),   J_g_w (linear rows) = -I_a R_sb R_bw            (:191-193,198)
with a_b_i = A_lin - R_bw g_w, F_a = hat(w_b)^2 + hat(alpha_b), a_b_m, w_s, a_s as in :125-131. These formulas are TRANSCRIBED here with
mpmath matrices. The state's Jacobians dT/dx (world-frame rotation tangent), dV/dx (body rates), dA/dx (body angular acceleration, linear
acceleration) are EXTERNAL upstream; here they are MEASURED: central differences (step 1e-20 at 100 digits) of the spline of make_golden.py
— Cox-de Boor basis, numerical time derivatives — through the Ceres retractions. No analytic spline Jacobian appears in this file. The
tangent conventions are those of DESIGN.md §3 (world-frame left rotation tangent of T_wb, right / additive tangent of T_bs, Ceres local
coordinates for every block).
Run:  python tests/golden/make_inertial_literal_golden.py   (~10 minutes)
"""
import json
import os
import sys

import mpmath as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (spline, body rates, retractions, case generator; sets mp.dps = 100)
from hyperslam_amd.synthetic import SplitMix64  # noqa: E402

H = G.H


def hat(v):
    return mp.matrix([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def rot(q):
    """Rotation matrix of a unit quaternion (x, y, z, w): columns = images of the unit vectors."""
    cols = [G.qrot(q, [mp.mpf(i == c) for i in range(3)]) for c in range(3)]
    return mp.matrix([[cols[c][r] for c in range(3)] for r in range(3)])


def kinematics(P):
    q, _ = G.spline_pose(P["cps"], P["k"], P["stamp"])
    w, al, _, a_w = G.body_rates(P["cps"], P["k"], P["stamp"])
    return q, w, al, a_w


def literal_jacobians(P):
    k = P["k"]
    q, w, al, a_w = kinematics(P)
    R_wb = rot(q)
    R_bw = R_wb.T
    R_bs = rot(P["T_bs"][:4])
    R_sb = R_bs.T
    t = mp.matrix(P["T_bs"][4:7])
    I_g, I_a = mp.matrix(G.align_matrix(P["i_g"])), mp.matrix(G.align_matrix(P["i_a"]))
    X_a = mp.matrix([[P["X_a"][3 * c + r] for c in range(3)] for r in range(3)])  # column-major maps (inertial.cpp:48-49)
    g = mp.matrix(P["gravity"])
    wv, alv = mp.matrix(w), mp.matrix(al)
    A_lin = R_bw * mp.matrix(a_w)
    a_b_i = A_lin - R_bw * g                      # :125
    F_a = hat(w) * hat(w) + hat(al)               # :127
    a_b_m = mp.matrix([a_b_i[i] + (F_a * (X_a[:, i] + t))[i] for i in range(3)])  # :128
    I_g_R_sb, I_a_R_sb = I_g * R_sb, I_a * R_sb
    w_s, a_s = R_sb * wv, R_sb * a_b_m            # :130-131
    # ---- measured Jacobians of the state: columns = Ceres-local coordinates of the k control points (rotation 3, translation 3) ----
    n = 6 * k
    J_th, J_w, J_al, J_alin = mp.zeros(3, n), mp.zeros(3, n), mp.zeros(3, n), mp.zeros(3, n)
    for j in range(k):
        for part, kind in ((0, "cp_rot"), (3, "cp_trans")):
            for c in range(3):
                Pp, Pm = G.perturbed(P, (kind, j), c, H), G.perturbed(P, (kind, j), c, -H)
                qp, wp, alp, awp = kinematics(Pp)
                qm, wm, alm, awm = kinematics(Pm)
                col = 6 * j + part + c
                th = G.qlog(G.qmul(qp, G.qconj(qm)))  # world-frame left tangent: R(x+) = Exp(theta) R(x-)
                for r in range(3):
                    J_th[r, col] = th[r] / (2 * H)
                    J_w[r, col] = (wp[r] - wm[r]) / (2 * H)
                    J_al[r, col] = (alp[r] - alm[r]) / (2 * H)
                da = R_bw * mp.matrix([(awp[r] - awm[r]) / (2 * H) for r in range(3)])  # d A_lin / dx at fixed rotation (the rotation part is J_value's)
                for r in range(3):
                    J_alin[r, col] = da[r]
    # ---- transcribed blocks ----
    L_w = -(2 * hat(w) * hat(t) - hat(t) * hat(w))  # :142
    L_al = -hat(t)                                   # :148
    J_ang = I_g_R_sb * J_w                                                                   # :141
    J_lin = I_g_R_sb * (hat(a_b_i) * R_bw * J_th + L_w * J_w + L_al * J_al) + I_a_R_sb * J_alin  # :136,142,148 | :150
    J_state = [[J_ang[r, c] for c in range(n)] for r in range(3)] + [[J_lin[r, c] for c in range(n)] for r in range(3)]
    # extrinsics: right tangent tau of R_bs (R_bs Exp(tau)) against Ceres' local delta (R_bs <- Exp(2 delta) R_bs): tau = 2 R_sb delta; t_bs additive
    M = 2 * R_sb
    E_aa, E_la, E_ll = I_g * hat(w_s) * M, I_g * hat(a_s) * M, I_a_R_sb * F_a  # :157-160
    J_ext = [[E_aa[r, c] for c in range(3)] + [mp.mpf(0)] * 3 for r in range(3)] + [[E_la[r, c] for c in range(3)] + [E_ll[r, c] for c in range(3)] for r in range(3)]
    # gravity: ambient -I_a R_sb R_bw (:198) times the measured d g / d (local 2-vector) of Ceres' SphereManifold
    dg = mp.zeros(3, 2)
    for c in range(2):
        d = [mp.mpf(0)] * 2
        d[c] = H
        gp = G.plus_sphere(P["gravity"], d)
        d[c] = -H
        gm = G.plus_sphere(P["gravity"], d)
        for r in range(3):
            dg[r, c] = (gp[r] - gm[r]) / (2 * H)
    Gl = -(I_a_R_sb * R_bw) * dg
    J_grav = [[mp.mpf(0)] * 2 for _ in range(3)] + [[Gl[r, c] for c in range(2)] for r in range(3)]
    # accelerometer offsets X_a: block i (columns 3 i .. 3 i + 2) of the linear rows = (I_a R_sb).col(i) F_a.row(i) (:191-193); no gyro rows
    J_xa = [[mp.mpf(0)] * 9 for _ in range(3)] + [[I_a_R_sb[r, i] * F_a[i, c] for i in range(3) for c in range(3)] for r in range(3)]
    return {"J_state": J_state, "J_extrinsics": J_ext, "J_gravity": J_grav, "J_acc_offsets": J_xa}


def main():
    rng = SplitMix64(0x48595045 ^ 0x117E4A1)
    cases = []
    for k in (4, 6):
        for rep in range(16):
            variant = None if rep < 14 else ("u0" if rep == 14 else "cp_near_pi")
            P = G.make_case("inertial", k, rng, variant)
            P = {key: (G.tofloat(v) if not isinstance(v, int) else v) for key, v in P.items()}  # exactly representable inputs first
            Pm = {key: ([[mp.mpf(x) for x in r] for r in v] if isinstance(v, list) and isinstance(v[0], list)
                        else ([mp.mpf(x) for x in v] if isinstance(v, list) else (mp.mpf(v) if isinstance(v, float) else v)))
                  for key, v in P.items()}
            out = {"r": G.tofloat(G.res_inertial(Pm))}
            out.update({key: G.tofloat(val) for key, val in literal_jacobians(Pm).items()})
            cases.append({"type": "inertial", "variant": variant, "inputs": P, "outputs": out})
            print(k, rep, variant, "ok", flush=True)
    with open(os.path.join(HERE, "inertial_literal.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_inertial_literal_golden.py (mpmath, 100 digits): inertial.cpp:131-198 as written, measured state Jacobians",
                   "cases": cases}, f, indent=None, separators=(",", ":"))
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
