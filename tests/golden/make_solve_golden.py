#!/usr/bin/env python3
"""Generates tests/golden/solve.json — a 100-digit pin of the SOLVER level that shares no code with the oracle
or the HIP library: what `ceres::Solve` does to one small window, restated from Ceres' documented trust-region Levenberg-Marquardt
algorithm on top of the residual functions of make_golden.py.

Window: order 4, 9 control points (the first two constant: a frozen prefix, optimizer.cpp:319-328), 6 landmarks, two cameras, pixel +
bearing + pose-prior + inertial residual blocks, both bias splines and gravity free, non-identity IMU parameters (the libraries are put
into HS_INERTIAL_EXACT mode: every Jacobian here is a derivative). Per LM iteration, all in mpmath at 100 digits:
  stacked local Jacobian by central differences through the Ceres retractions (step 1e-20)  ->  loss correction (Huber / ScaledLoss,
  rho'' <= 0: rows and residuals scaled by sqrt(rho'))  ->  Jacobi scaling 1 / (1 + |column|) fixed at the first iteration  ->
  H = J'J, g = J'r  ->  LM diagonal D^2 = clamp(diag H, 1e-6, 1e32) / radius  ->  dense solve of (H + D^2) step = -g  (and, for the
  record, the landmark Schur complement of the same system)  ->  model cost change -(J step).(r + J step / 2)  ->  candidate
  x [+] step  ->  rho = (cost - cost_new) / model cost change  ->  accept iff rho > 1e-3, radius /= max(1/3, 1 - (2 rho - 1)^3),
  else radius /= f, f *= 2.
Unknown order of the pose side (what hs_reduced_system returns): 6 per control point (constant ones: unit diagonal, zero rhs),
3 per gyroscope bias point, 3 per accelerometer bias point, 2 for gravity.
Run:  python tests/golden/make_solve_golden.py   (about 10 minutes)

Second window (`python tests/golden/make_solve_golden.py visual` -> solve_visual.json): order 6, 12 control points (the first six constant:
the gauge is fixed), 6 landmarks far from their true positions, 24 pixel + 24 bearing blocks, no priors, no IMU — the kind of window on
which the HIP library linearises at the candidate point and keeps the records of the current point across a rejected step; the recorded
iterations contain rejected steps.
"""
import json
import os
import sys
import time

import mpmath as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import (H, RES, SplitMix64, perturbed, plus_quat, plus_sphere, qconj, qexp, qmul, qnorm, qrot, spline_pose,  # noqa: E402
                         tofloat)

VISUAL = len(sys.argv) > 1 and sys.argv[1] == "visual"
K, N_CP, DT, T0 = (6, 12, mp.mpf("0.1"), mp.mpf(0)) if VISUAL else (4, 9, mp.mpf("0.1"), mp.mpf(0))
KB, N_BIAS, BIAS_DT, BIAS_T0 = 4, 0 if VISUAL else 4, mp.mpf("1.0"), mp.mpf("-1.0")  # one bias segment [0, 1) over the window: four points, all observed
N_LM = 6
N_ITER = 6 if VISUAL else 4
N_CONST = 6 if VISUAL else 2   # leading constant control points (frozen prefix)
LM_NOISE = float(os.environ.get("HS_GOLDEN_LM_NOISE", "1.0"))  # visual window: how far the landmarks start from their true positions [m]
OUT_NAME = "solve_visual.json" if VISUAL else "solve.json"
HUBER = {"pixel": mp.mpf("0.5"), "bearing": mp.mpf("1.6e-3")}  # optimizer.cpp:204,226
INERTIAL_SCALE = mp.mpf("1.6e-5")                               # optimizer.cpp:267
OFF_BG = 6 * N_CP
OFF_BA = OFF_BG + 3 * N_BIAS
OFF_G = OFF_BA + 3 * N_BIAS
NP = OFF_G + (0 if VISUAL else 2)
NT = NP + 3 * N_LM


def to_json(x):
    if isinstance(x, dict):
        return {k: to_json(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [to_json(v) for v in x]
    return float(x) if isinstance(x, mp.mpf) else x


def f(x):
    """mp value of the double nearest to x: every input of the window is exactly representable."""
    return mp.mpf(float(x))


def build_window():
    rng = SplitMix64(0x534F4C56)

    def u(lo, hi):
        return f(rng.uniform(lo=lo, hi=hi))

    # ground-truth trajectory, then measurements at the truth (+ noise), then a perturbed starting point
    q = qnorm([u(-1, 1) for _ in range(4)])
    p = [u(-1, 1) for _ in range(3)]
    truth = []
    for j in range(N_CP):
        q = qnorm(qmul(q, qexp([u(-0.08, 0.08) for _ in range(3)])))
        p = [p[c] + u(-0.15, 0.15) + (mp.mpf("0.2") if c == 0 else 0) for c in range(3)]
        truth.append(q + p + [T0 + DT * j])
    def unit(q4):
        return [f(x) for x in qnorm(q4)]  # unit up to double rounding, like every stored quaternion

    cams = [{"T_bs": unit([u(-0.1, 0.1), u(-0.1, 0.1), u(-0.1, 0.1), f(1)]) + [u(-0.1, 0.1) for _ in range(3)],
             "intrinsics": [f(367.215), f(248.375), f(458.654), f(457.296)],
             "distortion": [f(-0.28340811), f(0.07395907), f(1.76187114e-05), f(0.00019359)]},
            {"T_bs": unit([u(-0.1, 0.1), u(-0.1, 0.1), u(-0.1, 0.1), f(1)]) + [f(0.11), u(-0.02, 0.02), u(-0.02, 0.02)],
             "intrinsics": [f(379.999), f(255.238), f(457.587), f(456.134)],
             "distortion": [f(-0.28368365), f(0.07451284), f(-0.00010473), f(-3.55590700e-05)]}]
    sensor_T = unit([u(-0.2, 0.2), u(-0.2, 0.2), u(-0.2, 0.2), f(1)]) + [u(-0.2, 0.2) for _ in range(3)]
    lo, hi = T0 + DT * ((K - 1) // 2), T0 + DT * (N_CP - K + (K - 1) // 2 + 1)  # valid stamps [0.1, 0.7)

    def stamp():
        return lo + (hi - lo) * u(0.02, 0.98)

    # landmarks: a point in front of camera 0 at a stamp in the middle of the window
    qm, pm = spline_pose(truth[2:2 + K], K, T0 + DT * (2 + (K - 1) // 2) + DT / 2)  # (the segment of control points 2 .. 2 + K - 1)
    lms_true = []
    for _ in range(N_LM):
        ps = [u(-1.2, 1.2), u(-0.8, 0.8), u(3.0, 7.0)]
        q_ws = qmul(qm, cams[0]["T_bs"][:4])
        p_ws = [a + b for a, b in zip(qrot(qm, cams[0]["T_bs"][4:7]), pm)]
        lms_true.append([f(a + b) for a, b in zip(qrot(q_ws, ps), p_ws)])
    imu = {"T_bs": unit([u(-0.05, 0.05), u(-0.05, 0.05), u(-0.05, 0.05), f(1)]) + [u(-0.05, 0.05) for _ in range(3)],
           "i_g": [f(1) + u(-0.02, 0.02) for _ in range(3)] + [u(-0.02, 0.02) for _ in range(3)],
           "i_a": [f(1) + u(-0.02, 0.02) for _ in range(3)] + [u(-0.02, 0.02) for _ in range(3)],
           "S_g": [u(-0.002, 0.002) for _ in range(9)], "X_a": [u(-0.02, 0.02) for _ in range(9)]}
    bias_true = {n: [[u(-0.05, 0.05) for _ in range(3)] + [BIAS_T0 + BIAS_DT * j] for j in range(N_BIAS)] for n in ("bias_g", "bias_a")}
    g_dir = qrot(qexp([u(-0.05, 0.05), u(-0.05, 0.05), f(0)]), [f(0), f(0), f(-1)])
    gravity_true = [f(mp.mpf("9.80665") * c) for c in g_dir]
    W = {"cps": [[f(v) for v in cp] for cp in truth], "lms": lms_true, "bias_g": bias_true["bias_g"], "bias_a": bias_true["bias_a"],
         "gravity": gravity_true, "cams": cams, "sensor_T": sensor_T, "imu": imu, "blocks": []}
    W["cps"] = [[f(x) for x in qnorm(cp[:4])] + cp[4:] for cp in W["cps"]]

    def first_cp(t):
        return int(mp.floor((t - T0) / DT)) - (K - 1) // 2

    def first_bias(t):
        return int(mp.floor((t - BIAS_T0) / BIAS_DT)) - (KB - 1) // 2

    def add(ftype, t, lm=None, cam=None, meas=None):
        b = {"type": ftype, "stamp": t, "first": first_cp(t), "lm": lm, "cam": cam, "meas": meas, "fb": first_bias(t) if ftype == "inertial" else None}
        if meas is None:  # measurement = prediction at the truth (the residual function with a zero measurement) + noise, rounded to doubles
            zero = {"pixel": [f(0)] * 2, "prior": None, "inertial": [f(0)] * 6}.get(ftype)
            if ftype == "bearing":
                qw, pw = spline_pose(W["cps"][b["first"]:b["first"] + K], K, t)
                T = cams[cam]["T_bs"]
                v = qrot(qconj(T[:4]), [a - c for a, c in zip(qrot(qconj(qw), [a - c for a, c in zip(W["lms"][lm], pw)]), T[4:7])])
                v = [v[c] + u(-2e-3, 2e-3) * mp.sqrt(sum(x * x for x in v)) for c in range(3)]
                n = mp.sqrt(sum(x * x for x in v))
                b["meas"] = [f(x / n) for x in v]
            elif ftype == "prior":
                qw, pw = spline_pose(W["cps"][b["first"]:b["first"] + K], K, t)
                q_ws = qmul(qmul(qw, sensor_T[:4]), qexp([u(-0.01, 0.01) for _ in range(3)]))
                p_ws = [a + c + u(-0.02, 0.02) for a, c in zip(qrot(qw, sensor_T[4:7]), pw)]
                b["meas"] = [f(x) for x in qnorm(q_ws) + p_ws]
            else:
                b["meas"] = zero
                pred = RES[ftype](local_problem(W, b))
                noise = {"pixel": 0.4, "inertial": 0.02}[ftype]
                b["meas"] = [f(x + u(-noise, noise)) for x in pred]
        W["blocks"].append(b)

    pairs = 4 if VISUAL else 2
    for lm in range(3):           # pixel factors on landmarks 0..2
        for _ in range(pairs):
            t = stamp()
            add("pixel", t, lm=lm, cam=0)
            add("pixel", t, lm=lm, cam=1)
    W["blocks"][3]["meas"][0] = f(W["blocks"][3]["meas"][0] + 25)  # one gross outlier (Huber stays active at the solution)
    for lm in range(3, 6):        # bearing factors on landmarks 3..5
        for _ in range(pairs):
            t = stamp()
            add("bearing", t, lm=lm, cam=0)
            add("bearing", t, lm=lm, cam=1)
    if VISUAL:  # starting point: free control points moved a little, landmarks moved a lot (the first steps overshoot and are rejected)
        for j in range(N_CONST, N_CP):
            W["cps"][j][:4] = [f(x) for x in qnorm(plus_quat(W["cps"][j][:4], [u(-0.01, 0.01) for _ in range(3)]))]
            for c in range(3):
                W["cps"][j][4 + c] = f(W["cps"][j][4 + c] + u(-0.05, 0.05))
        W["lms"] = [[f(x + u(-LM_NOISE, LM_NOISE)) for x in lm] for lm in W["lms"]]
        W["cp_constant"] = [1] * N_CONST + [0] * (N_CP - N_CONST)
        return W
    # pose priors (unit weight), two of them late in the last segment: the newest control point only carries the basis weight u^3 / 6 of
    # the residuals of that segment, and without them its block is conditioned like 1e-10 (it still is the weakest block, as in every
    # sliding window; the LM diagonal is what keeps the newest control point in place)
    for frac in ("0.04", "0.37", "0.61", "0.93", "0.985"):
        add("prior", lo + (hi - lo) * f(frac))
    for i in range(8):
        add("inertial", lo + (hi - lo) * (mp.mpf(i) + u(0.1, 0.9)) / 8)
    # starting point: the truth moved away (control points 0, 1 stay: they are constant)
    for j in range(2, N_CP):
        W["cps"][j][:4] = [f(x) for x in qnorm(plus_quat(W["cps"][j][:4], [u(-0.003, 0.003) for _ in range(3)]))]
        for c in range(3):
            W["cps"][j][4 + c] = f(W["cps"][j][4 + c] + u(-0.01, 0.01))
    W["lms"] = [[f(x + u(-0.03, 0.03)) for x in lm] for lm in W["lms"]]
    for n in ("bias_g", "bias_a"):
        W[n] = [[f(x + u(-0.01, 0.01)) for x in b[:3]] + [b[3]] for b in W[n]]
    W["gravity"] = [f(x) for x in plus_sphere(W["gravity"], [u(-0.01, 0.01), u(-0.01, 0.01)])]
    W["cp_constant"] = [1, 1] + [0] * (N_CP - 2)
    return W


def local_problem(W, b):
    P = {"k": K, "cps": [list(cp) for cp in W["cps"][b["first"]:b["first"] + K]], "stamp": b["stamp"], "meas": b["meas"]}
    if b["type"] in ("pixel", "bearing"):
        cam = W["cams"][b["cam"]]
        P.update(T_bs=list(cam["T_bs"]), intrinsics=cam["intrinsics"], distortion=cam["distortion"], landmark=list(W["lms"][b["lm"]]))
    elif b["type"] == "prior":
        P["T_bs"] = list(W["sensor_T"])
    else:
        imu = W["imu"]
        P.update(T_bs=list(imu["T_bs"]), kb=KB, i_g=imu["i_g"], i_a=imu["i_a"], S_g=imu["S_g"], X_a=imu["X_a"], gravity=list(W["gravity"]),
                 bias_g=[list(x) for x in W["bias_g"][b["fb"]:b["fb"] + KB]], bias_a=[list(x) for x in W["bias_a"][b["fb"]:b["fb"] + KB]])
    return P


def rho(ftype, s):
    """(rho(s), rho'(s)) of the block's loss."""
    if ftype in HUBER:
        a = HUBER[ftype]
        return (s, mp.mpf(1)) if s <= a * a else (2 * a * mp.sqrt(s) - a * a, a / mp.sqrt(s))
    if ftype == "inertial":
        return INERTIAL_SCALE * s, INERTIAL_SCALE
    return s, mp.mpf(1)


def block_columns(W, b):
    """[(perturbation block, column of that block, global column)] of every non-constant local coordinate the block reads."""
    cols = []
    for j in range(K):
        cp = b["first"] + j
        if W["cp_constant"][cp]:
            continue
        cols += [(("cp_rot", j), c, 6 * cp + c) for c in range(3)] + [(("cp_trans", j), c, 6 * cp + 3 + c) for c in range(3)]
    if b["lm"] is not None:
        cols += [(("landmark", 0), c, NP + 3 * b["lm"] + c) for c in range(3)]
    if b["type"] == "inertial":
        for j in range(KB):
            cols += [(("bias_g", j), c, OFF_BG + 3 * (b["fb"] + j) + c) for c in range(3)]
            cols += [(("bias_a", j), c, OFF_BA + 3 * (b["fb"] + j) + c) for c in range(3)]
        cols += [(("gravity", 0), c, OFF_G + c) for c in range(2)]
    return cols


def evaluate(W, with_jacobian=True):
    """cost, stacked loss-corrected residual vector and (sparse, per block) loss-corrected local Jacobian."""
    cost = mp.mpf(0)
    rows = []  # (residual value, {global column: derivative})
    for b in W["blocks"]:
        P = local_problem(W, b)
        r = RES[b["type"]](P)
        s = sum(x * x for x in r)
        value, d1 = rho(b["type"], s)
        cost += value / 2
        w = mp.sqrt(d1)
        entries = [dict() for _ in r]
        if with_jacobian:
            for block, c, g in block_columns(W, b):
                rp = RES[b["type"]](perturbed(P, block, c, H))
                rm = RES[b["type"]](perturbed(P, block, c, -H))
                for i in range(len(r)):
                    entries[i][g] = w * (rp[i] - rm[i]) / (2 * H)
        rows += [(w * r[i], entries[i]) for i in range(len(r))]
    return cost, rows


def retract(W, delta):
    Q = dict(W)
    Q["cps"] = [list(cp) for cp in W["cps"]]
    for j in range(N_CP):
        if W["cp_constant"][j]:
            continue
        Q["cps"][j][:4] = plus_quat(W["cps"][j][:4], delta[6 * j:6 * j + 3])
        for c in range(3):
            Q["cps"][j][4 + c] += delta[6 * j + 3 + c]
    Q["lms"] = [[W["lms"][l][c] + delta[NP + 3 * l + c] for c in range(3)] for l in range(N_LM)]
    Q["bias_g"] = [[W["bias_g"][j][c] + delta[OFF_BG + 3 * j + c] for c in range(3)] + [W["bias_g"][j][3]] for j in range(N_BIAS)]
    Q["bias_a"] = [[W["bias_a"][j][c] + delta[OFF_BA + 3 * j + c] for c in range(3)] + [W["bias_a"][j][3]] for j in range(N_BIAS)]
    Q["gravity"] = W["gravity"] if VISUAL else plus_sphere(W["gravity"], delta[OFF_G:OFF_G + 2])
    return Q


def ambient(W):
    return [x for cp in W["cps"] for x in cp] + [x for lm in W["lms"] for x in lm] + [x for n in ("bias_g", "bias_a") for b in W[n] for x in b] + list(W["gravity"])


def ambient_free(W):
    """Ambient coordinates of the non-constant parameter blocks (x_norm of Ceres' reduced program)."""
    return ([x for j, cp in enumerate(W["cps"]) if not W["cp_constant"][j] for x in cp] + [x for lm in W["lms"] for x in lm]
            + [x for n in ("bias_g", "bias_a") for b in W[n] for x in b] + list(W["gravity"]))


def state_json(W):
    return {"control_points": tofloat(W["cps"]), "landmarks": tofloat(W["lms"]), "bias_g": tofloat(W["bias_g"]), "bias_a": tofloat(W["bias_a"]),
            "gravity": tofloat(W["gravity"])}


def main():
    t_start = time.time()
    W = build_window()
    out = {"generator": "tests/golden/make_solve_golden.py (mpmath, 100 digits)", "order": K, "t0": float(T0), "dt": float(DT),
           "cp_constant": W["cp_constant"], "cameras": to_json(W["cams"]), "sensor_T_bs": to_json(W["sensor_T"]), "imu": None if VISUAL else to_json(W["imu"]),
           "bias_order": KB, "bias_t0": float(BIAS_T0), "bias_dt": float(BIAS_DT),
           "blocks": [{"type": b["type"], "stamp": float(b["stamp"]), "landmark": b["lm"], "camera": b["cam"], "meas": tofloat(b["meas"])} for b in W["blocks"]],
           "initial": state_json(W), "iterations": []}
    if os.environ.get("HS_GOLDEN_DUMP_ONLY"):  # (tuning the starting point: the window alone, for a look at it through a library)
        with open(os.environ["HS_GOLDEN_DUMP_ONLY"], "w") as fh:
            json.dump(out, fh, separators=(",", ":"))
        return
    radius, decrease = mp.mpf(10) ** 4, mp.mpf(2)
    cost, rows = evaluate(W)
    out["initial_cost"] = float(cost)
    scale = None
    for it in range(1, N_ITER + 1):
        n_rows = len(rows)
        J = mp.zeros(n_rows, NT)
        r = mp.zeros(n_rows, 1)
        for i, (ri, e) in enumerate(rows):
            r[i] = ri
            for g, v in e.items():
                J[i, g] = v
        col2 = [sum(J[i, c] ** 2 for i in range(n_rows)) for c in range(NT)]
        active = [c for c in range(NT) if col2[c] > 0]
        if scale is None:  # TrustRegionMinimizer: jacobi_scaling computed once, at the first linearisation
            scale = [1 / (1 + mp.sqrt(col2[c])) for c in range(NT)]
        Js = mp.zeros(n_rows, NT)
        for i in range(n_rows):
            for c in active:
                Js[i, c] = J[i, c] * scale[c]
        Hm = Js.T * Js
        g = Js.T * r
        gradient_max = max(abs(sum(J[i, c] * r[i] for i in range(n_rows))) for c in active)  # local coordinates, unscaled
        d2 = [min(max(Hm[c, c], mp.mpf("1e-6")), mp.mpf("1e32")) / radius for c in range(NT)]
        na = len(active)
        A = mp.zeros(na, na)
        rhs = mp.zeros(na, 1)
        for a, ca in enumerate(active):
            for b_, cb in enumerate(active):
                A[a, b_] = Hm[ca, cb]
            A[a, a] += d2[ca]
            rhs[a] = -g[ca]
        sol = mp.cholesky_solve(A, rhs)
        import numpy as np
        An = np.array([[float(A[a, b_]) for b_ in range(na)] for a in range(na)])
        dn = 1 / np.sqrt(np.diag(An))
        ev = np.linalg.eigvalsh(An * dn[:, None] * dn[None, :])  # Cholesky is insensitive to diagonal scaling: this is the number that matters
        print("  condition of the damped system with unit diagonal: %.3g" % (ev[-1] / ev[0]), flush=True)
        step_s = [mp.mpf(0)] * NT
        for a, ca in enumerate(active):
            step_s[ca] = sol[a]
        delta = [scale[c] * step_s[c] for c in range(NT)]
        # reduced system: Schur complement of the landmark blocks, pose-side order, inactive coordinates = identity rows
        S = mp.zeros(NP, NP)
        gr = mp.zeros(NP, 1)
        act_p = [c for c in active if c < NP]
        for a in act_p:
            for b_ in act_p:
                S[a, b_] = Hm[a, b_]
            S[a, a] += d2[a]
            gr[a] = g[a]
        for l in range(N_LM):
            idx = [NP + 3 * l + c for c in range(3)]
            V = mp.matrix(3, 3)
            for a in range(3):
                for b_ in range(3):
                    V[a, b_] = Hm[idx[a], idx[b_]]
                V[a, a] += d2[idx[a]]
            Vi = V ** -1
            Wl = mp.matrix(len(act_p), 3)
            for a, ca in enumerate(act_p):
                for b_ in range(3):
                    Wl[a, b_] = Hm[ca, idx[b_]]
            WV = Wl * Vi
            gl = mp.matrix([g[i] for i in idx])
            corr = WV * Wl.T
            cg = WV * gl
            for a, ca in enumerate(act_p):
                for b_, cb in enumerate(act_p):
                    S[ca, cb] -= corr[a, b_]
                gr[ca] -= cg[a]
        for c in range(NP):
            if c not in act_p:
                S[c, c] = 1
        # consistency of the two routes: the Schur-reduced solve reproduces the pose part of the full step
        chk = mp.cholesky_solve(S, -gr)
        assert max(abs(chk[c] - step_s[c]) for c in range(NP)) < mp.mpf(10) ** -60
        Jd = J * mp.matrix(delta)
        model_change = -sum(Jd[i] * (r[i] + Jd[i] / 2) for i in range(n_rows))
        cand = retract(W, delta)
        cand_cost, _ = evaluate(cand, with_jacobian=False)
        x0, x1 = ambient(W), ambient(cand)
        step_norm = mp.sqrt(sum((a - b_) ** 2 for a, b_ in zip(x0, x1)))
        quality = (cost - cand_cost) / model_change
        rec = {"iteration": it, "radius_before": float(radius), "jacobi_scale": tofloat(scale), "gradient": tofloat([g[c] for c in range(NT)]),
               "lm_diagonal": tofloat(d2), "reduced_S": tofloat([[S[a, b_] for b_ in range(NP)] for a in range(NP)]),
               "reduced_g": tofloat([gr[c] for c in range(NP)]), "step_scaled": tofloat(step_s), "step": tofloat(delta),
               "model_cost_change": float(model_change), "candidate_cost": float(cand_cost), "cost_change": float(cost - cand_cost),
               "relative_decrease": float(quality), "step_norm": float(step_norm), "gradient_max_norm_before": float(gradient_max),
               "candidate": state_json(cand)}
        assert model_change > 0
        if os.environ.get("HS_GOLDEN_PROBE"):  # (tuning the starting point: print the decisions only)
            print("probe: iteration", it, "rho", mp.nstr(quality, 6), flush=True)
        # the convergence tests TrustRegionMinimizer runs before it looks at the step quality (parameter, then function tolerance)
        x_norm = mp.sqrt(sum(v * v for v in ambient_free(W)))
        rec["x_norm"] = float(x_norm)
        rec["parameter_tolerance_reached"] = int(step_norm <= mp.mpf("1e-8") * (x_norm + mp.mpf("1e-8")))
        rec["function_tolerance_reached"] = int(abs(cost - cand_cost) <= mp.mpf("1e-6") * cost)
        if rec["parameter_tolerance_reached"] or rec["function_tolerance_reached"]:
            rec["terminated"] = 1
            out["iterations"].append(rec)
            print("iteration", it, "converged: step", mp.nstr(step_norm, 6), "cost change", mp.nstr(cost - cand_cost, 6), flush=True)
            break
        if quality > mp.mpf("1e-3"):
            W = cand
            cost, rows = evaluate(W)
            radius = min(mp.mpf(10) ** 16, radius / max(mp.mpf(1) / 3, 1 - (2 * quality - 1) ** 3))
            decrease = mp.mpf(2)
            rec["step_is_successful"] = 1
            rec["cost"] = float(cost)
            n_rows = len(rows)
            rec["gradient_max_norm"] = float(max(abs(sum(e.get(c, 0) * ri for ri, e in rows)) for c in active))
        else:
            radius = radius / decrease
            decrease *= 2
            rec["step_is_successful"] = 0
            rec["cost"] = float(cand_cost)
            rec["gradient_max_norm"] = float(gradient_max)
        rec["radius"] = float(radius)
        rec["state"] = state_json(W)
        out["iterations"].append(rec)
        print("iteration", it, "cost", mp.nstr(cost, 12), "rho", mp.nstr(quality, 8), "radius", mp.nstr(radius, 8), "%.0f s" % (time.time() - t_start), flush=True)
    with open(os.path.join(HERE, OUT_NAME), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", OUT_NAME)


if __name__ == "__main__":
    main()
