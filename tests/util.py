"""Shared helpers of the test-suite."""
import json
import os

import numpy as np
import pytest

from hyperslam_amd import HS_BEARING, HS_INERTIAL, HS_INERTIAL_AS_REFERENCE, HS_INERTIAL_EXACT, HS_PIXEL, HS_PRIOR, Window

HERE = os.path.dirname(os.path.abspath(__file__))
TYPE_ID = {"pixel": HS_PIXEL, "bearing": HS_BEARING, "prior": HS_PRIOR, "inertial": HS_INERTIAL}


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def reference_vectors(name):
    """HS_REFERENCE_VECTORS=<dir>: outputs of the REAL reference on the inputs of tests/golden/<name> (written by tools/reference_dump.cpp where
    HyperSLAM builds; same case order, same schema). None when the variable is unset: the suite then checks against the 100-digit restatements
    of tests/golden/make_*.py — which is all this repository can do on its own ("parity unpinned")."""
    d = os.environ.get("HS_REFERENCE_VECTORS")
    if not d:
        return None
    with open(os.path.join(d, name)) as f:
        return json.load(f)


def _overlay_cases(cases, name, keys=None):
    ref = reference_vectors(name)
    if ref is None:
        return cases
    assert len(ref["cases"]) == len(cases), (name, len(ref["cases"]), len(cases))
    for c, r in zip(cases, ref["cases"]):
        if "outputs" in r:
            assert r["type"] == c["type"]
            c["outputs"], c["reference"] = r["outputs"], True
        else:
            assert r["kind"] == c["kind"] and r["ambient"] == c["ambient"] and r["tangent"] == c["tangent"]
            for k in (keys or r):
                if k in r:
                    c[k] = r[k]
            c["reference"] = True
    return cases


def golden_cases():
    with open(os.path.join(HERE, "golden", "factors.json")) as f:
        return _overlay_cases(json.load(f)["cases"], "factors.json")


def golden_cases_k5():
    """Order 5 (tests/golden/make_golden.py --order5): 48 cases, 12 per factor."""
    with open(os.path.join(HERE, "golden", "factors_k5.json")) as f:
        return _overlay_cases(json.load(f)["cases"], "factors_k5.json")


def golden_window(case) -> Window:
    """One-residual window reproducing a golden case (tests/golden/make_golden.py)."""
    P = case["inputs"]
    k = P["k"]
    cps = np.array(P["cps"], float)
    w = Window(order=k, t0=float(cps[0, 7]), dt=0.1, control_points=cps)
    st = np.array([P["stamp"]])
    t = case["type"]
    if t in ("pixel", "bearing"):
        w.cam_T_bs = np.array([P["T_bs"]])
        w.cam_intrinsics = np.array([P["intrinsics"]])
        w.cam_distortion = np.array([P["distortion"]])
        w.landmarks = np.array([P["landmark"]])
        z = np.zeros(1, np.int32)
        if t == "pixel":
            w.pixel_stamps, w.pixels, w.pixel_landmark, w.pixel_camera = st, np.array([P["meas"]]), z, z
        else:
            w.bearing_stamps, w.bearings, w.bearing_landmark, w.bearing_camera = st, np.array([P["meas"]]), z, z
    elif t == "prior":
        w.sensor_T_bs = np.array([P["T_bs"]])
        w.prior_stamps, w.prior_poses, w.prior_sensor = st, np.array([P["meas"]]), np.zeros(1, np.int32)
    else:
        bg, ba = np.array(P["bias_g"], float), np.array(P["bias_a"], float)
        w.imu = dict(T_bs=P["T_bs"], i_g=P["i_g"], i_a=P["i_a"], S_g=P["S_g"], X_a=P["X_a"], bias_order=P["kb"], bias_t0=float(bg[0, 3]), bias_dt=1.0,
                     bias_g=bg, bias_a=ba, bias_constant=False)
        w.gravity, w.gravity_constant = np.array(P["gravity"]), False
        w.inertial_stamps, w.inertial_measurements = st, np.array([P["meas"]])
    return w


SENSOR_KEYS = ("J_extrinsics", "J_intrinsics", "J_distortion", "J_gyro_intrinsics", "J_acc_intrinsics", "J_gyro_sensitivity", "J_acc_offsets")


def check_against_golden(problem, case, tol):
    """Compares an un-robustified linearisation (oracle or HIP) of a golden window with the 100-digit vectors: residual, state /
    landmark / bias / gravity columns and every sensor-block column. The vectors are derivatives of the prediction, i.e. the EXACT form
    of the inertial Jacobian; the `identity` inertial cases (I_g = I_a = I, S_g = X_a = 0) are also checked in the default
    as-written-upstream form, which coincides there."""
    out = case["outputs"]
    modes = [None]
    if case["type"] == "inertial":
        modes = [HS_INERTIAL_EXACT] + ([HS_INERTIAL_AS_REFERENCE] if case.get("variant") == "identity" else [])
        if case.get("reference"):  # vectors of the reference itself (HS_REFERENCE_VECTORS): its Jacobian is the one written in inertial.cpp
            modes = [HS_INERTIAL_AS_REFERENCE]
    # a bearing 1e-5 rad off its measurement: the direction of the Jacobian is conditioned like 1 / angle
    tol = tol * 1e3 if case.get("variant") == "small_angle" else tol
    errs = {}
    for mode in modes:
        if mode is not None:
            problem.set_inertial_jacobian(mode)
        lin = problem.linearize(TYPE_ID[case["type"]], robustify=False, sensor_blocks=True)
        errs = {"r": rel(lin["r"][0], out["r"]), "J_state": rel(lin["J_state"][0], out["J_state"])}
        for key in ("J_landmark", "J_bias_g", "J_bias_a", "J_gravity") + SENSOR_KEYS:
            if key in out:
                ref = np.asarray(out[key], float)
                # columns that are structurally zero (e.g. d r_rot / d t_bs) compare absolutely against the block's scale
                errs[key] = float(np.abs(np.asarray(lin[key][0]) - ref).max() / max(1e-300, np.abs(ref).max(), np.abs(np.asarray(out["J_state"])).max() * 1e-3))
        bad = {k: v for k, v in errs.items() if not v < tol}
        assert not bad, (case["type"], case.get("variant"), case["inputs"]["k"], mode, bad)
    if case["type"] == "inertial":
        problem.set_inertial_jacobian(HS_INERTIAL_AS_REFERENCE)
    return errs


def literal_inertial_cases():
    with open(os.path.join(HERE, "golden", "inertial_literal.json")) as f:
        return _overlay_cases(json.load(f)["cases"], "inertial_literal.json")


def check_against_literal_golden(problem, case, tol):
    """The DEFAULT mode of the libraries (HS_INERTIAL_AS_REFERENCE: the inertial Jacobian as written upstream, inertial.cpp:131-198) against
    the 100-digit transcription of that text at IMU parameters where it is not the derivative of the prediction
    (tests/golden/make_inertial_literal_golden.py): residual and the four blocks in which the two forms differ — state, extrinsics, gravity,
    accelerometer offsets. The same inputs in the exact mode must DIFFER from these vectors (the case is not vacuous)."""
    out = case["outputs"]
    problem.set_inertial_jacobian(HS_INERTIAL_AS_REFERENCE)
    lin = problem.linearize(HS_INERTIAL, robustify=False, sensor_blocks=True)
    scale = np.abs(np.asarray(out["J_state"])).max()
    errs = {"r": rel(lin["r"][0], out["r"])}
    for key in ("J_state", "J_extrinsics", "J_gravity", "J_acc_offsets"):
        ref = np.asarray(out[key], float)
        errs[key] = float(np.abs(np.asarray(lin[key][0]) - ref).max() / max(np.abs(ref).max(), 1e-3 * scale))
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, (case["inputs"]["k"], case.get("variant"), bad)
    problem.set_inertial_jacobian(HS_INERTIAL_EXACT)
    exact = problem.linearize(HS_INERTIAL, robustify=False, sensor_blocks=True)
    problem.set_inertial_jacobian(HS_INERTIAL_AS_REFERENCE)
    assert rel(exact["J_state"][0], out["J_state"]) > 1e-4  # the two forms are different matrices at these parameters
    return errs


def manifold_cases():
    with open(os.path.join(HERE, "golden", "manifolds.json")) as f:
        return _overlay_cases(json.load(f)["cases"], "manifolds.json", keys=("plus", "jacobian", "minus", "minus_jacobian"))


def check_manifolds_against_golden(problem, tol):
    """Manifold::Plus / PlusJacobian / Minus / MinusJacobian (hs_manifold_*) of a library against tests/golden/manifolds.json, batched per kind."""
    cases = manifold_cases()
    groups = {}
    for c in cases:
        groups.setdefault((c["kind"], c["ambient"]), []).append(c)
    assert {k for k, _ in groups} == {0, 1, 2, 3, 4, 5}
    worst = 0.0
    for (kind, ambient), cs in groups.items():
        x = np.array([c["x"] for c in cs], float)
        d = np.array([c["delta"] for c in cs], float).reshape(len(cs), -1)
        out = problem.manifold_plus(kind, x, d)
        jac = problem.manifold_plus_jacobian(kind, x)
        y = np.array([c["plus"] for c in cs], float)
        back = problem.manifold_minus(kind, y, x)
        mjac = problem.manifold_minus_jacobian(kind, x)
        for i, c in enumerate(cs):
            if c["tangent"]:
                # Minus: absolute error against max(1, |delta|) (a delta of 1e-9 is recovered from a y that carries 1e-16 of rounding)
                e = float(np.abs(back[i] - np.array(c["minus"])).max()) / max(1.0, float(np.abs(np.array(c["minus"])).max()))
                assert e <= 4 * tol, (kind, "minus", c["x"], c["delta"], e)
                e = rel(mjac[i], c["minus_jacobian"])
                # SphereManifold, sigma = x0^2 + x1^2 <= eps but not zero: Ceres' Householder reflection is only approximately orthogonal
                # there, Plus(x, 0) sits 2 sqrt(sigma) / |x| away from x, and Ceres' closed-form MinusJacobian (rows of H / |x|, what the
                # libraries return) differs from the derivative of the inverse of Plus (what the vectors hold) by that amount (4e-9 here).
                degenerate = kind == 4 and 0.0 < c["x"][0] ** 2 + c["x"][1] ** 2 <= 2.220446049250313e-16
                assert e <= (1e-8 if degenerate else tol), (kind, "minus_jacobian", c["x"], e)
                worst = max(worst, 0.0 if degenerate else e)
                # the two Jacobians are inverse to each other on the tangent space (Ceres' own manifold test): MinusJacobian * PlusJacobian = I
                assert np.abs(mjac[i] @ jac[i] - np.eye(c["tangent"])).max() <= 1e-13, (kind, c["x"])
            assert mjac[i].shape == (c["tangent"], ambient)
            e = rel(out[i], c["plus"])
            assert e <= tol, (kind, c["x"], c["delta"], e)
            worst = max(worst, e)
            if c["tangent"]:
                e = rel(jac[i], c["jacobian"])
                assert e <= tol, (kind, c["x"], e)
                worst = max(worst, e)
            assert jac[i].shape == (ambient, c["tangent"])
    return worst


def check_tracks_against_golden(lib, tol):
    """hs_process_tracks of a library against tests/golden/tracks.json (exact undistortion root, closest-point midpoint, spline
    pose; tests/golden/make_tracks_golden.py). Returns the worst relative error."""
    from hyperslam_amd import Problem
    with open(os.path.join(HERE, "golden", "tracks.json")) as f:
        d = json.load(f)
    cps = np.array(d["cps"], float)
    w = Window(order=d["k"], t0=float(cps[0, 7]), dt=0.1, control_points=cps, cam_T_bs=np.array(d["cam_T_bs"]),
               cam_intrinsics=np.array(d["intrinsics"]), cam_distortion=np.array(d["distortion"]))
    with Problem(w, lib=lib) as p:
        b0, b1, pw = p.process_tracks(d["stamp"], d["pixels0"], d["pixels1"])
    errs = (rel(b0, d["bearings0"]), rel(b1, d["bearings1"]), rel(pw, d["positions_w"]))
    assert max(errs) <= tol, errs
    return max(errs)


def solve_golden(name="solve.json"):
    """tests/golden/solve.json / solve_visual.json (tests/golden/make_solve_golden.py): the window and, per LM iteration, the 100-digit
    solver quantities. `name` may be a path."""
    with open(name if os.path.isabs(name) else os.path.join(HERE, "golden", name)) as f:
        d = json.load(f)
    ref = None if os.path.isabs(name) else reference_vectors(name)
    if ref is not None:  # the reference's own ceres::Solve on this window (tools/reference_dump.cpp): its records and states replace the restatement's
        assert len(ref["iterations"]) == len(d["iterations"])
        d["initial_cost"], d["reference"] = ref["initial_cost"], True
        for it, r in zip(d["iterations"], ref["iterations"]):
            for k in ("reduced_S", "reduced_g", "radius_before", "gradient_max_norm_before"):
                it.pop(k, None)  # quantities Ceres does not report
            it.update(r)
    blocks = d["blocks"]

    def table(ftype, key, width=None):
        rows = [b[key] for b in blocks if b["type"] == ftype]
        a = np.array(rows, float if key in ("stamp", "meas") else np.int32)
        return a.reshape(len(rows), width) if width else a

    ini = d["initial"]
    imu = None
    if d["imu"] is not None:
        imu = dict(d["imu"])
        imu.update(bias_order=d["bias_order"], bias_t0=d["bias_t0"], bias_dt=d["bias_dt"], bias_g=np.array(ini["bias_g"]), bias_a=np.array(ini["bias_a"]),
                   bias_constant=False)
    w = Window(order=d["order"], t0=d["t0"], dt=d["dt"], control_points=np.array(ini["control_points"]), cp_constant=np.array(d["cp_constant"], np.uint8),
               cam_T_bs=np.array([c["T_bs"] for c in d["cameras"]]), cam_intrinsics=np.array([c["intrinsics"] for c in d["cameras"]]),
               cam_distortion=np.array([c["distortion"] for c in d["cameras"]]), sensor_T_bs=np.array([d["sensor_T_bs"]]),
               landmarks=np.array(ini["landmarks"]),
               pixel_stamps=table("pixel", "stamp"), pixels=table("pixel", "meas", 2), pixel_landmark=table("pixel", "landmark"), pixel_camera=table("pixel", "camera"),
               bearing_stamps=table("bearing", "stamp"), bearings=table("bearing", "meas", 3), bearing_landmark=table("bearing", "landmark"),
               bearing_camera=table("bearing", "camera"),
               prior_stamps=table("prior", "stamp"), prior_poses=table("prior", "meas", 7), prior_sensor=np.zeros(len(table("prior", "stamp")), np.int32),
               inertial_stamps=table("inertial", "stamp"), inertial_measurements=table("inertial", "meas", 6),
               imu=imu, gravity=np.array(ini["gravity"]), gravity_constant=False)
    return d, w


def check_solver_against_golden(lib, tol_forward, tol_state, name="solve.json"):
    """The solver level of a library (a-11: cost, reduced normal equations, LM step, step quality, accept / reject, radius, and the state
    after every iteration) against tests/golden/solve.json. `tol_forward`: quantities that are evaluated (cost, reduced system, gradient);
    `tol_state`: quantities that went through the linear solve (steps, step quality, the state). Returns the worst errors seen."""
    from hyperslam_amd import HS_INERTIAL_EXACT, Problem
    d, w = solve_golden(name)
    its = d["iterations"]
    has_imu = w.imu is not None
    worst = {"forward": 0.0, "state": 0.0}

    def fwd(a, b, what):
        e = rel(a, b)
        assert e <= tol_forward, (what, e)
        worst["forward"] = max(worst["forward"], e)

    def sta(a, b, what, scale=1.0):
        e = float(np.abs(np.asarray(a, float) - np.asarray(b, float)).max()) / max(scale, float(np.abs(np.asarray(b, float)).max()))
        assert e <= tol_state, (what, e)
        worst["state"] = max(worst["state"], e)

    from hyperslam_amd import HS_INERTIAL_AS_REFERENCE
    # every vector of the restatement is a derivative (the IMU parameters are not at the identity point): exact mode; the reference's own
    # solve (HS_REFERENCE_VECTORS) runs on the Jacobian written in inertial.cpp: the libraries' default mode
    mode = HS_INERTIAL_AS_REFERENCE if d.get("reference") else HS_INERTIAL_EXACT
    with Problem(w, lib=lib) as p:
        if has_imu:
            p.set_inertial_jacobian(mode)
        fwd(p.cost(), d["initial_cost"], "initial cost")
        if "reduced_S" in its[0]:
            S, g = p.reduced_system(its[0]["radius_before"])
            fwd(S, its[0]["reduced_S"], "reduced system, first iteration")
            fwd(g, its[0]["reduced_g"], "reduced gradient, first iteration")
        for n in range(1, len(its) + 1):  # the state after n iterations, from the same starting point every time
            p.upload(w)
            if has_imu:
                p.set_inertial_jacobian(mode)
            s = p.solve(n)
            assert s["num_iterations"] == n
            golden_state = its[n - 1]["state"]
            sta(p.control_points(), golden_state["control_points"], f"control points after {n}")
            sta(p.landmarks(), golden_state["landmarks"], f"landmarks after {n}")
            if not has_imu:
                continue
            bg, ba = p.bias()
            # the bias points carry the weakest blocks of the window (inertial loss scale 1.6e-5): absolute scale = the step they took
            bias_scale = max(1.0, float(np.abs(np.array(its[n - 1]["step"])).max()))
            sta(bg, golden_state["bias_g"], f"gyroscope bias after {n}", bias_scale)
            sta(ba, golden_state["bias_a"], f"accelerometer bias after {n}", bias_scale)
            sta(p.gravity(), golden_state["gravity"], f"gravity after {n}")
        assert s["initial_cost"] == pytest.approx(d["initial_cost"], rel=tol_forward)
        assert s["num_successful_steps"] == sum(r["step_is_successful"] for r in its)
        for rec, gold in zip(s["iterations"][1:], its):
            assert rec["step_is_valid"] == 1 and rec["step_is_successful"] == gold["step_is_successful"], (gold["iteration"], rec)
            sta(rec["cost"], gold["cost"], "cost after the step")
            sta(rec["cost_change"], gold["cost_change"], "cost change", scale=abs(gold["cost"]))
            sta(rec["relative_decrease"], gold["relative_decrease"], "step quality")
            sta(rec["radius"], gold["radius"], "trust-region radius")
            sta(rec["step_norm"], gold["step_norm"], "step norm")
            # the last record of a solve keeps the gradient from before its step (include/hyperslam_hip.h): the product library does not
            # linearise at the final point; the oracle, like Ceres, does
            last = gold is its[-1]
            if last and lib.prefix == "hs_" and "gradient_max_norm_before" not in gold:
                continue  # (reference vectors: Ceres reports the gradient at the final point only, which the product library does not evaluate)
            sta(rec["gradient_max_norm"], gold["gradient_max_norm_before"] if (last and lib.prefix == "hs_") else gold["gradient_max_norm"],
                "gradient max norm (local coordinates)")
    return worst


def check_knot_uniformity_is_enforced(lib):
    """hs_set_spline refuses a control-point table whose stamps are not t0 + j dt (a hole, a shifted knot) and accepts stamps that differ
    from it in the last bits (accumulated sums, /root/reference/internal/hyper/optimizers/abstract.cpp:128). Same rule in both libraries."""
    import numpy as np
    import hyperslam_amd as ha
    from hyperslam_amd import synthetic
    w = synthetic.small_visual(order=4, n_cp=14, n_landmarks=20, obs_pairs=2)
    with ha.Problem(w, lib=lib) as p:
        c0 = p.cost()
    # stamps accumulated the way upstream extends the state: t += dt (last-bit differences from t0 + j dt)
    acc = w.control_points.copy()
    t = acc[0, 7]
    for j in range(1, len(acc)):
        t = t + w.dt
        acc[j, 7] = t
    assert not np.array_equal(acc[:, 7], w.control_points[:, 7])
    w2 = synthetic.small_visual(order=4, n_cp=14, n_landmarks=20, obs_pairs=2)
    w2.control_points = acc
    with ha.Problem(w2, lib=lib) as p:
        assert abs(p.cost() - c0) <= 1e-12 * c0  # (the oracle, like the reference, reads the knots from the rows; the product library derives them)
    # epoch-scale stamps made from integer nanoseconds (one ulp of 1.7e9 s is 2.4e-7 s: the tolerance has to follow the magnitude of the
    # stamps, not only the knot spacing): accepted
    w4 = synthetic.small_visual(order=4, n_cp=14, n_landmarks=20, obs_pairs=2)
    epoch_ns = 1_700_000_000_000_000_000
    t0_ns = epoch_ns + int(round(w4.t0 * 1e9))
    w4.control_points = w4.control_points.copy()
    w4.control_points[:, 7] = [(t0_ns + j * int(round(w4.dt * 1e9))) * 1e-9 for j in range(len(w4.control_points))]
    lo, hi = w4.t0 + w4.dt, w4.t0 + (len(w4.control_points) - 2) * w4.dt  # valid range of an order-4 spline; stamps kept off its ends
    w4.pixel_stamps = np.clip(w4.pixel_stamps, lo + 1e-3, hi - 1e-3) + epoch_ns * 1e-9
    w4.t0 = t0_ns * 1e-9
    with ha.Problem(w4, lib=lib) as p:
        assert np.isfinite(p.cost())
    for what in ("hole", "shifted"):
        bad = w.control_points.copy()
        if what == "hole":  # element 6 pruned: every later row moves up, a fresh one is appended at the end
            bad[6:-1] = w.control_points[7:]
            bad[-1, 7] = bad[-2, 7] + w.dt
        else:
            bad[5, 7] += 1e-3 * w.dt
        w3 = synthetic.small_visual(order=4, n_cp=14, n_landmarks=20, obs_pairs=2)
        w3.control_points = bad
        try:
            ha.Problem(w3, lib=lib).close()
        except ha.HsError as e:
            assert "uniform" in str(e), e
        else:
            raise AssertionError(f"a control-point table with a {what} was accepted")
