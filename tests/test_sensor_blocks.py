"""Sensor-parameter-block Jacobians (extrinsics for all four factors; camera intrinsics + radtan distortion for pixel; i_g, i_a, S_g, X_a
for inertial): bearing.cpp:74, pixel.cpp:91-135,141, manifold.cpp:57, inertial.cpp:155-194.

The reference's own tests probe exactly these (all parameter blocks non-constant): `Probe`, tests/include/tests/optimizers/evaluators/
evaluator.hpp:22-65 — ceres::GradientChecker with relative step 1e-6, analytic LOCAL Jacobian vs numeric at 1e-5 (relative check, or the
check on the normalised matrices). `probe()` below restates that protocol on top of the Ceres-style entry point
hs_cost_function_evaluate + hs_manifold_plus / hs_manifold_plus_jacobian of whichever library it is given: the oracle on the CPU, the HIP
library on the GPU."""
import numpy as np
import pytest

import hyperslam_amd as ha
from hyperslam_amd import synthetic
from util import rel

TOL = 1e-5  # evaluator.hpp:23 kDefaultNumericTolerance


def block_kinds(ftype, k, kb):
    """Manifold of every parameter block in ExteroceptiveCost::update order, all blocks variable (the reference's tests)."""
    kinds = [ha.HS_MANIFOLD_CONTROL_POINT] * k + [ha.HS_MANIFOLD_SE3]
    if ftype in (ha.HS_PIXEL, ha.HS_BEARING):
        kinds += [ha.HS_MANIFOLD_EUCLIDEAN, ha.HS_MANIFOLD_EUCLIDEAN, ha.HS_MANIFOLD_EUCLIDEAN]
    elif ftype == ha.HS_INERTIAL:
        kinds += [ha.HS_MANIFOLD_EUCLIDEAN] * 4 + [ha.HS_MANIFOLD_BIAS_POINT] * (2 * kb) + [ha.HS_MANIFOLD_SPHERE3]
    return kinds


def probe(p, ftype, idx, blocks, tol=TOL):
    """ceres::GradientChecker::Probe through `p`'s entry points. Returns the worst relative error over all blocks."""
    k = p.window.order
    kb = int(p.window.imu["bias_order"]) if ftype == ha.HS_INERTIAL else 0
    kinds = block_kinds(ftype, k, kb)
    assert len(kinds) == len(blocks)
    r0, jac = p.cost_function_evaluate(ftype, idx, blocks, [True] * len(blocks))
    # entries far below the block row's overall scale (a control point whose weight vanishes, B_3 = u^3 / 6 at u -> 0) are compared on
    # that scale: the numeric derivative carries |r| eps / h of noise
    floor = 1e-3 * max(np.abs(J).max() for J in jac)
    worst = 0.0
    for b, (kind, x) in enumerate(zip(kinds, blocks)):
        P = p.manifold_plus_jacobian(kind, x[None, :])[0]  # ambient x tangent
        J_local = jac[b] @ P
        J_num = np.zeros_like(J_local)
        for c in range(P.shape[1]):
            h = 1e-6 * max(1.0, float(np.abs(x).max()))  # relative step (NumericDiffOptions, evaluator.hpp:28-33)
            d = np.zeros((1, P.shape[1]))
            res = []
            for sgn in (+1.0, -1.0):
                d[0, c] = sgn * h
                moved = list(blocks)
                moved[b] = p.manifold_plus(kind, x[None, :], d)[0]
                res.append(p.cost_function_evaluate(ftype, idx, moved)[0])
            J_num[:, c] = (res[0] - res[1]) / (2 * h)
        # relative check per entry (GradientChecker: |a - n| / max(|a|, |n|)), or the reference's normalised-matrix check
        scale = np.maximum(np.maximum(np.abs(J_local), np.abs(J_num)), floor)
        relative = np.abs(J_local - J_num) / scale
        na, nn = max(np.linalg.norm(J_local), floor), max(np.linalg.norm(J_num), floor)
        absolute = np.abs(J_local / na - J_num / nn).max()
        err = min(float(relative.max()), float(absolute))
        assert err < tol, (ftype, idx, b, kind, err, J_local, J_num)
        worst = max(worst, err)
    return worst


def probe_window(ftype, order):
    if ftype == ha.HS_INERTIAL:
        return synthetic.small_inertial(order=order, n_cp=18, identity=False)
    return synthetic.small_visual(order=order, n_cp=16, n_landmarks=20, obs_pairs=2, bearing=(ftype == ha.HS_BEARING), seed=12, with_priors=6)


def run_probe(lib, ftype, order, mode=None):
    w = probe_window(ftype, order)
    if mode == ha.HS_INERTIAL_AS_REFERENCE:  # the reference's own test point: Mock<IMU>::Create(), tests/include/tests/sensors/imu.hpp:20-26
        w.imu.update(i_g=[1, 1, 1, 0, 0, 0], i_a=[1, 1, 1, 0, 0, 0], S_g=np.zeros(9), X_a=np.zeros(9))
    with ha.Problem(w, lib=lib) as p:
        if mode is not None:
            p.set_inertial_jacobian(mode)
        worst = 0.0
        for idx in (0, 3, p.num_residuals(ftype) - 1):
            worst = max(worst, probe(p, ftype, idx, p.parameter_blocks(ftype, idx)))
    return worst


CASES = [(ha.HS_PIXEL, 4, None), (ha.HS_PIXEL, 6, None), (ha.HS_BEARING, 4, None), (ha.HS_BEARING, 6, None), (ha.HS_PRIOR, 4, None),
         (ha.HS_PRIOR, 6, None), (ha.HS_INERTIAL, 4, ha.HS_INERTIAL_EXACT), (ha.HS_INERTIAL, 6, ha.HS_INERTIAL_EXACT),
         (ha.HS_INERTIAL, 4, ha.HS_INERTIAL_AS_REFERENCE), (ha.HS_INERTIAL, 6, ha.HS_INERTIAL_AS_REFERENCE)]


@pytest.mark.parametrize("ftype,order,mode", CASES)
def test_probe_oracle(ftype, order, mode, oracle):
    """The reference's Gradients tests (tests/internal/tests/optimizers/evaluators/{pixel,bearing,manifold,inertial}.cpp) on the oracle."""
    assert run_probe(oracle, ftype, order, mode) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("ftype,order,mode", CASES)
def test_probe_hip(ftype, order, mode, hip):
    """The same protocol through the HIP entry points: every block non-null, analytic local Jacobian vs numeric at 1e-5."""
    assert run_probe(hip, ftype, order, mode) < TOL


def test_in_tree_inertial_jacobian_differs_from_exact_off_identity(oracle):
    """inertial.cpp:136,142,148,158 carry I_g where the prediction has I_a and drop the S_g / X_a terms: with I_g != I_a the default
    (as written upstream) Jacobian is not the derivative of the residual — both forms are kept, the default is upstream's."""
    w = synthetic.small_inertial(order=4, n_cp=18, identity=False)
    with ha.Problem(w, lib=oracle) as p:
        lit = p.linearize(ha.HS_INERTIAL, False, sensor_blocks=True)
        p.set_inertial_jacobian(ha.HS_INERTIAL_EXACT)
        exact = p.linearize(ha.HS_INERTIAL, False, sensor_blocks=True)
        assert rel(lit["r"], exact["r"]) == 0.0
        assert rel(lit["J_state"], exact["J_state"]) > 1e-3 and rel(lit["J_extrinsics"], exact["J_extrinsics"]) > 1e-3
        for key in ("J_gyro_intrinsics", "J_acc_intrinsics", "J_gyro_sensitivity", "J_bias_g", "J_bias_a"):
            assert np.array_equal(lit[key], exact[key]), key
        with pytest.raises(ha.HsError):
            p.set_inertial_jacobian(7)
    wi = synthetic.small_inertial(order=4, n_cp=18, identity=True)
    with ha.Problem(wi, lib=oracle) as p:
        lit = p.linearize(ha.HS_INERTIAL, False, sensor_blocks=True)
        p.set_inertial_jacobian(ha.HS_INERTIAL_EXACT)
        exact = p.linearize(ha.HS_INERTIAL, False, sensor_blocks=True)
        for key in lit:
            assert rel(lit[key], exact[key]) < 1e-14, key


SENSOR_KEYS = {ha.HS_PIXEL: ("J_extrinsics", "J_intrinsics", "J_distortion"), ha.HS_BEARING: ("J_extrinsics",), ha.HS_PRIOR: ("J_extrinsics",),
               ha.HS_INERTIAL: ("J_extrinsics", "J_gyro_intrinsics", "J_acc_intrinsics", "J_gyro_sensitivity", "J_acc_offsets")}


@pytest.mark.gpu
@pytest.mark.parametrize("ftype,order,mode", CASES)
@pytest.mark.parametrize("robustify", [False, True])
def test_sensor_blocks_hip_vs_oracle(ftype, order, mode, robustify, hip, oracle):
    """hs_linearize's optional sensor-block outputs, whole tables, HIP vs oracle at 1e-9 (both forms of the inertial Jacobian, off identity)."""
    w = probe_window(ftype, order)
    with ha.Problem(w, lib=hip) as g, ha.Problem(w, lib=oracle) as c:
        if mode is not None:
            g.set_inertial_jacobian(mode), c.set_inertial_jacobian(mode)
        a, b = g.linearize(ftype, robustify, sensor_blocks=True), c.linearize(ftype, robustify, sensor_blocks=True)
        for key in ("r", "J_state") + SENSOR_KEYS[ftype]:
            scale = np.abs(b["J_state"]).max() if key != "r" else 0.0
            assert np.abs(a[key] - b[key]).max() <= 1e-9 * max(np.abs(b[key]).max(), 1e-3 * scale), (key, rel(a[key], b[key]))
        # the ambient Jacobians of the Ceres-style entry point, every block, against the oracle's after projection on the tangent
        kb = int(w.imu["bias_order"]) if ftype == ha.HS_INERTIAL else 0
        kinds = block_kinds(ftype, w.order, kb)
        for idx in (1, g.num_residuals(ftype) - 2):
            blocks = g.parameter_blocks(ftype, idx)
            rg, Jg = g.cost_function_evaluate(ftype, idx, blocks, [True] * len(blocks))
            rc, Jc = c.cost_function_evaluate(ftype, idx, blocks, [True] * len(blocks))
            assert rel(rg, rc) < 1e-9
            scale = max(np.abs(J).max() for J in Jc)
            for bi, kind in enumerate(kinds):
                P = c.manifold_plus_jacobian(kind, blocks[bi][None, :])[0]
                assert np.abs(Jg[bi] @ P - Jc[bi] @ P).max() <= 1e-9 * scale, (bi, kind)


# ---- CostConfiguration::weights (exteroceptive.cpp:109-121,129-147): output = W * distance(..), J_w = W * J_m * J_e ------------------
def weight_matrix(ftype, seed=5):
    nr = {ha.HS_PIXEL: 2, ha.HS_BEARING: 1, ha.HS_PRIOR: 6, ha.HS_INERTIAL: 6}[ftype]
    rng = np.random.default_rng(seed + ftype)
    return np.eye(nr) * rng.uniform(0.5, 2.0, size=nr) + 0.2 * rng.normal(size=(nr, nr))


WEIGHT_CASES = [(ha.HS_PIXEL, 4), (ha.HS_BEARING, 6), (ha.HS_PRIOR, 4), (ha.HS_INERTIAL, 6)]


@pytest.mark.parametrize("ftype,order", WEIGHT_CASES)
def test_weights_oracle(ftype, order, oracle):
    """With W set the entry point returns W r and W J (checked against the unweighted call), and the gradient probe still holds."""
    w = probe_window(ftype, order)
    W = weight_matrix(ftype)
    with ha.Problem(w, lib=oracle) as p:
        if ftype == ha.HS_INERTIAL:
            p.set_inertial_jacobian(ha.HS_INERTIAL_EXACT)
        blocks = p.parameter_blocks(ftype, 2)
        r0, J0 = p.cost_function_evaluate(ftype, 2, blocks, [True] * len(blocks))
        p.set_weights(ftype, W)
        r1, J1 = p.cost_function_evaluate(ftype, 2, blocks, [True] * len(blocks))
        assert np.allclose(r1, W @ r0, rtol=1e-13, atol=1e-15)
        for a, b in zip(J0, J1):
            assert np.allclose(b, W @ a, rtol=1e-12, atol=1e-13 * max(1.0, np.abs(a).max()))
        assert probe(p, ftype, 2, blocks) < TOL
        p.set_weights(ftype, None)
        r2, _ = p.cost_function_evaluate(ftype, 2, blocks)
        assert np.array_equal(r2, r0)


@pytest.mark.gpu
@pytest.mark.parametrize("ftype,order", WEIGHT_CASES)
@pytest.mark.parametrize("robustify", [False, True])
def test_weights_hip_vs_oracle(ftype, order, robustify, hip, oracle):
    """hs_set_weights through the HIP evaluation entry points (whole tables and the Ceres-style single block) against the oracle; the
    solver refuses to run with weights (the reference never passes any: optimizer.cpp:191,214,236,255)."""
    w = probe_window(ftype, order)
    W = weight_matrix(ftype)
    with ha.Problem(w, lib=hip) as g, ha.Problem(w, lib=oracle) as c:
        if ftype == ha.HS_INERTIAL:  # (the gradient probe below needs the derivative of the residual: I_g != I_a in this window)
            g.set_inertial_jacobian(ha.HS_INERTIAL_EXACT), c.set_inertial_jacobian(ha.HS_INERTIAL_EXACT)
        g.set_weights(ftype, W), c.set_weights(ftype, W)
        a, b = g.linearize(ftype, robustify, sensor_blocks=True), c.linearize(ftype, robustify, sensor_blocks=True)
        keys = ("r", "J_state", "cost") + SENSOR_KEYS[ftype] + (("J_landmark",) if ftype in (ha.HS_PIXEL, ha.HS_BEARING) else ()) + \
            (("J_bias_g", "J_bias_a", "J_gravity") if ftype == ha.HS_INERTIAL else ())
        scale = np.abs(b["J_state"]).max()
        for key in keys:
            assert np.abs(a[key] - b[key]).max() <= 1e-9 * max(np.abs(b[key]).max(), 1e-3 * scale if key.startswith("J") else 0.0), key
        if robustify and ftype in (ha.HS_PIXEL, ha.HS_BEARING):  # the corrector acts on the WEIGHTED residual
            u = g.linearize(ftype, False)
            assert rel(a["r"], u["r"]) > 1e-6
        blocks = g.parameter_blocks(ftype, 1)
        rg, Jg = g.cost_function_evaluate(ftype, 1, blocks, [True] * len(blocks))
        rc, Jc = c.cost_function_evaluate(ftype, 1, blocks, [True] * len(blocks))
        assert rel(rg, rc) < 1e-9
        kb = int(w.imu["bias_order"]) if ftype == ha.HS_INERTIAL else 0
        sc = max(np.abs(J).max() for J in Jc)
        for bi, kind in enumerate(block_kinds(ftype, w.order, kb)):
            P = c.manifold_plus_jacobian(kind, blocks[bi][None, :])[0]
            assert np.abs(Jg[bi] @ P - Jc[bi] @ P).max() <= 1e-9 * sc, bi
        assert probe(g, ftype, 1, blocks) < TOL
        with pytest.raises(ha.HsError, match="weight"):
            g.solve(1)
        g.set_weights(ftype, None)
        assert g.solve(1)["num_iterations"] == 1
