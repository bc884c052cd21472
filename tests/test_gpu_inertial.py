"""GPU parity of the inertial factor (inertial.cpp:13-205) and of the bordered (bias splines + gravity) solve."""
import numpy as np
import pytest

import hyperslam_amd as ha
from hyperslam_amd import synthetic
from util import check_against_golden, golden_cases, golden_window, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("idx", [i for i, c in enumerate(golden_cases()) if c["type"] == "inertial"])
def test_hip_inertial_matches_golden(idx, hip):
    case = golden_cases()[idx]
    with ha.Problem(golden_window(case), lib=hip) as p:
        check_against_golden(p, case, 1e-9)


@pytest.mark.parametrize("order,identity", [(4, False), (6, False), (4, True), (5, False)])
@pytest.mark.parametrize("robustify", [False, True])
def test_inertial_linearization_vs_oracle(order, identity, robustify, hip, oracle):
    w = synthetic.small_inertial(order=order, n_cp=18, identity=identity)
    with ha.Problem(w, lib=hip) as g, ha.Problem(w, lib=oracle) as c:
        a, b = g.linearize(ha.HS_INERTIAL, robustify), c.linearize(ha.HS_INERTIAL, robustify)
        assert np.array_equal(a["first_cp"], b["first_cp"]) and np.array_equal(a["first_bias"], b["first_bias"])
        for k in ("r", "J_state", "J_bias_g", "J_bias_a", "J_gravity", "cost"):
            assert rel(a[k], b[k]) < 1e-9, (k, rel(a[k], b[k]))
        L = g.residual_layout(ha.HS_INERTIAL, 3)
        Lc = c.residual_layout(ha.HS_INERTIAL, 3)
        for k in L:
            assert np.array_equal(L[k], Lc[k]), k


@pytest.mark.parametrize("order,identity", [(4, False), (6, True), (5, False)])
def test_bordered_reduced_system(order, identity, hip, oracle):
    w = synthetic.small_inertial(order=order, n_cp=18, identity=identity)
    with ha.Problem(w, lib=hip) as g, ha.Problem(w, lib=oracle) as c:
        assert g.dim_pose() == c.dim_pose() == 6 * 18 + 6 * len(w.imu["bias_g"]) + 2
        Sg, gg = g.reduced_system(1e4)
        Sc, gc = c.reduced_system(1e4)
        assert rel(Sg, Sc) < 1e-9 and rel(gg, gc) < 1e-9, (rel(Sg, Sc), rel(gg, gc))


@pytest.mark.parametrize("order,identity,grav_const", [(4, False, False), (6, True, False), (4, True, True), (5, False, False)])
def test_bordered_solve_trajectory(order, identity, grav_const, hip, oracle):
    w = synthetic.small_inertial(order=order, n_cp=18, identity=identity)
    w.gravity_constant = grav_const
    w.cp_constant = np.r_[np.ones(order, np.uint8), np.zeros(18 - order, np.uint8)]
    with ha.Problem(w, lib=hip) as g, ha.Problem(w, lib=oracle) as c:
        sg, sc = g.solve(5), c.solve(5)
        assert sg["num_iterations"] == sc["num_iterations"] and sg["num_successful_steps"] == sc["num_successful_steps"]
        for ig, ic in zip(sg["iterations"], sc["iterations"]):
            assert ig["step_is_successful"] == ic["step_is_successful"]
            assert abs(ig["cost"] - ic["cost"]) <= 1e-6 * abs(ic["cost"]) + 1e-8 * sc["initial_cost"], (ig["iteration"], ig["cost"], ic["cost"])
        assert rel(g.control_points(), c.control_points()) < 1e-6
        assert rel(g.landmarks(), c.landmarks()) < 1e-6
        assert rel(g.gravity(), c.gravity()) < 1e-6
        bg, ba = g.bias()
        cg, ca = c.bias()
        assert rel(bg, cg) < 1e-6 and rel(ba, ca) < 1e-6
        if grav_const:
            assert np.array_equal(g.gravity(), w.gravity)


@pytest.mark.parametrize("order,n_cp", [(4, 72), (6, 80), (5, 76)])
def test_two_ended_bordered_solve(order, n_cp, hip, oracle, monkeypatch):
    """Windows long enough (n_cp >= 4 band widths) for the bordered system to be factored from both ends (k_border_forward2, the y view over
    both ends, border outputs of the two-ended backward sweep): the 5-iteration trajectory against the oracle and, bias points and gravity
    included, against the one-ended path of the same library (A/B switch 536870912)."""
    w = synthetic.small_inertial(order=order, n_cp=n_cp, n_landmarks=300, obs_pairs=3, n_inertial=600, seed=33)
    w.cp_constant = np.r_[np.ones(order, np.uint8), np.zeros(n_cp - order, np.uint8)]

    def run(lib):
        with ha.Problem(w, lib=lib) as p:
            bw = p.lib.band_blocks(p.h)
            s = p.solve(5)
            bg, ba = p.bias()
            return bw, s, p.control_points(), p.landmarks(), p.gravity(), bg, ba

    bw, sg, cpg, lmg, gg, bgg, bag = run(hip)
    assert n_cp >= 4 * bw and bw <= 16, (bw, "the window must take the two-ended look-ahead path")
    _, sc, cpc, lmc, gc, bgc, bac = run(oracle)
    assert sg["num_iterations"] == sc["num_iterations"] and sg["num_successful_steps"] == sc["num_successful_steps"]
    for ig, ic in zip(sg["iterations"], sc["iterations"]):
        assert ig["step_is_successful"] == ic["step_is_successful"]
        assert abs(ig["cost"] - ic["cost"]) <= 1e-6 * abs(ic["cost"]) + 1e-8 * sc["initial_cost"]
    assert rel(cpg, cpc) < 1e-6 and rel(lmg, lmc) < 1e-6 and rel(gg, gc) < 1e-6 and rel(bgg, bgc) < 1e-6 and rel(bag, bac) < 1e-6
    monkeypatch.setenv("HS_DEBUG_FLAGS", str(536870912))  # the same window through the one-ended bordered path
    _, s1, cp1, lm1, g1, bg1, ba1 = run(hip)
    assert [it["step_is_successful"] for it in s1["iterations"]] == [it["step_is_successful"] for it in sg["iterations"]]
    assert rel(cpg, cp1) < 1e-8 and rel(lmg, lm1) < 1e-8 and rel(gg, g1) < 1e-8 and rel(bgg, bg1) < 1e-7 and rel(bag, ba1) < 1e-7
    # The forward sweep of the border columns runs NEXT TO the factorisation and follows its progress words (round 5: k_border_forward2 on
    # the side stream, MfmaJob::progress); A/B switch 128 runs it behind the factorisation. Same arithmetic on the same values: every run of
    # either arrangement must give the same bits — a sweep that read a factor row before it was complete would not.
    monkeypatch.setenv("HS_DEBUG_FLAGS", "128")
    _, s2, cp2, lm2, g2, bg2, ba2 = run(hip)
    monkeypatch.setenv("HS_DEBUG_FLAGS", "0")
    for _ in range(4):
        _, s3, cp3, lm3, g3, bg3, ba3 = run(hip)
        assert np.array_equal(cp3, cp2) and np.array_equal(lm3, lm2) and np.array_equal(g3, g2) and np.array_equal(bg3, bg2) and np.array_equal(ba3, ba2)
        assert [it["cost"] for it in s3["iterations"]] == [it["cost"] for it in s2["iterations"]]


@pytest.mark.parametrize("gravity_constant", [True, False])
def test_imu_tables_without_inertial_residuals(gravity_constant, hip, oracle):
    """An IMU whose samples have all left the window (or have not arrived yet): the bias splines and gravity are unknowns without
    residuals — structurally zero columns of the border. The solve must be the visual-only solve, biases and gravity must come back
    untouched (the reference-side plugin relies on it: it hands the empty inertial table over, include/hyper/optimizers/hip/optimizer.hpp)."""
    import copy
    w = synthetic.small_inertial(order=4, n_cp=18, n_landmarks=40, seed=23)
    w.gravity_constant = gravity_constant
    empty = copy.copy(w)
    empty.inertial_stamps, empty.inertial_measurements = w.inertial_stamps[:0], w.inertial_measurements[:0]
    visual = copy.copy(empty)
    visual.imu = None
    with ha.Problem(empty, lib=hip) as g, ha.Problem(visual, lib=hip) as v, ha.Problem(visual, lib=oracle) as c:
        sg, sv, sc = g.solve(5), v.solve(5), c.solve(5)
        assert sg["num_iterations"] == sv["num_iterations"] == sc["num_iterations"]
        assert [it["step_is_successful"] for it in sg["iterations"]] == [it["step_is_successful"] for it in sc["iterations"]]
        assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-8 * sc["final_cost"] and abs(sg["final_cost"] - sv["final_cost"]) <= 1e-10 * sv["final_cost"]
        assert rel(g.control_points(), c.control_points()) < 1e-6 and rel(g.landmarks(), c.landmarks()) < 1e-6
        bg, ba = g.bias()
        assert np.array_equal(bg, np.asarray(w.imu["bias_g"]).reshape(-1, 4)) and np.array_equal(ba, np.asarray(w.imu["bias_a"]).reshape(-1, 4))
        assert np.array_equal(g.gravity(), np.asarray(w.gravity, dtype=np.float64))


def test_hip_matches_literal_inertial_golden(hip):
    """The library's DEFAULT inertial Jacobian (as written upstream, inertial.cpp:131-198) at non-identity IMU parameters against the
    100-digit transcription of tests/golden/make_inertial_literal_golden.py: 32 cases, same bar as the oracle's CPU test."""
    from util import check_against_literal_golden, golden_window, literal_inertial_cases
    for case in literal_inertial_cases():
        with ha.Problem(golden_window(case), lib=hip) as p:
            check_against_literal_golden(p, case, 1e-9)


def test_long_inertial_windows(hip, oracle):
    """An IMU window of 20 s at one bias control point per second: 146 border unknowns — the dense solve of the border system keeps its Schur
    complement in LDS (k_border_solve) up to 22 bias control points; beyond, the window is refused when the tables are prepared, with a message
    (before round 5 the launch failed inside hs_solve with 'invalid argument'). 17 s (21 bias control points, 128 border unknowns) is solved: the
    LDS-resident variant behind the register one, against the oracle. Such windows also carry thousands of inertial residuals next to the
    visual ones — the shape on which the cost-partial tables of the inertial kernels were found too short (tools/fuzz_parity.py large)."""
    w = synthetic.small_inertial(order=4, n_cp=170, n_landmarks=600, obs_pairs=3, n_inertial=3000, seed=51)
    assert len(w.imu["bias_g"]) == 21
    with ha.Problem(w, lib=hip) as g, ha.Problem(w, lib=oracle) as c:
        assert abs(g.cost() - c.cost()) <= 1e-11 * c.cost()
        sg, sc = g.solve(3), c.solve(3)
        assert [i["step_is_successful"] for i in sg["iterations"]] == [i["step_is_successful"] for i in sc["iterations"]]
        assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
        bg, ba = g.bias()
        cg, ca = c.bias()
        assert rel(g.control_points(), c.control_points()) < 1e-6 and rel(bg, cg) < 1e-6 and rel(ba, ca) < 1e-6
    w = synthetic.small_inertial(order=4, n_cp=206, n_landmarks=100, obs_pairs=3, n_inertial=500, seed=52)
    assert len(w.imu["bias_g"]) > 22
    with ha.Problem(w, lib=hip) as g:
        with pytest.raises(RuntimeError, match="too many border unknowns"):
            g.cost()
