"""Solver level (`ceres::Solve` behind CeresOptimizer::optimize, /root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:38-54,
276-280) against tests/golden/solve.json: an independent 100-digit restatement of Ceres' trust-region Levenberg-Marquardt iteration
(tests/golden/make_solve_golden.py: central-difference Jacobian of the mpmath residual functions, loss correction, Jacobi scaling, LM
diagonal, dense solve and landmark Schur complement, model cost change, step quality, accept / reject, radius update) on a small window with
every factor type, a frozen prefix, free bias splines and free gravity. The vectors share no code with the oracle or the HIP library; the
four recorded iterations contain three accepted steps and one rejected step.

Second window, tests/golden/solve_visual.json (`make_solve_golden.py visual`): order 6, twelve control points with a constant first segment,
six landmarks that start a metre from their true positions, pixel and bearing blocks only. Six iterations: accept, reject, reject, accept,
reject, reject — the sequence on which the HIP library's visual-only path (linearisation at the candidate point, records of the current
point kept across rejected steps, deferred commit) has to reproduce Ceres' bookkeeping."""
import os

import pytest

from hyperslam_amd import _lib
from util import check_solver_against_golden, solve_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_golden_window_shape():
    d, w = solve_golden()
    assert w.order == 4 and w.n_cp == 9 and list(d["cp_constant"]) == [1, 1, 0, 0, 0, 0, 0, 0, 0]
    assert len(w.pixel_stamps) == 12 and len(w.bearing_stamps) == 12 and len(w.prior_stamps) == 5 and len(w.inertial_stamps) == 8
    assert [r["step_is_successful"] for r in d["iterations"]] == [1, 1, 0, 1]  # accepted and rejected steps are both pinned
    assert all(0.0 < r["model_cost_change"] for r in d["iterations"])


def test_oracle_solver_matches_golden():
    """The CPU oracle: evaluated quantities to 1e-9, everything behind the linear solve to 1e-6 (the north-star tolerance; measured 8e-14 / 2e-8)."""
    lib = _lib.Library(os.path.join(ROOT, "oracle", "liboracle.so"), "hso_")
    worst = check_solver_against_golden(lib, 1e-9, 1e-6)
    assert worst["forward"] <= 1e-11 and worst["state"] <= 1e-6, worst


@pytest.mark.gpu
def test_hip_solver_matches_golden():
    """The HIP library through the C ABI: reduced normal equations to 1e-9, LM trajectory and state to 1e-6 (BASELINE.json north_star)."""
    worst = check_solver_against_golden(_lib.load(), 1e-9, 1e-6)
    print("worst errors vs the 100-digit solver vectors:", worst)


def test_visual_golden_window_shape():
    d, w = solve_golden("solve_visual.json")
    assert w.order == 6 and w.n_cp == 12 and list(d["cp_constant"]) == [1] * 6 + [0] * 6 and w.imu is None
    assert len(w.pixel_stamps) == 24 and len(w.bearing_stamps) == 24 and len(w.prior_stamps) == 0 and len(w.inertial_stamps) == 0
    assert [r["step_is_successful"] for r in d["iterations"]] == [1, 0, 0, 1, 0, 0]
    assert all(0.0 < r["model_cost_change"] for r in d["iterations"])


def test_oracle_solver_matches_visual_golden():
    lib = _lib.Library(os.path.join(ROOT, "oracle", "liboracle.so"), "hso_")
    worst = check_solver_against_golden(lib, 1e-9, 1e-6, "solve_visual.json")
    assert worst["forward"] <= 1e-10 and worst["state"] <= 1e-6, worst


@pytest.mark.gpu
def test_hip_solver_matches_visual_golden():
    worst = check_solver_against_golden(_lib.load(), 1e-9, 1e-6, "solve_visual.json")
    print("worst errors vs the 100-digit solver vectors (visual-only window):", worst)
