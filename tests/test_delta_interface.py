"""Delta interface (include/hyperslam_hip.h: hs_append_* / hs_retire_* / hs_stage): tables kept incrementally between solves, the way
the reference keeps ceres::Problem (/root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:189-274 add(...), :347-382 addLandmark /
updateLandmarks; optimize() only solves, :276-280).

A window reached through appends and retirements must be THE SAME PROBLEM as the window handed over whole: same residual counts and
layouts, same cost, same reduced system, same solve. CPU: the oracle's restatement of the interface. GPU: the HIP library against the
oracle, and against itself (staged == unstaged bit for bit; value-only control-point updates; state that persists across solves)."""
import copy

import numpy as np
import pytest

import hyperslam_amd as ha
from hyperslam_amd import HS_BEARING, HS_INERTIAL, HS_PIXEL, HS_PRIOR, synthetic
from util import rel


def _window(order=4, seed=21, n_cp=18, n_landmarks=36, n_inertial=90, with_priors=12):
    w = synthetic.small_inertial(order=order, n_cp=n_cp, n_landmarks=n_landmarks, obs_pairs=3, n_inertial=n_inertial, seed=seed)
    p = synthetic.small_visual(order=order, n_cp=n_cp, n_landmarks=4, obs_pairs=2, seed=seed + 1, with_priors=with_priors)
    w.sensor_T_bs, w.prior_stamps, w.prior_poses, w.prior_sensor = p.sensor_T_bs, p.prior_stamps, p.prior_poses, p.prior_sensor
    return w


def _delta_build(w, lib, rng, stage_between=True):
    """Reaches window `w` through the delta interface: a start window holding part of w plus rows that will be retired, then appends and
    retirements in several steps. Returns (problem, new index of every landmark of w)."""
    n_lm = w.landmarks.shape[0]
    stale_lm = 5  # landmarks (with observations) that leave again
    first_half = np.arange(n_lm) < n_lm // 2
    s = copy.copy(w)
    # start: stale landmarks FIRST (so that every later index moves when they retire), then the first half of w's landmarks
    stale_xyz = w.landmarks[:stale_lm] + 0.3
    s.landmarks = np.concatenate([stale_xyz, w.landmarks[first_half]])
    pm = first_half[w.pixel_landmark]
    stale_obs = np.flatnonzero(w.pixel_landmark < stale_lm)
    s.pixel_stamps = np.concatenate([w.pixel_stamps[stale_obs], w.pixel_stamps[pm]])
    s.pixels = np.concatenate([w.pixels[stale_obs] + 1.0, w.pixels[pm]])
    s.pixel_landmark = np.concatenate([w.pixel_landmark[stale_obs], w.pixel_landmark[pm] + stale_lm]).astype(np.int32)
    s.pixel_camera = np.concatenate([w.pixel_camera[stale_obs], w.pixel_camera[pm]]).astype(np.int32)
    lo = w.valid_range()[0]
    old = lo + 1e-4 * (1 + np.arange(3))  # inertial / prior rows older than everything in w: retired by stamp
    cut = float(min(w.inertial_stamps.min(), w.prior_stamps.min(), old.max() + 1.0))
    assert old.max() < cut
    half_i = len(w.inertial_stamps) // 2
    s.inertial_stamps = np.concatenate([old, w.inertial_stamps[:half_i]])
    s.inertial_measurements = np.concatenate([w.inertial_measurements[:3] * 1.1, w.inertial_measurements[:half_i]])
    s.prior_stamps = np.concatenate([old[:2], w.prior_stamps[:4]])
    s.prior_poses = np.concatenate([w.prior_poses[:2], w.prior_poses[:4]])
    s.prior_sensor = np.zeros(len(s.prior_stamps), np.int32)
    P = ha.Problem(s, lib=lib)
    if stage_between:
        P.stage()
    # step 1: the second half of the landmarks arrives, with its observations, in two batches
    rest = np.flatnonzero(~first_half)
    index_of = np.full(n_lm, -1, np.int64)
    index_of[first_half] = stale_lm + np.arange(first_half.sum())
    for batch in np.array_split(rest, 2):
        first = P.append_landmarks(w.landmarks[batch])
        index_of[batch] = first + np.arange(len(batch))
        m = np.isin(w.pixel_landmark, batch)
        order = rng.permutation(np.flatnonzero(m))  # rows arrive in any order
        P.append_residuals(HS_PIXEL, w.pixel_stamps[order], w.pixels[order], landmark=index_of[w.pixel_landmark[order]], camera=w.pixel_camera[order])
        if stage_between:
            P.stage()
    # step 2: more inertial / prior rows
    P.append_residuals(HS_INERTIAL, w.inertial_stamps[half_i:], w.inertial_measurements[half_i:])
    P.append_residuals(HS_PRIOR, w.prior_stamps[4:], w.prior_poses[4:], sensor=w.prior_sensor[4:])
    if stage_between:
        P.stage()
    # step 3: the stale rows leave
    remap = P.retire_landmarks(np.arange(stale_lm))
    assert (remap[:stale_lm] == -1).all() and (remap[stale_lm:] == np.arange(len(remap) - stale_lm)).all()
    index_of = remap[index_of]
    P.retire_residuals_before(HS_INERTIAL, cut)
    P.retire_residuals_before(HS_PRIOR, cut)
    if stage_between:
        P.stage()
    P.window = w
    return P, index_of


def _same_problem(A, B, w, index_of, tol):
    """A: window w handed over whole; B: reached through deltas (landmark t of w = row index_of[t] of B)."""
    for t in (HS_PIXEL, HS_BEARING, HS_PRIOR, HS_INERTIAL):
        assert A.num_residuals(t) == B.num_residuals(t)
    assert A.dim_pose() == B.dim_pose()
    assert abs(A.cost() - B.cost()) <= tol * A.cost()
    Sa, ga = A.reduced_system(1e4)
    Sb, gb = B.reduced_system(1e4)
    assert rel(Sb, Sa) < tol and rel(gb, ga) < tol
    sa, sb = A.solve(4), B.solve(4)
    assert sa["num_iterations"] == sb["num_iterations"] and sa["num_successful_steps"] == sb["num_successful_steps"]
    assert abs(sa["final_cost"] - sb["final_cost"]) <= 1e3 * tol * sa["final_cost"]
    assert np.abs(A.control_points() - B.control_points()).max() < 1e4 * tol
    assert np.abs(A.landmarks() - B.landmarks()[index_of]).max() < 1e4 * tol


@pytest.mark.parametrize("order", [4, 6])
def test_oracle_delta_equals_whole_tables(oracle, order):
    w = _window(order=order)
    with ha.Problem(w, lib=oracle) as A:
        B, index_of = _delta_build(w, oracle, np.random.default_rng(5))
        with B:
            # residual layouts: the same block ids up to the landmark renumbering
            n = A.num_residuals(HS_INERTIAL)
            la, lb = A.residual_layout(HS_INERTIAL, n - 1), B.residual_layout(HS_INERTIAL, n - 1)
            assert (la["block_ids"] == lb["block_ids"]).all() and (la["sizes"] == lb["sizes"]).all()
            _same_problem(A, B, w, index_of, 1e-11)


def test_oracle_retire_reports_the_new_rows(oracle):
    w = _window()
    with ha.Problem(w, lib=oracle) as P:
        n = w.landmarks.shape[0]
        before = P.num_residuals(HS_PIXEL)
        gone = np.array([1, 4, n - 1], np.int32)
        remap = P.retire_landmarks(gone)
        keep = np.setdiff1d(np.arange(n), gone)
        assert (remap[gone] == -1).all() and (remap[keep] == np.arange(len(keep))).all()
        assert P.num_residuals(HS_PIXEL) == before - int(np.isin(w.pixel_landmark, gone).sum())
        with pytest.raises(ha.HsError):
            P.retire_landmarks(np.array([len(keep)], np.int32))  # outside the table


# ---- GPU -----------------------------------------------------------------------------------------------------------------------------


@pytest.mark.gpu
@pytest.mark.parametrize("order,stage_between", [(4, True), (4, False), (6, True)])
def test_hip_delta_equals_whole_tables(hip, oracle, order, stage_between):
    """The HIP library through the delta interface == the HIP library given the whole tables == the oracle given the whole tables."""
    w = _window(order=order)
    with ha.Problem(w) as A, ha.Problem(w, lib=oracle) as O:
        B, index_of = _delta_build(w, None, np.random.default_rng(5), stage_between=stage_between)
        with B:
            Sb, gb = B.reduced_system(1e4)
            So, go = O.reduced_system(1e4)
            assert rel(Sb, So) < 1e-9 and rel(gb, go) < 1e-9  # the parity bar of tests/test_gpu_parity.py
            _same_problem(A, B, w, index_of, 1e-10)
            so = O.solve(4)
            sb = B.solve(0)
            assert abs(sb["final_cost"] - so["final_cost"]) <= 1e-6 * so["final_cost"]


@pytest.mark.gpu
def test_hip_staged_solve_is_the_unstaged_solve(hip):
    """hs_stage only moves the sort + upload in front of the solve: bit-identical results."""
    w = _window()
    with ha.Problem(w) as A, ha.Problem(w) as B:
        B.stage()
        B.stage()  # (nothing changed: no-op)
        sa, sb = A.solve(5), B.solve(5)
        assert sa["final_cost"] == sb["final_cost"] and sa["num_iterations"] == sb["num_iterations"]
        assert np.array_equal(A.control_points(), B.control_points()) and np.array_equal(A.landmarks(), B.landmarks())


@pytest.mark.gpu
def test_hip_control_points_resent_with_the_same_knots(hip, oracle):
    """optimize() of a sliding window re-sends the control points before every solve: with the knots of the resident table that is a
    value-only update (one small copy, no sort) — and it must behave exactly like the whole-table path."""
    w = _window()
    with ha.Problem(w) as A, ha.Problem(w, lib=oracle) as O:
        first = A.solve(5)
        cp_after = A.control_points()
        assert np.abs(cp_after - w.control_points).max() > 1e-6  # the solve moved them
        # same values as the resident table AFTER a solve: the device is ahead of the host copy, so this must reset the control points
        A.set_control_points(w.control_points, w.cp_constant)
        again = A.solve(0)
        # landmarks / biases stayed at the solved point, control points went back: cost between the two
        assert again["final_cost"] > first["final_cost"]
        # the oracle does the same with the same calls
        O.solve(5)
        O.set_control_points(w.control_points, w.cp_constant)
        ref = O.solve(0)
        assert abs(again["final_cost"] - ref["final_cost"]) <= 1e-6 * ref["final_cost"]
        # new values, frozen prefix changed: still no structural upload, results follow the oracle
        frozen = np.zeros(w.n_cp, np.uint8)
        frozen[:5] = 1
        cp2 = w.control_points.copy()
        cp2[6:, 4:7] += 1e-3
        A.set_control_points(cp2, frozen)
        O.set_control_points(cp2, frozen)
        sa, so = A.solve(3), O.solve(3)
        assert abs(sa["final_cost"] - so["final_cost"]) <= 1e-6 * so["final_cost"]
        assert np.abs(A.control_points() - O.control_points()).max() < 1e-6
        assert np.array_equal(A.control_points()[:5], cp2[:5])  # frozen rows untouched


@pytest.mark.gpu
def test_hip_state_persists_across_delta_calls(hip, oracle):
    """solve -> append rows -> solve: the second solve starts where the first one ended (the library pulls its host copies of the variables
    up to the device before it edits the tables), as the reference's in-place variables do."""
    w = _window()
    half = len(w.inertial_stamps) // 2
    s = copy.copy(w)
    s.inertial_stamps, s.inertial_measurements = w.inertial_stamps[:half], w.inertial_measurements[:half]
    with ha.Problem(s) as A, ha.Problem(s, lib=oracle) as O:
        for P in (A, O):
            P.solve(3)
            P.append_residuals(HS_INERTIAL, w.inertial_stamps[half:], w.inertial_measurements[half:])
            new_lm = P.append_landmarks(np.array([[0.5, 0.25, 2.0]]))  # an unobserved landmark rides along
            assert new_lm == w.landmarks.shape[0]
            P.stage()
        sa, so = A.solve(3), O.solve(3)
        assert abs(sa["initial_cost"] - so["initial_cost"]) <= 1e-6 * so["initial_cost"]
        assert abs(sa["final_cost"] - so["final_cost"]) <= 1e-6 * so["final_cost"]
        assert A.landmarks().shape == (w.landmarks.shape[0] + 1, 3) and np.array_equal(A.landmarks()[-1], [0.5, 0.25, 2.0])
        assert np.abs(A.landmarks() - O.landmarks()).max() < 1e-6
        assert np.abs(A.control_points() - O.control_points()).max() < 1e-6
