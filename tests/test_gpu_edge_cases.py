"""Edge cases of the hot path, HIP vs oracle: the shapes the reference's window maintenance produces (frozen / unobserved
control points, constant or unobserved landmarks, clamped boundary stamps, minimal windows, rotation- or translation-constant
splines, constant biases) plus degenerate inputs (empty tables)."""
import copy
import os

import numpy as np
import pytest

import hyperslam_amd as ha
from hyperslam_amd import synthetic
from util import rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compare(w, hip, oracle, iters=5, tol=1e-6, check_lm=True, sys_tol=1e-9):
    with ha.Problem(w, lib=hip) as g, ha.Problem(w, lib=oracle) as c:
        cg, cc = g.cost(), c.cost()
        assert abs(cg - cc) <= 1e-11 * max(cc, 1e-300)
        Sg, gg = g.reduced_system(1e4)
        Sc, gc = c.reduced_system(1e4)
        assert rel(Sg, Sc) < sys_tol and rel(gg, gc) < sys_tol, (rel(Sg, Sc), rel(gg, gc))
        sg, sc = g.solve(iters), c.solve(iters)
        assert sg["num_iterations"] == sc["num_iterations"] and sg["termination"] == sc["termination"]
        assert abs(sg["final_cost"] - sc["final_cost"]) <= tol * abs(sc["final_cost"]) + 1e-8 * sc["initial_cost"]
        assert rel(g.control_points(), c.control_points()) < tol
        if check_lm and len(w.landmarks):
            assert rel(g.landmarks(), c.landmarks()) < tol
        return sg


def test_minimal_window(hip, oracle):
    """n_cp == k: a single segment (the bootstrap state of abstract.cpp:76-96 right after the first frames)."""
    for k in (4, 6):
        w = synthetic.small_visual(order=k, n_cp=k, n_landmarks=12, obs_pairs=2, seed=31)
        w.cp_constant = np.r_[np.ones(2, np.uint8), np.zeros(k - 2, np.uint8)]
        compare(w, hip, oracle)


def test_frozen_and_unobserved_control_points(hip, oracle):
    """Old control points frozen (optimizer.cpp:323-328); the newest ones carry no residual at all (held poses after an extension,
    abstract.cpp:127-137): their diagonal blocks are pure LM damping."""
    w = synthetic.small_visual(order=4, n_cp=20, n_landmarks=50, obs_pairs=3, seed=32)
    lo, hi = w.valid_range()
    cut = lo + 0.7 * (hi - lo)  # no observation after `cut`
    keep = w.pixel_stamps < cut
    for name in ("pixel_stamps", "pixels", "pixel_landmark", "pixel_camera"):
        setattr(w, name, getattr(w, name)[keep])
    w.cp_constant = np.r_[np.ones(5, np.uint8), np.zeros(15, np.uint8)]
    compare(w, hip, oracle)


def test_constant_and_unobserved_landmarks(hip, oracle):
    w = synthetic.small_visual(order=4, n_cp=16, n_landmarks=40, obs_pairs=3, seed=33)
    lc = np.zeros(40, np.uint8)
    lc[::5] = 1  # every fifth landmark constant
    w.landmark_constant = lc
    drop = np.isin(w.pixel_landmark, [3, 17])  # two landmarks lose all their observations (still in the table)
    for name in ("pixel_stamps", "pixels", "pixel_landmark", "pixel_camera"):
        setattr(w, name, getattr(w, name)[~drop])
    before = w.landmarks.copy()
    with ha.Problem(w, lib=hip) as g:
        g.solve(5)
        after = g.landmarks()
    assert np.array_equal(after[lc == 1], before[lc == 1]) and np.array_equal(after[[3, 17]], before[[3, 17]])
    compare(w, hip, oracle)


def test_boundary_stamps(hip, oracle):
    """Residual stamps exactly on the first knot of the valid range and one ulp below its end."""
    w = synthetic.small_visual(order=4, n_cp=14, n_landmarks=30, obs_pairs=3, seed=34)
    lo, hi = w.valid_range()
    st = w.pixel_stamps.copy()
    st[:10] = lo
    st[10:20] = np.nextafter(hi, lo)
    w.pixel_stamps = st
    compare(w, hip, oracle, tol=1e-5)


@pytest.mark.parametrize("which", ["rotation", "translation"])
def test_constancy_flags(which, hip, oracle):
    """`rotation_constant` / `translation_constant` of the backend YAML: half of every control point frozen."""
    w = synthetic.small_visual(order=4, n_cp=14, n_landmarks=40, obs_pairs=3, seed=35, with_priors=10)
    setattr(w, which + "_constant", True)
    cp0 = w.control_points.copy()
    with ha.Problem(w, lib=hip) as g:
        g.solve(5)
        cp1 = g.control_points()
    frozen = slice(0, 4) if which == "rotation" else slice(4, 7)
    assert np.array_equal(cp1[:, frozen], cp0[:, frozen])
    compare(w, hip, oracle)


def test_constant_biases_and_gravity(hip, oracle):
    w = synthetic.small_inertial(order=4, n_cp=16, n_landmarks=30, obs_pairs=3, n_inertial=80, seed=36)
    w.imu["bias_constant"] = True
    w.gravity_constant = True
    bg0 = w.imu["bias_g"].copy()
    with ha.Problem(w, lib=hip) as g:
        g.solve(5)
        bg1, _ = g.bias()
        assert np.array_equal(bg1[:, :3], bg0[:, :3]) and np.array_equal(g.gravity(), w.gravity)
    compare(w, hip, oracle)


def test_inertial_only_window(hip, oracle):
    """No visual residual at all: the reduced system is the pose block + border, no landmark elimination."""
    w = synthetic.small_inertial(order=4, n_cp=14, n_landmarks=10, obs_pairs=2, n_inertial=120, seed=37)
    for name in ("pixel_stamps", "pixels", "pixel_landmark", "pixel_camera"):
        setattr(w, name, getattr(w, name)[:0])
    w.landmarks = w.landmarks[:0]
    w.cp_constant = np.r_[np.ones(3, np.uint8), np.zeros(11, np.uint8)]
    compare(w, hip, oracle, check_lm=False)


def test_empty_problem(hip):
    """A spline without any residual: cost 0, the solve terminates at once, nothing moves."""
    w = synthetic.small_visual(order=4, n_cp=10, n_landmarks=5, obs_pairs=2, seed=38)
    for name in ("pixel_stamps", "pixels", "pixel_landmark", "pixel_camera"):
        setattr(w, name, getattr(w, name)[:0])
    w.landmarks = w.landmarks[:0]
    with ha.Problem(w, lib=hip) as g:
        assert g.cost() == 0.0
        s = g.solve(5)
        assert s["final_cost"] == 0.0 and np.array_equal(g.control_points(), w.control_points)


def test_many_observations_of_one_landmark(hip, oracle):
    """One landmark seen in every frame (a long track next to short ones): exercises uneven group / segment work lists."""
    w = synthetic.small_visual(order=4, n_cp=18, n_landmarks=30, obs_pairs=3, seed=39)
    lo, hi = w.valid_range()
    n_extra = 120
    st = np.linspace(lo, hi - 1e-6, n_extra)
    rng = np.random.default_rng(5)
    w2 = copy.deepcopy(w)
    w2.pixel_stamps = np.r_[w.pixel_stamps, st]
    w2.pixels = np.r_[w.pixels, np.stack([rng.uniform(100, 600, n_extra), rng.uniform(100, 400, n_extra)], -1)]
    w2.pixel_landmark = np.r_[w.pixel_landmark, np.zeros(n_extra, np.int32)]
    w2.pixel_camera = np.r_[w.pixel_camera, np.zeros(n_extra, np.int32)]
    compare(w2, hip, oracle, tol=1e-5)


def test_long_tracks_with_imu(hip, oracle):
    """Window-wide feature tracks (wide-band factorisation) together with the bias / gravity border, as in the sliding-window replay."""
    w = synthetic.small_visual(order=4, n_cp=30, n_landmarks=60, obs_pairs=6, seed=40, span=2.8)
    synthetic.add_imu(w, synthetic.SplitMix64(77), 150, identity=True, gravity_constant=False)
    w.cp_constant = np.r_[np.ones(3, np.uint8), np.zeros(27, np.uint8)]
    with ha.Problem(w, lib=hip) as g:
        g.cost()
        assert g.lib.band_blocks(g.h) > 22  # really on the wide path
    compare(w, hip, oracle, tol=1e-5)


def test_band_width_limits(hip, oracle):
    """Tracks touching up to 42 control points are supported (6 bw <= 256); wider ones are rejected with a message, never silently."""
    w = synthetic.small_visual(order=4, n_cp=44, n_landmarks=60, obs_pairs=8, seed=41, span=3.75)
    with ha.Problem(w, lib=hip) as g:
        g.cost()
        bw = g.lib.band_blocks(g.h)
    assert 36 <= bw <= 42, bw
    compare(w, hip, oracle, tol=1e-5)
    w = synthetic.small_visual(order=4, n_cp=60, n_landmarks=40, obs_pairs=10, seed=42, span=5.6)
    with ha.Problem(w, lib=hip) as g:
        with pytest.raises(RuntimeError, match="span too many control points"):
            g.solve(2)


def window_with_band(order, bw, n_cp=48, seed=41, imu=False):
    """A window whose reduced system has exactly `bw` band blocks: one landmark is observed (consistently: pixels re-projected
    through the ground truth) in two segments bw - k apart, every other track is short."""
    w = synthetic.small_inertial(order=order, n_cp=n_cp, n_landmarks=50, obs_pairs=3, n_inertial=120, seed=seed) if imu else \
        synthetic.small_visual(order=order, n_cp=n_cp, n_landmarks=60, obs_pairs=3, seed=seed, span=0.4)
    lo, _ = w.valid_range()
    ta, tb = lo + 1.05 * w.dt, lo + (1.05 + (bw - order)) * w.dt
    T = synthetic.EUROC_CAM_T_BS

    def pixel(l, t, cam):
        qb, pb = synthetic.gt_pose(np.array([t]))
        qs, ps = synthetic.compose(qb, pb, T[cam:cam + 1, :4], T[cam:cam + 1, 4:])
        p_s = np.einsum("nji,nj->ni", synthetic.quat_to_matrix(qs), w.landmarks[l:l + 1] - ps)
        return p_s[0], synthetic.project_radtan(p_s, synthetic.EUROC_CAM_INTRINSICS[cam], synthetic.EUROC_CAM_DISTORTION[cam])[0]

    for l in range(len(w.landmarks)):  # a landmark in front of both cameras at both stamps
        views = [pixel(l, t, cam) for t in (ta, tb) for cam in (0, 1)]
        if all(p[2] > 1.0 and abs(px[0]) < 3000 and abs(px[1]) < 3000 for p, px in views):
            break
    else:
        raise AssertionError("no landmark visible at both ends")
    idx = np.flatnonzero(w.pixel_landmark == l)
    st, px = w.pixel_stamps.copy(), w.pixels.copy()
    for i, (t, cam) in zip(idx[:4], [(ta, 0), (ta, 1), (tb, 0), (tb, 1)]):
        st[i], px[i] = t, pixel(l, t, cam)[1]
        w.pixel_camera[i] = cam
    st[idx[4:]], px[idx[4:]] = ta, pixel(l, ta, 0)[1]
    w.pixel_camera[idx[4:]] = 0
    w.pixel_stamps, w.pixels = st, px
    w.cp_constant = np.r_[np.ones(order, np.uint8), np.zeros(n_cp - order, np.uint8)]
    return w


@pytest.mark.parametrize("bw", list(range(13, 25)) + [31, 32, 33, 39, 40, 41, 42])
def test_every_band_width(bw, hip, oracle):
    """Every band width around the switch-over points of the factorisation kernels (register windows of 96 / 144 / 192 / 256 columns;
    round 1 shipped a defect at exactly 22 band blocks that only the sliding-window replay passed through)."""
    w = window_with_band(4, bw)
    with ha.Problem(w, lib=hip) as g:
        g.cost()
        assert g.lib.band_blocks(g.h) == bw
    compare(w, hip, oracle)


@pytest.mark.parametrize("bw,imu", [(13, False), (14, False), (15, False), (16, False), (14, True), (16, True)])
def test_band_widths_from_both_ends(bw, imu, hip, oracle):
    """Windows of 72 control points (>= 4 band widths): the factorisation runs from both ends — three compute waves up to 14 band blocks,
    four for 15 and 16 — visual-only and bordered; super-block sweeps, border forward sweep in the two-ended elimination order."""
    w = window_with_band(4, bw, n_cp=72, imu=imu)
    with ha.Problem(w, lib=hip) as g:
        g.cost()
        assert g.lib.band_blocks(g.h) == bw and w.n_cp >= 4 * bw
    compare(w, hip, oracle)


@pytest.mark.parametrize("span", [0.02, 0.12, 0.22, 0.32, 0.45, 0.62, 0.78])
def test_narrow_bands_from_both_ends(span, hip, oracle):
    """k_band_factor_mx (round 5: trailing window in f64-MFMA accumulators, 16 phases of a 96-position ring) on the band widths below the
    ones the tests above pin (feature tracks of 0.02 .. 0.78 s: 5 .. 12 control points per landmark): windows of 72 control points are
    factored from both ends for any band of up to 16 control points."""
    w = synthetic.small_visual(order=4, n_cp=72, n_landmarks=90, obs_pairs=3, seed=47, span=span)
    w.cp_constant = np.r_[np.ones(4, np.uint8), np.zeros(68, np.uint8)]
    with ha.Problem(w, lib=hip) as g:
        g.cost()
        bw = g.lib.band_blocks(g.h)
    assert 4 <= bw <= 12 and w.n_cp >= 4 * bw, bw
    compare(w, hip, oracle)


@pytest.mark.parametrize("bw,imu", [(10, False), (14, False), (16, False), (14, True), (16, True)])
def test_mfma_and_valu_factorisations_agree(bw, imu, hip, monkeypatch):
    """The two two-ended factorisations of the library — k_band_factor_mx (default) and k_band_factor_la (measurement switch 64) — on the same
    window: same accept / reject sequence, final state to 1e-9 (two summation orders of one Cholesky factorisation)."""
    w = window_with_band(4, bw, n_cp=72, imu=imu)
    runs = []
    for flags in ("0", "64"):
        monkeypatch.setenv("HS_DEBUG_FLAGS", flags)
        with ha.Problem(w, lib=hip) as g:
            s = g.solve(5)
            runs.append((s, g.control_points().copy(), g.landmarks().copy()))
    (sa, ca, la), (sb, cb, lb) = runs
    assert sa["num_iterations"] == sb["num_iterations"] and sa["num_successful_steps"] == sb["num_successful_steps"]
    assert abs(sa["final_cost"] - sb["final_cost"]) <= 1e-9 * abs(sb["final_cost"])
    assert np.abs(ca - cb).max() <= 1e-9 * max(1.0, np.abs(cb).max()) and np.abs(la - lb).max() <= 1e-8 * max(1.0, np.abs(lb).max())


@pytest.mark.parametrize("bw", [14, 15, 16, 22, 23, 24, 34])
def test_band_widths_order6_bordered(bw, hip, oracle):
    """The same with an order-6 spline and the bordered (inertial) system."""
    w = window_with_band(6, bw, imu=True)
    compare(w, hip, oracle)


def test_knot_uniformity_is_enforced(hip):
    """hs_set_spline: a table with a hole or a shifted knot is refused, last-bit differences are not (tests/util.py; the oracle's CPU test
    runs the same function)."""
    from util import check_knot_uniformity_is_enforced
    check_knot_uniformity_is_enforced(hip)


def test_long_window(hip, monkeypatch):
    """900 control points: the backward sweeps keep the right-hand side in LDS (> 64 KiB here). The two-ended and the one-ended
    factorisation / sweep must agree; windows beyond the LDS budget (> 1024 control points) are rejected with a message."""
    w = synthetic.small_visual(order=4, n_cp=900, n_landmarks=2700, obs_pairs=2, seed=43, with_priors=900)
    sols = []
    for flags in ("0", "2048"):  # 2048: one-ended (measurement switch of the library)
        monkeypatch.setenv("HS_DEBUG_FLAGS", flags)
        with ha.Problem(w, lib=hip) as g:
            s = g.solve(3)
            assert s["num_iterations"] == 3 and np.isfinite(s["final_cost"]) and s["final_cost"] < 0.5 * s["initial_cost"]
            sols.append((s["final_cost"], g.control_points(), g.landmarks()))
    assert abs(sols[0][0] - sols[1][0]) <= 1e-8 * sols[1][0]
    assert rel(sols[0][1], sols[1][1]) < 1e-7 and rel(sols[0][2], sols[1][2]) < 1e-7
    monkeypatch.setenv("HS_DEBUG_FLAGS", "0")
    w = synthetic.small_visual(order=4, n_cp=1100, n_landmarks=200, obs_pairs=2, seed=44)
    with ha.Problem(w, lib=hip) as g:
        with pytest.raises(RuntimeError, match="window too long"):
            g.solve(1)


def test_input_order_invariance(hip, oracle):
    """The residual tables may arrive in any order (the library sorts them landmark- / segment- / bias-segment-major itself):
    sorted by time, reversed and shuffled inputs give the oracle's normal equations and, among themselves, the same system up to
    summation order; landmarks may be permuted too."""
    base = synthetic.small_inertial(order=4, n_cp=24, n_landmarks=40, obs_pairs=3, n_inertial=900, seed=23, identity=False)
    extra = synthetic.small_visual(order=4, n_cp=24, n_landmarks=8, obs_pairs=2, seed=23, with_priors=60)
    lo, hi = base.valid_range()
    base.sensor_T_bs = extra.sensor_T_bs
    base.prior_stamps, base.prior_poses, base.prior_sensor = np.clip(extra.prior_stamps, lo, hi - 1e-9), extra.prior_poses, extra.prior_sensor
    rng = np.random.default_rng(3)

    def reordered(w, how):
        v = copy.deepcopy(w)
        for names in (("pixel_stamps", "pixels", "pixel_landmark", "pixel_camera"), ("prior_stamps", "prior_poses", "prior_sensor"),
                      ("inertial_stamps", "inertial_measurements")):
            n = len(getattr(v, names[0]))
            order = {"sorted": np.argsort(getattr(v, names[0]), kind="stable"), "reversed": np.argsort(getattr(v, names[0]), kind="stable")[::-1],
                     "shuffled": rng.permutation(n)}[how]
            for f in names:
                setattr(v, f, np.ascontiguousarray(getattr(v, f)[order]))
        if how == "shuffled":  # relabel the landmarks as well
            perm = rng.permutation(len(v.landmarks))  # new index of old landmark l is perm[l]
            lm = np.empty_like(v.landmarks)
            lm[perm] = v.landmarks
            v.landmarks, v.pixel_landmark = lm, perm[v.pixel_landmark].astype(np.int32)
        return v

    systems = []
    for how in ("sorted", "reversed", "shuffled"):
        w = reordered(base, how)
        with ha.Problem(w, lib=hip) as g, ha.Problem(w, lib=oracle) as c:
            assert abs(g.cost() - c.cost()) <= 1e-11 * c.cost()
            Sg, gg = g.reduced_system(1e4)
            Sc, gc = c.reduced_system(1e4)
            assert rel(Sg, Sc) < 1e-9 and rel(gg, gc) < 1e-9, (how, rel(Sg, Sc), rel(gg, gc))
            systems.append((Sg, gg))
            sg, sc = g.solve(4), c.solve(4)
            assert sg["num_iterations"] == sc["num_iterations"]
            assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * abs(sc["final_cost"]) + 1e-8 * sc["initial_cost"]
    for S, g_ in systems[1:]:  # the pose-side system does not depend on the order of the tables
        assert rel(S, systems[0][0]) < 1e-11 and rel(g_, systems[0][1]) < 1e-11


def test_table_shapes(hip, oracle):
    """Shapes the other windows do not cover: several prior sensors with mixed indices, bias splines whose knots are not aligned
    with the pose knots (many bias control points, knots inside pose segments), windows one segment longer than the minimum,
    landmarks with a single observation."""
    # three prior sensors, mixed
    w = synthetic.small_visual(order=4, n_cp=14, n_landmarks=20, obs_pairs=2, seed=51, with_priors=45)
    rng = np.random.default_rng(51)
    T = np.repeat(w.sensor_T_bs, 3, axis=0)
    T[1, 4:] += [0.1, -0.05, 0.02]
    T[2, :4] = synthetic.quat_mul(synthetic.quat_exp(np.array([[0.1, -0.2, 0.05]])), T[2:3, :4])[0]
    w.sensor_T_bs, w.prior_sensor = T, rng.integers(0, 3, len(w.prior_stamps)).astype(np.int32)
    compare(w, hip, oracle)
    # bias knots every 0.25 s / 0.37 s on a 0.1 s pose spline
    for bias_dt, order in ((0.25, 4), (0.37, 6)):
        w, r = synthetic._visual_window(synthetic.SEED ^ (0x300 + order), order, 18, 30, 3)
        w = synthetic.add_imu(w, r, 500, bias_dt=bias_dt, identity=False)
        compare(w, hip, oracle, tol=1e-5)
    # one segment more than the minimum
    for k in (4, 6):
        compare(synthetic.small_visual(order=k, n_cp=k + 1, n_landmarks=16, obs_pairs=2, seed=52), hip, oracle)
    # landmarks seen once (one pixel observation: H_ll is rank 2, regularised only by the LM damping)
    w = synthetic.small_visual(order=4, n_cp=16, n_landmarks=40, obs_pairs=3, seed=53)
    keep = np.ones(len(w.pixel_stamps), bool)
    for l in range(0, 40, 4):
        idx = np.flatnonzero(w.pixel_landmark == l)
        keep[idx[1:]] = False
    for f in ("pixel_stamps", "pixels", "pixel_landmark", "pixel_camera"):
        setattr(w, f, np.ascontiguousarray(getattr(w, f)[keep]))
    compare(w, hip, oracle, tol=1e-5, check_lm=False)


@pytest.mark.parametrize("bearing", [False, True])
def test_huber_at_the_threshold(bearing, hip, oracle):
    """Residual norms ON the Huber threshold (pixel 0.5, optimizer.cpp:226; bearing 1.6e-3, :204) and one ulp-scale step either side:
    rho' switches from 1 to a / |r| there; both libraries must take the same branch's value (the two agree at the threshold) and the
    corrector-scaled residual / Jacobian must match."""
    w = synthetic.small_visual(order=4, n_cp=16, n_landmarks=24, obs_pairs=2, bearing=bearing, seed=31)
    a = 1.6e-3 if bearing else 0.5
    ftype = ha.HS_BEARING if bearing else ha.HS_PIXEL
    with ha.Problem(w, lib=oracle) as c:
        r0 = c.linearize(ftype, robustify=False)["r"]
    n = len(r0)
    factors = np.array([1.0, 1.0 - 1e-13, 1.0 + 1e-13, 0.999, 1.001, 2.0, 0.5, 1.0 + 1e-9])[np.arange(n) % 8]
    if bearing:
        # move every measured bearing along the sphere until the angular residual atan2(|p x b|, p . b) equals a * factor: a few
        # Gauss-Newton steps on the oracle with a numerical tangent gradient (tiny problem)
        w2 = copy.copy(w)
        br = np.array(w.bearings, float)
        target = a * factors

        def angles(b):
            w2.bearings = b
            with ha.Problem(w2, lib=oracle) as c:
                return c.linearize(ftype, robustify=False)["r"][:, 0]
        for _ in range(8):
            r = angles(br)
            g = np.zeros_like(br)
            for ax in range(3):
                d = np.zeros(3)
                d[ax] = 1e-7
                g[:, ax] = (angles(br + d) - r) / 1e-7
            g -= (g * br).sum(1, keepdims=True) * br
            br = br - ((r - target) / np.maximum((g * g).sum(1), 1e-30))[:, None] * g
            br /= np.linalg.norm(br, axis=1, keepdims=True)
        w2.bearings = br
    else:
        target = np.zeros_like(r0)
        target[:, 0], target[:, 1] = 0.6 * a * factors, 0.8 * a * factors  # |target| = a * factor (3-4-5: exact in binary up to rounding)
        w2 = copy.copy(w)
        w2.pixels = np.array(w.pixels, float) + (r0 - target)
    with ha.Problem(w2, lib=hip) as g, ha.Problem(w2, lib=oracle) as c:
        raw = c.linearize(ftype, robustify=False)["r"]
        norms = np.linalg.norm(raw, axis=1)
        assert np.abs(norms / a - factors).max() < (1e-6 if bearing else 1e-12)  # the construction hit the threshold neighbourhood
        lg, lc = g.linearize(ftype, robustify=True), c.linearize(ftype, robustify=True)
        for key in ("r", "J_state", "J_landmark", "cost"):
            assert rel(lg[key], lc[key]) < 1e-9, key
        assert abs(g.cost() - c.cost()) <= 1e-12 * c.cost()


@pytest.mark.parametrize("order,bw,n_cp,imu", [(4, 15, 30, False), (4, 16, 24, False), (4, 22, 30, False), (4, 30, 36, True), (6, 33, 41, True), (4, 33, 47, False)])
def test_short_windows_with_window_wide_bands(order, bw, n_cp, imu, hip, oracle, monkeypatch):
    """The shape of the sliding-window replay: a few dozen free control points and tracks as long as the window. These systems take the
    register-resident k_dense_factor (n - frozen <= 2 bw, bw > 14, <= 1024 tile slots); the banded kernels (HS_DEBUG_FLAGS=2097152) must
    give the same step."""
    w = window_with_band(order, bw, n_cp=n_cp, imu=imu)
    compare(w, hip, oracle)
    with ha.Problem(w, lib=hip) as g:
        s1 = g.solve(3)
        cp1 = g.control_points()
    monkeypatch.setenv("HS_DEBUG_FLAGS", "2097152")
    with ha.Problem(w, lib=hip) as g:
        s2 = g.solve(3)
        cp2 = g.control_points()
    assert s1["num_successful_steps"] == s2["num_successful_steps"] and rel(cp1, cp2) < 1e-9
    assert abs(s1["final_cost"] - s2["final_cost"]) <= 1e-9 * s2["final_cost"]
    # Both build paths: the fused build takes windows with more than 256 window tiles (bw > 22) in passes over the tiles (round 5; before, these
    # windows went to the record path), HS_BUILD_PATH=records is the record path — same normal equations (compare: against the oracle), same steps.
    monkeypatch.setenv("HS_DEBUG_FLAGS", "0")
    monkeypatch.setenv("HS_BUILD_PATH", "records")
    compare(w, hip, oracle)
    with ha.Problem(w, lib=hip) as g:
        s3 = g.solve(3)
        cp3 = g.control_points()
    assert s1["num_successful_steps"] == s3["num_successful_steps"] and rel(cp1, cp3) < 1e-9
    assert abs(s1["final_cost"] - s3["final_cost"]) <= 1e-9 * s3["final_cost"]


@pytest.mark.parametrize("flags,what", [(4194304, "k_landmark<K,4,1> instead of k_landmark_rows"), (8388608, "five finalisation launches instead of one"),
                                         (16777216, "k_commit launch instead of the inline copy"), (262144, "no frozen-prefix shortcuts")])
@pytest.mark.parametrize("imu", [False, True])
def test_replay_shape_launch_variants_agree(flags, what, imu, hip, monkeypatch):
    """The launch arrangements added for the sliding-window shape (long tracks, frozen prefix, bordered single shard, small state) against
    the arrangements they replaced (measurement switches of the library): the first three reorder no floating-point operation, so the
    iteration records must be identical; eliminating the frozen prefix instead of skipping it changes rounding only."""
    w = window_with_band(4, 33, n_cp=47, imu=imu)
    w.cp_constant = np.r_[np.ones(12, np.uint8), np.zeros(47 - 12, np.uint8)]  # a frozen prefix as in a slid window
    runs = []
    for f in ("0", str(flags)):
        monkeypatch.setenv("HS_DEBUG_FLAGS", f)
        with ha.Problem(w, lib=hip) as g:
            s = g.solve(4)
            runs.append((s, g.control_points(), g.landmarks()))
    monkeypatch.setenv("HS_DEBUG_FLAGS", "0")
    (s0, cp0, lm0), (s1, cp1, lm1) = runs
    assert s0["num_iterations"] == s1["num_iterations"] and s0["num_successful_steps"] == s1["num_successful_steps"], what
    if flags == 262144:
        assert rel(cp0, cp1) < 1e-9 and rel(lm0, lm1) < 1e-9 and abs(s0["final_cost"] - s1["final_cost"]) <= 1e-9 * s1["final_cost"], what
    else:
        assert np.array_equal(cp0, cp1) and np.array_equal(lm0, lm1) and s0["final_cost"] == s1["final_cost"], what


@pytest.mark.parametrize("imu", [False, True])
def test_result_readback_through_the_pinned_cache(imu, hip):
    """A caller that reads the state after a solve gets the next solve's result copied into pinned host memory inside hs_solve; the getters
    must return exactly what a direct device read returns (also after hs_restore, which invalidates the cache)."""
    w = synthetic.small_inertial(order=4, n_cp=20, seed=5) if imu else synthetic.small_visual(order=4, n_cp=20, seed=5)

    def state(g):
        out = [g.control_points(), g.landmarks()]
        if imu:
            out += list(g.bias()) + [g.gravity()]
        return out

    with ha.Problem(w, lib=hip) as a, ha.Problem(w, lib=hip) as b:
        a.snapshot()
        a.solve(2)
        state(a)            # device path; switches the read-back cache on
        a.solve(2)
        cached = state(a)   # served from the cache
        b.solve(2)
        b.solve(2)
        direct = state(b)   # never cached: device path
        for x, y in zip(cached, direct):
            assert np.array_equal(x, y)
        a.restore()
        restored = state(a)
        with ha.Problem(w, lib=hip) as c:
            initial = state(c)
        for x, y in zip(restored, initial):
            assert np.array_equal(x, y)


def perturbed_visual(seed, lm_noise, cp_noise, order=4, bearing=False, n_lm=120):
    w, _ = synthetic._visual_window(seed, order, 20, n_lm, 3, bearing=bearing, lm_noise=lm_noise, span=1.0)
    rng = np.random.default_rng(seed)
    w.control_points = w.control_points.copy()
    w.control_points[:, 4:7] += cp_noise * rng.standard_normal((w.control_points.shape[0], 3))
    return w


@pytest.mark.parametrize("seed,lm_noise,cp_noise,order,bearing,n_lm", [(1, 0.5, 0.3, 4, False, 120), (4, 0.5, 0.3, 4, False, 120), (1, 2.0, 0.0, 4, False, 120),
                                                                     (3, 2.0, 0.3, 6, False, 120), (1, 0.5, 0.6, 4, True, 120),
                                                                     (1, 0.5, 0.3, 4, False, 1500), (4, 0.5, 0.3, 4, False, 1500)])
def test_rejected_steps_on_visual_windows(seed, lm_noise, cp_noise, order, bearing, n_lm, hip, oracle, monkeypatch):
    """Rejected steps on visual-only windows, on both build paths of the library. Fused build (default): the current point is linearised,
    eliminated and accumulated by k_build_visual at the top of every iteration, with the radius the decision left behind. Record path
    (HS_BUILD_PATH=records): solves linearise at the candidate point and keep the records of the current point across a rejected step
    (capi.hip: speculative_solve). On both, above 4096 state scalars (the 1500-landmark cases) the accepted candidate is copied to x by the
    next iteration's k_backsub_retract instead of a k_commit launch. Starts far enough from the optimum that steps are rejected: the accept /
    reject sequence, every recorded quantity and the final state must match the oracle's (which linearises the current point at the top of
    every iteration), the path with a commit per iteration (HS_DEBUG_FLAGS=67108864) and, on the record path, the one that does
    everything the oracle's way on the device (1073741824)."""
    w = perturbed_visual(seed, lm_noise, cp_noise, order, bearing, n_lm)
    n_it = 6
    with ha.Problem(w, lib=oracle) as c:
        sc = c.solve(n_it)
        cp_c, lm_c = c.control_points(), c.landmarks()
    flags_seen = [it["step_is_successful"] for it in sc["iterations"]]
    assert 0 in flags_seen[1:] and 1 in flags_seen[1:], flags_seen  # the case exercises both branches
    outs = []
    for path, flags in (("fused", "0"), ("fused", "67108864"), ("records", "0"), ("records", "67108864"), ("records", "1073741824")):
        monkeypatch.setenv("HS_DEBUG_FLAGS", flags)
        monkeypatch.setenv("HS_BUILD_PATH", path)
        with ha.Problem(w, lib=hip) as g:
            sg = g.solve(n_it)
            outs.append((sg, g.control_points(), g.landmarks()))
        flags = (path, flags)
        assert [it["step_is_successful"] for it in sg["iterations"]] == flags_seen, flags
        assert sg["termination"] == sc["termination"] and sg["num_successful_steps"] == sc["num_successful_steps"]
        for ig, ic in zip(sg["iterations"], sc["iterations"]):
            assert abs(ig["cost"] - ic["cost"]) <= 1e-6 * abs(ic["cost"]) + 1e-8 * sc["initial_cost"], (flags, ig["iteration"], ig["cost"], ic["cost"])
            for k in ("radius", "step_norm", "relative_decrease"):
                assert abs(ig[k] - ic[k]) <= 1e-5 * max(abs(ic[k]), 1e-12), (flags, ig["iteration"], k, ig[k], ic[k])
        assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
        assert rel(outs[-1][1], cp_c) < 1e-6 and rel(outs[-1][2], lm_c) < 1e-6
    monkeypatch.delenv("HS_BUILD_PATH")
    monkeypatch.setenv("HS_DEBUG_FLAGS", "0")
    # with and without the deferred commit a path evaluates the same expressions at the same points
    for i, j in ((0, 1), (2, 3)):
        assert rel(outs[i][1], outs[j][1]) < 1e-9 and rel(outs[i][2], outs[j][2]) < 1e-9
        for a, b in zip(outs[i][0]["iterations"], outs[j][0]["iterations"]):
            assert abs(a["cost"] - b["cost"]) <= 1e-12 * abs(b["cost"])


def test_deferred_commit_when_a_solve_converges_early(hip, oracle):
    """A speculative solve that stops on a convergence test before max_iterations: the last accepted candidate must be in x (the commit of
    an accepted step is deferred to the next iteration's k_backsub_retract, which a finished solve no longer runs)."""
    w = perturbed_visual(7, 0.05, 0.02, 4, False, 1500)
    with ha.Problem(w, lib=oracle) as c, ha.Problem(w, lib=hip) as g:
        sc, sg = c.solve(25), g.solve(25)
        assert sc["num_iterations"] < 25 and sg["num_iterations"] == sc["num_iterations"] and sg["termination"] == sc["termination"]
        assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-9 * sc["final_cost"]
        assert rel(g.control_points(), c.control_points()) < 1e-7 and rel(g.landmarks(), c.landmarks()) < 1e-7
        assert abs(g.cost() - sg["final_cost"]) <= 1e-12 * sg["final_cost"]  # x is the point the summary reports


def test_guarded_tables(tmp_path):
    """HS_GUARD=1 (host_tables.hpp GuardRegistry: every device table at its exact size with a pattern behind it, checked before hs_solve / hs_cost /
    hs_reduced_system return), in a process of its own — the mode is read once per process: windows of inertial residuals and of priors only with
    13 and 14 control points (bands of four control points: the inverted super-blocks of the backward sweep need more room than np x ncb — the
    overflow this mode found), an IMU window with thousands of inertial residuals next to the visual ones (the cost-partial tables the randomised
    sweep found too short), and 60 random window shapes of tools/fuzz_parity.py."""
    import subprocess
    import sys
    script = tmp_path / "guarded.py"
    script.write_text("""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import hyperslam_amd as ha
from hyperslam_amd import synthetic
for n_cp, imu in ((14, True), (13, True), (13, False), (18, False)):
    w = synthetic.small_inertial(order=4, n_cp=n_cp, n_landmarks=10, obs_pairs=2, n_inertial=120, seed=37) if imu else \
        synthetic.small_visual(order=4, n_cp=n_cp, n_landmarks=10, obs_pairs=2, seed=37, with_priors=40)
    for name in ("pixel_stamps", "pixels", "pixel_landmark", "pixel_camera"):
        setattr(w, name, getattr(w, name)[:0])
    w.landmarks = w.landmarks[:0]
    with ha.Problem(w) as g:
        g.cost(); g.reduced_system(1e4); s = g.solve(4)
        assert np.isfinite(s["final_cost"])
w = synthetic.small_inertial(order=4, n_cp=120, n_landmarks=2500, obs_pairs=4, n_inertial=6000, seed=5)
with ha.Problem(w) as g:
    g.cost(); s = g.solve(3)
    assert s["final_cost"] < s["initial_cost"]
print("guarded windows ok")
""" % (ROOT, os.path.join(ROOT, "tools")))
    env = dict(os.environ, HS_GUARD="1")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "guarded windows ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "60", "92"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0 and "60 cases, 0 failures" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
