"""CPU tests (no GPU): the oracle against the committed 50-digit golden vectors and its own reference-style gradient probe;
host-side structure; synthetic generators; the C ABI library loads and exports every declared symbol."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import hyperslam_amd as ha
from hyperslam_amd import _lib, synthetic
from util import check_against_golden, check_manifolds_against_golden, check_tracks_against_golden, golden_cases, golden_window, rel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_golden_file_present_and_complete():
    cases = golden_cases()
    kinds = {(c["type"], c["inputs"]["k"]) for c in cases}
    assert kinds == {(t, k) for t in ("pixel", "bearing", "prior", "inertial") for k in (4, 6)}


@pytest.mark.parametrize("idx", range(len(golden_cases())))
def test_oracle_matches_golden(idx, oracle):
    case = golden_cases()[idx]
    with ha.Problem(golden_window(case), lib=oracle) as p:
        # pixel Jacobians reach 1e3 in magnitude: relative 1e-9 == the reference's own 1e-5 probe, four digits tighter
        check_against_golden(p, case, 1e-9)


def test_oracle_matches_golden_order5(oracle):
    """Order 5 (instantiated on the device since round 4): 48 independent 100-digit cases, 12 per factor, every parameter block."""
    from util import golden_cases_k5
    cases = golden_cases_k5()
    assert len(cases) == 48 and {c["inputs"]["k"] for c in cases} == {5}
    for case in cases:
        with ha.Problem(golden_window(case), lib=oracle) as p:
            check_against_golden(p, case, 1e-9)


def test_oracle_matches_literal_inertial_golden(oracle):
    """The default inertial Jacobian (as written upstream) off the identity point: 32 cases, orders 4 and 6."""
    from util import check_against_literal_golden, literal_inertial_cases
    cases = literal_inertial_cases()
    assert len(cases) == 32 and {c["inputs"]["k"] for c in cases} == {4, 6}
    worst = {}
    for case in cases:
        with ha.Problem(golden_window(case), lib=oracle) as p:
            for k, v in check_against_literal_golden(p, case, 1e-9).items():
                worst[k] = max(worst.get(k, 0.0), v)
    assert max(worst.values()) < 1e-9, worst


def test_oracle_gradient_probe():
    """Mirror of the reference's four `Gradients` tests (tests/internal/tests/optimizers/evaluators/*.cpp)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "selftest"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "oracle", "selftest")], capture_output=True, text=True)
    assert out.returncode == 0 and "SELFTEST OK" in out.stdout, out.stdout[-2000:]


def test_knot_uniformity_is_enforced(oracle):
    from util import check_knot_uniformity_is_enforced
    check_knot_uniformity_is_enforced(oracle)


def test_layout_matches_exteroceptive_update(oracle):
    """Block structure of ExteroceptiveCost::update (exteroceptive.cpp:25-99): sizes, offsets, indices, counts."""
    w = synthetic.small_visual(order=4, n_cp=12, n_landmarks=5, obs_pairs=2, with_priors=3)
    with ha.Problem(w, lib=oracle) as p:
        L = p.residual_layout(ha.HS_PIXEL, 0)
        assert L["sizes"].tolist() == [8, 8, 8, 8, 7, 4, 4, 3] and L["num_parameters"] == 8 * 4 + 18 and L["num_residuals"] == 2
        assert L["offsets"].tolist() == [0, 8, 16, 24, 32, 39, 43, 47] and L["indices"].tolist() == [0, 4, 7, 7]
        L = p.residual_layout(ha.HS_PRIOR, 1)
        assert L["sizes"].tolist() == [8] * 4 + [7] and L["num_residuals"] == 6 and L["indices"].tolist() == [0, 4, 5, 5]
        first = L["block_ids"][0]
        assert L["block_ids"][:4].tolist() == list(range(first, first + 4))


def test_oracle_lm_converges_and_is_deterministic(oracle):
    w = synthetic.small_visual(order=4, n_cp=14, n_landmarks=30, obs_pairs=3, with_priors=10)
    runs = []
    for _ in range(2):
        with ha.Problem(w, lib=oracle) as p:
            s = p.solve(5)
            runs.append((s["final_cost"], p.control_points().copy()))
            assert s["final_cost"] < 0.2 * s["initial_cost"] and s["num_iterations"] == 5
            costs = [it["cost"] for it in s["iterations"]]
            assert all(b <= a * (1 + 1e-12) for a, b in zip(costs, costs[1:]))  # monotonic steps
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1])


def test_frozen_control_points_do_not_move(oracle):
    w = synthetic.small_visual(order=4, n_cp=14, n_landmarks=30, obs_pairs=3)
    w.cp_constant = np.r_[np.ones(5, np.uint8), np.zeros(9, np.uint8)]
    with ha.Problem(w, lib=oracle) as p:
        p.solve(3)
        cp = p.control_points()
    assert np.array_equal(cp[:5], w.control_points[:5]) and not np.array_equal(cp[5:], w.control_points[5:])


def test_synthetic_configs_are_deterministic_and_sized():
    a, b = synthetic.config1(), synthetic.config1()
    assert a.num_residual_blocks() == 50000 and len(a.landmarks) == 5000 and a.n_cp == 128
    assert np.array_equal(a.pixels, b.pixels) and np.array_equal(a.control_points, b.control_points)
    lo, hi = a.valid_range()
    assert a.pixel_stamps.min() >= lo and a.pixel_stamps.max() < hi
    c0 = synthetic.config0()
    assert c0.n_cp == 32 and len(c0.prior_stamps) == 1000
    # SplitMix64 known answer (first outputs for seed 0 of the published algorithm)
    r = synthetic.SplitMix64(0)
    assert [int(x) for x in r._raw(3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_shard_by_landmark_partitions_residuals():
    w = synthetic.small_visual(order=4, n_cp=16, n_landmarks=37, obs_pairs=3, with_priors=9)
    shards = [synthetic.shard_by_landmark(w, r, 3) for r in range(3)]
    assert sum(len(s.pixel_stamps) for s in shards) == len(w.pixel_stamps)
    assert sum(len(s.prior_stamps) for s in shards) == len(w.prior_stamps)
    owners = [set(s.pixel_landmark.tolist()) for s in shards]
    assert not (owners[0] & owners[1]) and not (owners[0] & owners[2]) and not (owners[1] & owners[2])


def test_product_library_exports_every_declared_symbol():
    """libhyperslam_hip.so loads on a machine without a GPU and exports each function include/hyperslam_hip.h declares."""
    header = open(os.path.join(ROOT, "include", "hyperslam_hip.h")).read()
    declared = set(re.findall(r"\b(hs_[a-z_]+)\s*\(", header)) - {"hs_allreduce_fn"}
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    assert os.path.exists(_lib.PRODUCT_LIB), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.PRODUCT_LIB)
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.hs_arch.restype = ctypes.c_char_p
    assert lib.hs_arch() == b"gfx950"


def test_c_header_is_plain_c99():
    """The drop-in boundary is a C ABI: the header must be consumable by a C compiler (cgo / JNI / ctypes generators), not only by C++."""
    header = os.path.join(ROOT, "include", "hyperslam_hip.h")
    for std, cc in (("-std=c99", "gcc"), ("-std=c++17", "g++")):
        out = subprocess.run([cc, std, "-fsyntax-only", "-Wall", "-Wextra", "-pedantic", "-x", "c" if cc == "gcc" else "c++", header], capture_output=True, text=True)
        assert out.returncode == 0 and not out.stderr.strip(), out.stderr


def test_reference_side_plugin_header_uses_only_the_declared_abi():
    """include/hyper/optimizers/hip/optimizer.hpp (Optimizer<OptimizerSuite::HIP>, compiled inside the HyperSLAM tree, not here) may call
    only functions the C header declares and the library exports, and must override every pure virtual of AbstractOptimizer
    (/root/reference/include/hyper/optimizers/abstract.hpp:53-139)."""
    shim = open(os.path.join(ROOT, "include", "hyper", "optimizers", "hip", "optimizer.hpp")).read()
    code = "\n".join(line.split("//")[0] for line in shim.splitlines())  # comments stripped
    called = set(re.findall(r"\b(hs_[a-z_]+)\s*\(", code))
    header = open(os.path.join(ROOT, "include", "hyperslam_hip.h")).read()
    declared = set(re.findall(r"\b(hs_[a-z_]+)\s*\(", header))
    assert called and called <= declared, called - declared
    lib = ctypes.CDLL(_lib.PRODUCT_LIB)
    for sym in called:
        assert hasattr(lib, sym), sym
    for virtual in ("swapEnvironment", "swapState", "add(VisualBearingObservation&", "add(VisualPixelObservation&", "add(ManifoldObservation<Manifold>&",
                    "add(InertialObservation<Manifold>&", "hasSensor", "setGravityConstant", "optimize()", "updateState", "addLandmark", "updateLandmarks",
                    "updateSensor"):
        assert re.search(r"auto\s+" + re.escape(virtual) + r"[^;{]*\bfinal\b", code), virtual
    # (round 6: the residual and landmark tables go through the delta interface — one row per add() / addLandmark, retired in updateLandmarks, staged
    #  between solves — as the Ceres backend keeps its problem; optimize() sends what a window change touches)
    for needed in ("hs_create", "hs_destroy", "hs_set_spline", "hs_set_cameras", "hs_append_landmarks", "hs_append_bearing_residuals", "hs_append_pixel_residuals",
                   "hs_append_prior_residuals", "hs_append_inertial_residuals", "hs_retire_landmarks", "hs_retire_residuals_before", "hs_stage", "hs_set_imu",
                   "hs_set_gravity", "hs_solve", "hs_get_control_points", "hs_get_landmarks", "hs_get_bias", "hs_get_gravity"):
        assert needed in called, needed


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ha.HsError):
        ha.Problem(synthetic.small_visual())


def test_process_tracks_round_trip(oracle):
    """Front half of AbstractOptimizer::process(VisualTracks): bearings are unit vectors that re-project onto the pixels and the
    triangulated midpoint recovers a synthetic point seen by both cameras."""
    from hyperslam_amd import synthetic
    w = synthetic.small_visual(order=4, n_cp=12, n_landmarks=8, obs_pairs=2)
    lo, hi = w.valid_range()
    stamp = 0.5 * (lo + hi)
    rng = np.random.default_rng(3)
    n = 50
    # points in front of camera 0, projected through the radtan model into both cameras
    px0 = np.stack([rng.uniform(100, 650, n), rng.uniform(80, 400, n)], -1)
    with ha.Problem(w, lib=oracle) as p:
        b0, _, _ = p.process_tracks(stamp, px0, px0)
        assert np.allclose(np.linalg.norm(b0, axis=1), 1.0, atol=1e-14)
        depth = rng.uniform(2.0, 8.0, n)
        p0 = b0 / b0[:, 2:3] * depth[:, None]                      # sensor frame of camera 0
        T0, T1 = w.cam_T_bs[0], w.cam_T_bs[1]
        R0, R1 = synthetic.quat_to_matrix(T0[None, :4])[0], synthetic.quat_to_matrix(T1[None, :4])[0]
        pb = p0 @ R0.T + T0[4:]
        p1 = (pb - T1[4:]) @ R1
        px1 = synthetic.project_radtan(p1, np.broadcast_to(w.cam_intrinsics[1], (n, 4)), np.broadcast_to(w.cam_distortion[1], (n, 4)))
        re0 = synthetic.project_radtan(p0, np.broadcast_to(w.cam_intrinsics[0], (n, 4)), np.broadcast_to(w.cam_distortion[0], (n, 4)))
        assert np.abs(re0 - px0).max() < 1e-6                       # undistortion inverts the projection
        b0, b1, pw = p.process_tracks(stamp, px0, px1)
        pose = p.sample_trajectory([stamp])[0]
        Rwb = synthetic.quat_to_matrix(pose[None, :4])[0]
        assert np.abs(pw - (pb @ Rwb.T + pose[4:])).max() < 1e-6    # midpoint triangulation recovers the point


def test_oracle_manifolds_match_golden(oracle):
    """Manifold::Plus / PlusJacobian of every variable class (wrapper.hpp:32-38) against the 100-digit vectors, incl. the
    delta = 0, |delta| > pi/2 and SphereManifold pivot branches (tests/golden/make_manifold_golden.py)."""
    with ha.Problem(synthetic.small_visual(), lib=oracle) as p:
        assert check_manifolds_against_golden(p, 1e-14) <= 1e-14
        with pytest.raises(ha.HsError):
            p.manifold_plus(ha.HS_MANIFOLD_SPHERE3, np.zeros((1, 4)), np.zeros((1, 2)))


def test_oracle_process_tracks_matches_golden(oracle):
    """Pixel -> bearing (20 fixed-point undistortion steps vs the exact root) and stereo triangulation against 100-digit vectors."""
    assert check_tracks_against_golden(oracle, 1e-10) <= 1e-10


def test_bench_helpers(oracle):
    """bench.py plumbing that runs without a GPU: natural ordering of the committed profile files (r01_v12 after r01_v7) and the
    all-cores CPU leg's worker process (one oracle optimize() started at a common wall-clock instant, result as one JSON line)."""
    import json
    import sys
    import time
    sys.path.insert(0, ROOT)
    import bench
    names = [os.path.basename(f) for f in bench._newest("r*_bench_kernel_stats.csv")]
    keys = [[int(x) for x in re.findall(r"\d+", n)] for n in names]
    assert keys == sorted(keys) and len(names) >= 2
    assert bench.rocprof_kernel_ms("void hs::k_build_visual<4>") > 0 and bench.pmc_traffic("hs::k_build_visual<4>") > 1e6  # (newest committed profile: the fused build)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", oracle.path, repr(time.time() + 1.0)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["iters"] == bench.LM_ITERATIONS and rec["late"] in (0, 1) and rec["end"] > time.time() - 300


def _mixed_window():
    w = synthetic.small_inertial(order=4, n_cp=16, n_landmarks=30, obs_pairs=3, n_inertial=200, seed=61, identity=False)
    e = synthetic.small_visual(order=4, n_cp=16, n_landmarks=4, obs_pairs=2, seed=61, with_priors=30)
    lo, hi = w.valid_range()
    w.sensor_T_bs, w.prior_stamps, w.prior_poses, w.prior_sensor = e.sensor_T_bs, np.clip(e.prior_stamps, lo, hi - 1e-9), e.prior_poses, e.prior_sensor
    return w


def test_oracle_world_frame_invariance(oracle):
    """Physics check that needs no reference values: moving the whole world by a rigid transform G (control points, landmarks,
    gravity, pose-prior measurements) leaves every pixel / inertial residual and the total cost unchanged (the cumulative SU2 x R^3
    spline is left-equivariant); pose-prior residuals are expressed in the world frame, so they rotate but keep their norms."""
    import copy
    w = _mixed_window()
    qg = synthetic.quat_exp(np.array([[0.7, -0.4, 1.1]]))[0]
    Rg, tg = synthetic.quat_to_matrix(qg[None])[0], np.array([3.0, -2.0, 0.5])
    v = copy.deepcopy(w)
    for tab in ("control_points", "prior_poses"):
        a = getattr(v, tab).copy()
        a[:, :4] = synthetic.quat_mul(np.broadcast_to(qg, (len(a), 4)), a[:, :4])
        a[:, 4:7] = a[:, 4:7] @ Rg.T + tg
        setattr(v, tab, a)
    v.landmarks, v.gravity = w.landmarks @ Rg.T + tg, Rg @ w.gravity
    with ha.Problem(w, lib=oracle) as a, ha.Problem(v, lib=oracle) as b:
        assert abs(a.cost() - b.cost()) <= 1e-12 * a.cost()
        for t in (ha.HS_PIXEL, ha.HS_INERTIAL):
            ra, rb = a.linearize(t, robustify=False)["r"], b.linearize(t, robustify=False)["r"]
            assert np.abs(ra - rb).max() <= 1e-11 * np.abs(ra).max()
        ra, rb = a.linearize(ha.HS_PRIOR, robustify=False)["r"], b.linearize(ha.HS_PRIOR, robustify=False)["r"]
        assert np.abs(np.linalg.norm(ra, axis=1) - np.linalg.norm(rb, axis=1)).max() <= 1e-11


def test_oracle_time_shift_invariance(oracle):
    """Shifting every stamp (control points, bias control points, residuals) by the same amount changes nothing."""
    import copy
    w = _mixed_window()
    lo, hi = w.valid_range()
    for f in ("pixel_stamps", "prior_stamps", "inertial_stamps"):  # keep clear of the window ends: (t - t0) / dt at a knot is not
        setattr(w, f, np.clip(getattr(w, f), lo + 1e-6, hi - 1e-6))  # shift-invariant in floating point (0.1 is not a binary fraction)
    v, shift = copy.deepcopy(w), 8.0
    v.t0 = w.t0 + shift
    cp = v.control_points.copy()
    cp[:, 7] += shift
    v.control_points = cp
    for f in ("pixel_stamps", "prior_stamps", "inertial_stamps"):
        setattr(v, f, getattr(w, f) + shift)
    v.imu = dict(w.imu)
    v.imu["bias_t0"] = w.imu["bias_t0"] + shift
    for key in ("bias_g", "bias_a"):
        bb = np.array(w.imu[key], float).copy()
        bb[:, 3] += shift
        v.imu[key] = bb
    with ha.Problem(w, lib=oracle) as a, ha.Problem(v, lib=oracle) as b:
        assert abs(a.cost() - b.cost()) <= 1e-9 * a.cost()
        for t in (ha.HS_PIXEL, ha.HS_PRIOR, ha.HS_INERTIAL):
            la, lb = a.linearize(t, robustify=True), b.linearize(t, robustify=True)
            assert np.array_equal(la["first_cp"], lb["first_cp"])
            for key in ("r", "J_state"):
                assert rel(la[key], lb[key]) < 1e-8, (t, key, rel(la[key], lb[key]))
        sa, sb = a.solve(3), b.solve(3)
        assert abs(sa["final_cost"] - sb["final_cost"]) <= 1e-6 * sa["final_cost"]


def test_reference_vectors_overlay_is_what_the_golden_tests_compare_with(oracle, tmp_path, monkeypatch):
    """HS_REFERENCE_VECTORS=<dir> (outputs of the real reference written by tools/reference_dump.cpp) replaces the outputs of the
    golden cases: a faithful copy passes — inertial cases then in the as-written mode, which the as-written vectors of inertial_literal.json
    stand in for here — and a copy with one residual changed fails, i.e. the overlay is what is compared."""
    import json

    import util

    plain = util.golden_cases()
    with open(os.path.join(util.HERE, "golden", "inertial_literal.json")) as f:
        literal = json.load(f)["cases"]
    ref = {"cases": [{"type": c["type"], "outputs": c["outputs"]} for c in plain]}
    (tmp_path / "factors.json").write_text(json.dumps(ref))
    (tmp_path / "inertial_literal.json").write_text(json.dumps({"cases": [{"type": c["type"], "outputs": c["outputs"]} for c in literal]}))
    monkeypatch.setenv("HS_REFERENCE_VECTORS", str(tmp_path))
    cases = util.golden_cases()
    assert all(c.get("reference") for c in cases)
    pixel = next(c for c in cases if c["type"] == "pixel")
    with ha.Problem(golden_window(pixel), lib=oracle) as p:
        check_against_golden(p, pixel, 1e-9)
    lit = util.literal_inertial_cases()
    assert lit[0].get("reference")
    with ha.Problem(golden_window(lit[0]), lib=oracle) as p:  # a reference vector of an inertial factor is compared in the default mode only
        lit[0]["variant"] = None
        check_against_golden(p, {**lit[0], "outputs": {k: lit[0]["outputs"][k] for k in ("r", "J_state", "J_extrinsics", "J_gravity", "J_acc_offsets")}}, 1e-9)
    ref["cases"][plain.index(next(c for c in plain if c["type"] == "pixel"))]["outputs"] = dict(pixel["outputs"], r=[x + 1e-6 for x in pixel["outputs"]["r"]])
    (tmp_path / "factors.json").write_text(json.dumps(ref))
    changed = next(c for c in util.golden_cases() if c["type"] == "pixel")
    with ha.Problem(golden_window(changed), lib=oracle) as p, pytest.raises(AssertionError):
        check_against_golden(p, changed, 1e-9)
