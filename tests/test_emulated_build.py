"""CPU tests (no GPU): the fused build kernel — linearisation, landmark elimination and both Gram terms in one pass
(hyperslam_amd/csrc/kernels_build.hpp) followed by k_assemble and k_finalize_reduced — compiled from the product's kernel SOURCES for the
host (tests/emul/: one thread per lane, barriers and wave exchanges emulated) and compared with the oracle's reduced normal equations of
the same window. This pins the kernel's index arithmetic, LDS layout and summation structure before a GPU is involved; the `-m gpu`
tests run the same comparison on the real device through the C ABI."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import hyperslam_amd as ha
from hyperslam_amd import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "emul")


@pytest.fixture(scope="session")
def harness():
    exe = os.path.join(EMUL, "build_harness")
    srcs = [os.path.join(EMUL, "build_harness.cpp"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    csrc = os.path.join(ROOT, "hyperslam_amd", "csrc")
    srcs += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in srcs):
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-pthread", "-I", EMUL, "-o", exe, os.path.join(EMUL, "build_harness.cpp")])
    return exe


def _pack_cameras(w):
    n = len(w.cam_T_bs)
    cam = np.zeros((n, 16))
    cam[:, :7], cam[:, 7:11], cam[:, 11:15] = w.cam_T_bs, w.cam_intrinsics, w.cam_distortion
    return cam


def run_emulated_build(exe, w, radius=1e4, R=0, L=0, scaling=None, fold=False, full=False):
    """Returns dict(S, g, cost, bw, n_chunk, Y, lm_scale, scale_p) of the emulated fused build on window `w`."""
    n_cp, n_lm = w.n_cp, len(w.landmarks)
    n_px, n_br = len(w.pixel_stamps), len(w.bearing_stamps)
    f64, i32 = np.float64, np.int32
    cpc = np.zeros(n_cp, i32) if w.cp_constant is None else np.asarray(w.cp_constant, i32)
    lmc = np.zeros(n_lm, i32) if w.landmark_constant is None else np.asarray(w.landmark_constant, i32)
    hdr = np.array([w.order, n_cp, n_lm, n_px, n_br, len(w.cam_T_bs), int(w.rotation_constant), int(w.translation_constant), R, L,
                    1 if scaling is not None else 0, (1 if fold else 0) | (2 if full else 0)], i32)
    parts = [hdr, np.array([w.t0, w.dt, radius], f64), np.asarray(w.control_points, f64), cpc, _pack_cameras(w), np.asarray(w.landmarks, f64), lmc,
             np.asarray(w.pixel_stamps, f64), np.asarray(w.pixels, f64), np.asarray(w.pixel_landmark, i32), np.asarray(w.pixel_camera, i32),
             np.asarray(w.bearing_stamps, f64), np.asarray(w.bearings, f64), np.asarray(w.bearing_landmark, i32), np.asarray(w.bearing_camera, i32)]
    if scaling is not None:
        parts += [np.asarray(scaling[0], f64), np.asarray(scaling[1], f64)]
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "w.bin"), os.path.join(tmp, "o.bin")
        with open(fin, "wb") as f:
            for a in parts:
                f.write(np.ascontiguousarray(a).tobytes())
        subprocess.check_call([exe, fin, fout], timeout=600)
        raw = open(fout, "rb").read()
    bw, np_, n_chunk, R_, L_, y_total, lds, n_upd = struct.unpack("8i", raw[:32])
    off = 32
    cost = struct.unpack("d", raw[off:off + 8])[0]
    off += 8
    ncb = 6 * bw

    def take(n):
        nonlocal off
        a = np.frombuffer(raw, np.float64, n, off).copy()
        off += 8 * n
        return a

    Sb, g, Y = take(np_ * ncb).reshape(np_, ncb), take(np_), take(y_total)
    lm_scale, scale_p = take(3 * n_lm), take(np_)
    upd = take(n_upd)
    S = np.zeros((np_, np_))
    for rho in range(np_):
        c0 = 6 * (rho // 6)
        for c in range(min(ncb, np_ - c0)):
            if c0 + c >= rho:
                S[rho, c0 + c] = S[c0 + c, rho] = Sb[rho, c]
    return dict(S=S, g=g, cost=cost, bw=bw, n_chunk=n_chunk, R=R_, L=L_, Y=Y, lm_scale=lm_scale, scale_p=scale_p, lds=lds, upd=upd)


def _check(exe, oracle, w, radius=1e4, tol=1e-9, **kw):
    out = run_emulated_build(exe, w, radius, **kw)
    with ha.Problem(w, lib=oracle) as p:
        S, g = p.reduced_system(radius)
        cost = p.cost()
    n = out["S"].shape[0]
    scale = np.abs(S[:n, :n]).max()
    assert np.abs(out["S"] - S[:n, :n]).max() <= tol * scale, (np.abs(out["S"] - S[:n, :n]).max() / scale, out["n_chunk"])
    assert np.abs(out["g"] - g[:n]).max() <= tol * max(1.0, np.abs(g).max())
    assert abs(out["cost"] - cost) <= 1e-11 * cost
    # the candidate point of a (fabricated) step two ways: k_update_visual (per chunk: candidate control points of the window, landmark
    # back-substitution, candidate cost) against k_backsub_retract + k_cost_visual — landmarks, control points, cost, decision terms, norms
    u = out["upd"]
    assert u[0] <= 1e-13 and u[1] == 0.0, u[:2]
    for a, b in zip(u[2::2], u[3::2]):
        assert abs(a - b) <= 1e-12 * max(abs(a), 1e-300), (a, b)
    return out


@pytest.mark.parametrize("order,bearing", [(4, False), (4, True), (6, False), (5, False)])
def test_emulated_fused_build_matches_oracle(harness, oracle, order, bearing):
    w = synthetic.small_visual(order=order, n_cp=14 if order == 4 else 16, n_landmarks=40, obs_pairs=3, bearing=bearing)
    out = _check(harness, oracle, w)
    assert out["n_chunk"] >= 2


@pytest.mark.parametrize("order,span,pairs", [(4, 0.05, 5), (6, 0.15, 4)])
def test_emulated_fused_build_crowded_segments(harness, oracle, order, span, pairs):
    """Every observation of a landmark inside one or two knot intervals: the records of a chunk crowd into a few segments, the J'J tiles of those
    segments get eight or sixteen record streams from the per-chunk deal (kernels_build.hpp, streams of phase 3: the fourth butterfly level)
    and most other tiles a single lane with nothing to walk."""
    w = synthetic.small_visual(order=order, n_cp=14 if order == 4 else 16, n_landmarks=36, obs_pairs=pairs, span=span)
    out = _check(harness, oracle, w)
    assert out["n_chunk"] >= 3


def test_emulated_fused_build_small_chunks(harness, oracle):
    """Forced tiny geometry: chunks of <= 3 landmarks and <= 64 records (landmarks with 40 residuals: one or two per chunk), records of one
    landmark spread over several waves."""
    w = synthetic.small_visual(order=4, n_cp=16, n_landmarks=12, obs_pairs=20, span=0.6)
    out = _check(harness, oracle, w, R=64, L=3)
    assert out["R"] == 64 and out["n_chunk"] >= 6


@pytest.mark.parametrize("order,n_cp,span,n_lm,pairs,R,L", [(4, 42, 3.6, 14, 10, 0, 0), (6, 34, 2.4, 10, 6, 0, 0), (5, 36, 3.0, 10, 6, 64, 3)])
def test_emulated_fused_build_window_wide_bands(harness, oracle, order, n_cp, span, n_lm, pairs, R, L):
    """Tracks as long as the window (the steady state of the sliding window: bw = 25 .. 38 control points, 325 .. 741 window tiles): the
    landmark term takes the tiles in passes of 256 (kernels_build.hpp phase 5)."""
    w = synthetic.small_visual(order=order, n_cp=n_cp, n_landmarks=n_lm, obs_pairs=pairs, span=span)
    out = _check(harness, oracle, w, R=R, L=L)
    assert out["bw"] * (out["bw"] + 1) // 2 > 256 and out["n_chunk"] >= 4


def test_emulated_fused_build_constants_and_radius(harness, oracle):
    """Constant control points (zero columns), constant landmarks (no elimination, translation columns intact), another radius."""
    w = synthetic.small_visual(order=4, n_cp=14, n_landmarks=30, obs_pairs=3)
    w.cp_constant = np.r_[np.ones(4, np.uint8), np.zeros(10, np.uint8)]
    w.landmark_constant = (np.arange(30) % 5 == 0).astype(np.uint8)
    _check(harness, oracle, w, radius=37.0)


def test_emulated_fused_build_is_reproducible(harness):
    w = synthetic.small_visual(order=4, n_cp=12, n_landmarks=24, obs_pairs=3)
    a, b = run_emulated_build(harness, w), run_emulated_build(harness, w)
    assert np.array_equal(a["S"], b["S"]) and np.array_equal(a["g"], b["g"]) and np.array_equal(a["Y"], b["Y"]) and a["cost"] == b["cost"]


@pytest.mark.parametrize("order,R,L", [(4, 0, 0), (6, 64, 3)])
def test_emulated_decision_folded_into_build(harness, order, R, L):
    """Iterations after the first: the trust-region decision of the previous iteration is taken by workgroup 0 of k_build_visual, the chunk
    workgroups wait for its flag (Tables::fold_decision). The harness runs it against k_pack_decision(3) + the plain build on the same inputs,
    once with an accepted and once with a rejected step, and exits with code 8 unless solver state, control points and every output of the
    build agree bit for bit."""
    w = synthetic.small_visual(order=order, n_cp=12 if order == 4 else 14, n_landmarks=14, obs_pairs=3)
    run_emulated_build(harness, w, R=R, L=L, fold=True)


@pytest.mark.parametrize("order,bearing,n_cp", [(4, False, 14), (6, False, 16), (4, True, 20), (5, False, 15)])
def test_emulated_full_iteration_matches_oracle(harness, oracle, order, bearing, n_cp):
    """One whole LM iteration of the product's kernel chain on the CPU — k_build_visual -> k_assemble -> k_finalize_reduced -> band Cholesky ->
    sweeps (inverse builders + super-block sweep) -> k_update_visual along the step the sweeps delivered -> k_pack_decision — against the
    oracle's first iteration: cost before and after, gradient max norm, step norm, step quality, new radius, acceptance."""
    w = synthetic.small_visual(order=order, n_cp=n_cp, n_landmarks=40, obs_pairs=3, bearing=bearing)
    out = run_emulated_build(harness, w, full=True)
    u = out["upd"]
    assert u[0] <= 1e-12 and u[1] == 0.0, u[:2]  # k_update_visual against k_backsub_retract + k_cost_visual along the real step
    cost_after, cost_change, gmax, step_norm, rel_dec, radius, valid, ok, st_cost, accepted, done, dcp = u[-12:]
    with ha.Problem(w, lib=oracle) as p:
        s = p.solve(1)
    r = s["iterations"][0]
    assert s["num_iterations"] == 1 and valid == r["step_is_valid"] == 1 and ok == r["step_is_successful"] and accepted == ok and dcp == 0.0
    for got, name in ((cost_after, "cost"), (cost_change, "cost_change"), (gmax, "gradient_max_norm"), (step_norm, "step_norm"), (rel_dec, "relative_decrease"),
                      (radius, "radius")):
        assert abs(got - r[name]) <= 1e-9 * max(abs(r[name]), 1e-300), (name, got, r[name])
    assert abs(st_cost - s["final_cost"]) <= 1e-9 * s["final_cost"]
