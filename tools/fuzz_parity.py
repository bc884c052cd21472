"""Randomised parity sweep, HIP against the oracle (TEST TOOLING; the oracle is the checker): window shapes the fixed tests do not enumerate —
spline order, window length, track span (band width from 4 to window-wide), observation density, bearing / pixel factors, with and without an IMU,
frozen prefixes, constant landmarks, pose priors, rotation- / translation-only windows. Per case: cost 1e-11, residuals and local Jacobians of every residual block (sensor blocks included), trajectory samples with derivatives and stereo
triangulation 1e-9 (`lin`), reduced normal equations 1e-9, 4-iteration trajectory 1e-6 (the bars of
tests/test_gpu_edge_cases.py::compare). Prints one line per case and the failures at the end; exit code = number of failures.
usage (GPU box): python tools/fuzz_parity.py [cases=60] [seed=1] [large]"""
import os
import sys

sys.path.insert(0, ".")
import numpy as np
import hyperslam_amd as ha
from hyperslam_amd import synthetic, _lib


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if a.size else 0.0


def cases(n_cases, seed, large=False):
    """The windows of a sweep (deterministic in the seed). large: windows of BASELINE size (100 .. 512 control points, 500 .. 5 000 landmarks)."""
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        order = int(rng.choice([4, 4, 5, 6]))
        n_cp = int(rng.integers(order + 2, 72)) if rng.random() < 0.8 else int(rng.integers(72, 150))
        imu = bool(rng.random() < 0.4)
        span = float(rng.choice([0.3, 0.6, 1.0, 1.6, 2.4, 0.1 * n_cp]))  # seconds a landmark's observations are spread over (dt = 0.1 s)
        n_lm = int(rng.integers(8, 160))
        pairs = int(rng.integers(2, 9))
        if large:
            n_cp, n_lm, pairs = int(rng.integers(100, 180 if imu else 513)), int(rng.integers(500, 5001)), int(rng.integers(3, 7))  # (IMU: <= 22 bias control points)
            span = float(rng.choice([0.4, 0.8, 1.2, 1.6]))
        bearing = bool(rng.random() < 0.3)
        wseed = int(rng.integers(1, 1 << 20))
        n_ine = (int(rng.integers(40, 500)) if not large else int(rng.integers(1000, 10001))) if imu else 0
        if imu:
            w = synthetic.small_inertial(order=order, n_cp=n_cp, n_landmarks=n_lm, obs_pairs=pairs, n_inertial=n_ine, seed=wseed)
        else:
            w = synthetic.small_visual(order=order, n_cp=n_cp, n_landmarks=n_lm, obs_pairs=pairs, bearing=bearing, seed=wseed, span=span,
                                       with_priors=int(rng.integers(1, 40)) if rng.random() < 0.25 else 0)
        novis = bool(rng.random() < 0.06) and not large and (imu or len(w.prior_stamps) > 0)  # windows of priors / inertial residuals only: bands of k control points
        if novis:
            for name in ("pixel_stamps", "pixels", "pixel_landmark", "pixel_camera", "bearing_stamps", "bearings", "bearing_landmark", "bearing_camera"):
                setattr(w, name, getattr(w, name)[:0])
            w.landmarks = w.landmarks[:0]
        frozen = int(rng.integers(0, max(1, n_cp // 2))) if rng.random() < 0.7 else 0
        w.cp_constant = np.r_[np.ones(max(frozen, 2), np.uint8), np.zeros(n_cp - max(frozen, 2), np.uint8)]
        if rng.random() < 0.15:
            w.rotation_constant = True
        elif rng.random() < 0.15:
            w.translation_constant = True
        if rng.random() < 0.3 and not novis:
            w.landmark_constant = (rng.random(n_lm) < 0.15).astype(np.uint8)
        yield f"case {case:3d}: k {order} n_cp {n_cp:2d} imu {int(imu)} span {span:4.1f} lm {n_lm:3d} pairs {pairs} bearing {int(bearing)} frozen {frozen:2d}" + (" no-visual" if novis else ""), w


def end_points(p, w, iters=4):
    s = p.solve(iters)
    return s, np.concatenate([p.control_points().ravel(), p.landmarks().ravel() if len(w.landmarks) else np.zeros(0)])


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    hip = _lib.load()
    oracle = _lib.Library(os.path.join("oracle", "liboracle.so"), "hso_")
    referee = _lib.Library(os.path.join("oracle", "liboracle_ld.so"), "hsl_", strict=False)  # the oracle in 80-bit long double
    failures = []
    for tag, w in cases(n_cases, seed, large):
        try:
            with ha.Problem(w, lib=hip) as g, ha.Problem(w, lib=oracle) as c:
                bw = g.lib.band_blocks(g.h)
                cg, cc = g.cost(), c.cost()
                lin = 0.0  # residuals and Ceres-local Jacobians of every factor type present, sensor blocks included (hs_linearize), relative per array
                for ftype in range(4):
                    if g.num_residuals(ftype) == 0:
                        continue
                    robust = bool((ftype + len(w.landmarks)) & 1)
                    lg, lc = g.linearize(ftype, robustify=robust, sensor_blocks=True), c.linearize(ftype, robustify=robust, sensor_blocks=True)
                    for name in lc:
                        if lc[name].dtype.kind == "i":
                            lin = max(lin, float(np.any(lg[name] != lc[name])))
                        elif lc[name].size:
                            lin = max(lin, float(np.abs(lg[name] - lc[name]).max() / max(np.abs(lc[name]).max(), 1e-300)))
                # trajectory samples with both derivatives (hs_sample_trajectory) and the front half of process(VisualTracks) (hs_process_tracks) on this window
                lo, hi = w.valid_range()
                srng = np.random.default_rng(len(w.landmarks) + 1000 * w.n_cp)
                st = srng.uniform(lo, hi - 1e-6, 24)
                for a_, b_ in zip(g.sample_trajectory(st, derivatives=True), c.sample_trajectory(st, derivatives=True)):
                    lin = max(lin, float(np.abs(a_ - b_).max() / max(np.abs(b_).max(), 1e-300)))
                if len(w.cam_T_bs) >= 2:
                    px0 = srng.uniform([50, 50], [700, 430], (16, 2))
                    px1 = px0 - np.c_[srng.uniform(2, 40, 16), srng.uniform(-0.5, 0.5, 16)]
                    for a_, b_ in zip(g.process_tracks(float(st[0]), px0, px1), c.process_tracks(float(st[0]), px0, px1)):
                        lin = max(lin, float(np.abs(a_ - b_).max() / max(np.abs(b_).max(), 1e-300)))
                Sg, gg = g.reduced_system(1e4)
                Sc, gc = c.reduced_system(1e4)
                (sg, xg), (sc, xc) = end_points(g, w), end_points(c, w)
                errs = dict(cost=abs(cg - cc) / max(cc, 1e-300), lin=lin, S=rel(Sg, Sc), g=rel(gg, gc), final=abs(sg["final_cost"] - sc["final_cost"]) / abs(sc["final_cost"]),
                            x=rel(xg, xc))
                same = (sg["num_iterations"] == sc["num_iterations"] and sg["termination"] == sc["termination"] and
                        [i["step_is_successful"] for i in sg["iterations"]] == [i["step_is_successful"] for i in sc["iterations"]])
                ok = errs["cost"] < 1e-11 and errs["lin"] < 1e-9 and errs["S"] < 1e-9 and errs["g"] < 1e-9 and same and errs["final"] < 1e-6 and errs["x"] < 1e-6
                note = ""
                if not ok and errs["cost"] < 1e-11 and errs["lin"] < 1e-9 and errs["S"] < 1e-9 and errs["g"] < 1e-9 and same:
                    # Same normal equations, same decisions, end points apart: an ill-conditioned window (control points no residual reaches, held by
                    # the LM damping alone) amplifies the rounding of BOTH sides. The long-double oracle is the referee, by a written rule: the case
                    # passes if the HIP end points are no farther from it than max(3 x the double oracle's own distance, 1e-6) — or if the reduced
                    # system the last iteration solved (damped at the solve's final trust-region radius) has a condition estimate above 1e10, in which
                    # case double precision does not determine the end points to 1e-6 at all and the estimate is printed next to the verdict
                    # (islands of control points connected by no track, free gauge up to the damping: the two double solvers wander apart by as much).
                    with ha.Problem(w, lib=referee) as r:
                        _, xr = end_points(r, w)
                    errs["x_hip_ld"], errs["x_d_ld"] = rel(xg, xr), rel(xc, xr)
                    with ha.Problem(w, lib=oracle) as c2:
                        S_last, _ = c2.reduced_system(float(sc["iterations"][-1]["radius"]))
                    errs["cond"] = float(np.linalg.cond(S_last))
                    ok = errs["x_hip_ld"] <= max(3.0 * errs["x_d_ld"], 1e-6) or (errs["cond"] > 1e10 and errs["x_hip_ld"] <= 10.0 * errs["x_d_ld"] + 1e-7)
                    note = "  (referee)"
                print(tag, f"bw {bw:2d} |", " ".join(f"{k} {v:.1e}" for k, v in errs.items()), note if ok else "  <-- FAIL", flush=True)
                if not ok:
                    failures.append((tag, errs))
        except Exception as e:  # refused windows (limits of DESIGN 8) are not parity failures; anything else is
            msg = str(e)
            refused = "INVALID" in msg or "invalid" in msg or "too long" in msg or "span too many" in msg
            print(tag, "| refused:" if refused else "| ERROR:", msg[:160], flush=True)
            if not refused:
                failures.append((tag, msg))
    print(f"{n_cases} cases, {len(failures)} failures")
    for f in failures:
        print("FAILED", f)
    sys.exit(min(len(failures), 100))


if __name__ == "__main__":
    main()
