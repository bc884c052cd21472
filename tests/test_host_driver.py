"""C++ host mirror of AbstractOptimizer (hyperslam_amd/host/optimizer.hpp): window logic on CPU through the oracle, and the
same binary driven through the HIP library on the GPU."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "hyperslam_amd", "host")


def run(binary, *args):
    out = subprocess.run([os.path.join(HOST, binary), *map(str, args)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()


def test_sliding_window_logic_cpu(built):
    """submit()/setWindow semantics of abstract.cpp:74-147 with separation 0.1 s, max_window 3.0 s."""
    r = run("replay_oracle", 3.6, 0, 4)
    # one optimize() per separation of data once the state is exhausted: stamps in (0.1, 3.6) -> 35 extensions
    assert r["optimizations"] == 35
    lo, hi = r["window"]
    assert abs((hi - lo) - 3.0) < 1e-9 and 3.6 - 1e-9 <= hi <= 3.7 + 1e-9   # grew to max_window, then slid
    assert r["control_points"] <= 36 + 4 and r["landmarks"] > 100
    assert r["position_rmse_m"] < 0.5 and r["last_cost"][1] <= r["last_cost"][0]


def test_stereo_inertial_replay_cpu(built):
    r = run("replay_oracle", 1.2, 1, 4)
    assert r["imu"] == 1 and r["optimizations"] == 11 and r["last_cost"][1] <= r["last_cost"][0]


def run_traced(binary, *args):
    """(summary JSON, per-optimize() trace rows) — HS_REPLAY_TRACE prints one row per optimize() on stderr."""
    out = subprocess.run([os.path.join(HOST, binary), *map(str, args)], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HS_REPLAY_TRACE="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    rows = []
    for line in out.stderr.splitlines():
        f = line.split()
        if f and f[0] == "opt":
            rows.append(dict(cps=int(f[3]), lms=int(f[5]), blocks=int(f[7]), iters=int(f[9]), initial=float(f[15]), final=float(f[17])))
    return json.loads(out.stdout.strip().splitlines()[-1]), rows


def check_replay_pair(a, ra, b, rb, rmse_bound):
    # which residuals / control points / landmarks exist at every optimize() is host logic: identical
    assert a["optimizations"] == b["optimizations"] and a["landmarks"] == b["landmarks"] and a["control_points"] == b["control_points"]
    assert [(r["cps"], r["lms"], r["blocks"]) for r in ra] == [(r["cps"], r["lms"], r["blocks"]) for r in rb]
    # the first optimize() starts from identical inputs: same initial cost to round-off, same result to the trajectory tolerance.
    # (Windows whose newest control points are not yet observed are rank deficient up to the LM damping, condition number
    # ~1e10: later optimize() calls start from states that differ at 1e-6 and the accept/reject sequences decouple, as
    # they would between two CPU builds of the reference — so beyond the first call only the quality is compared.)
    assert abs(ra[0]["initial"] - rb[0]["initial"]) <= 1e-10 * rb[0]["initial"]
    assert abs(ra[0]["final"] - rb[0]["final"]) <= 1e-4 * rb[0]["final"]
    for r in ra:
        assert r["final"] <= r["initial"] * (1 + 1e-12)
    assert a["position_rmse_m"] < rmse_bound and b["position_rmse_m"] < rmse_bound
    assert a["last_cost"][1] < 3.0 * b["last_cost"][1] + 1e-3 and a["position_rmse_m"] < 3.0 * b["position_rmse_m"] + 1e-2


def test_estimation_dump_cpu(built, tmp_path):
    """The SIGUSR1 dump (apps/hyperslam/main.cpp:52-80): 100 Hz samples over the state range, `stamp, q(xyzw), p`, 20 digits."""
    import numpy as np
    out = tmp_path / "estimation.hyper"
    r = run("replay_oracle", 1.2, 0, 4, out)
    rows = np.loadtxt(out, delimiter=",")
    assert rows.shape[1] == 8 and rows.shape[0] > 100
    assert np.allclose(np.diff(rows[:, 0]), 0.01, atol=1e-9)                      # 100 Hz
    assert np.allclose(np.linalg.norm(rows[:, 1:5], axis=1), 1.0, atol=1e-12)     # unit quaternions
    assert "e+" in out.read_text().splitlines()[0] or "e-" in out.read_text().splitlines()[0]
    assert r["optimizations"] == 11


@pytest.mark.gpu
def test_estimation_dump_hip_matches_oracle(built, tmp_path):
    import numpy as np
    a, b = tmp_path / "a.hyper", tmp_path / "b.hyper"
    run("replay", 0.6, 1, 4, a), run("replay_oracle", 0.6, 1, 4, b)
    ra, rb = np.loadtxt(a, delimiter=","), np.loadtxt(b, delimiter=",")
    assert ra.shape == rb.shape and np.array_equal(ra[:, 0], rb[:, 0])
    assert np.abs(ra[:, 5:] - rb[:, 5:]).max() < 1e-3   # same trajectory up to the solver-path differences discussed below


@pytest.mark.gpu
def test_replay_hip_matches_oracle(built):
    """Stereo-only replay long enough for feature tracks to span more than 22 control points (wide-band factorisation)."""
    (a, ra), (b, rb) = run_traced("replay", 2.6, 0, 4), run_traced("replay_oracle", 2.6, 0, 4)
    check_replay_pair(a, ra, b, rb, rmse_bound=0.5)


@pytest.mark.gpu
def test_replay_hip_stereo_inertial(built):
    (a, ra), (b, rb) = run_traced("replay", 1.5, 1, 4), run_traced("replay_oracle", 1.5, 1, 4)
    check_replay_pair(a, ra, b, rb, rmse_bound=0.5)
