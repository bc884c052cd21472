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


@pytest.mark.gpu
def test_replay_hip_matches_oracle(built):
    a, b = run("replay", 2.0, 0, 4), run("replay_oracle", 2.0, 0, 4)
    assert a["optimizations"] == b["optimizations"] and a["landmarks"] == b["landmarks"] and a["control_points"] == b["control_points"]
    assert abs(a["position_rmse_m"] - b["position_rmse_m"]) < 1e-5
    assert abs(a["last_cost"][1] - b["last_cost"][1]) <= 1e-5 * b["last_cost"][1]


@pytest.mark.gpu
def test_replay_hip_stereo_inertial(built):
    a, b = run("replay", 1.5, 1, 4), run("replay_oracle", 1.5, 1, 4)
    assert a["optimizations"] == b["optimizations"]
    assert abs(a["position_rmse_m"] - b["position_rmse_m"]) < 1e-4
