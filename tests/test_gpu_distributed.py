"""Residual-sharded HIP path: two processes (sharing the one GPU of the test box, gloo with host staging in the hook) must
reproduce the single-process HIP solve. On a multi-GPU node the same hook runs RCCL on device tensors (bench.py --gpus N)."""
import numpy as np
import pytest

import hyperslam_amd as ha
from hyperslam_amd import synthetic
from test_distributed_cpu import run_workers
from util import rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which", ["hip_visual", "hip_inertial"])
def test_sharded_hip_matches_single_process(which, tmp_path, hip):
    full = synthetic.small_inertial(order=4, n_cp=18, n_landmarks=40) if which.endswith("inertial") else \
        synthetic.small_visual(order=4, n_cp=18, n_landmarks=64, obs_pairs=3, with_priors=21)
    with ha.Problem(full, lib=hip) as p:
        S, g = p.reduced_system(1e4)
        s = p.solve(5)
        cp, lm = p.control_points(), p.landmarks()
    ranks = run_workers(which, tmp_path)
    for r in ranks:
        assert rel(r["S"], S) < 1e-10 and rel(r["g"], g) < 1e-10, (rel(r["S"], S), rel(r["g"], g))
        assert int(r["iters"]) == s["num_iterations"]
        assert np.allclose(r["costs"], [it["cost"] for it in s["iterations"]], rtol=1e-7, atol=0)
        assert rel(r["cp"], cp) < 1e-7
        ids = r["lm_ids"]
        assert rel(r["lm"][ids], lm[ids]) < 1e-7
    assert np.array_equal(ranks[0]["S"], ranks[1]["S"]) and np.array_equal(ranks[0]["cp"], ranks[1]["cp"])


def test_rccl_hook_single_rank(tmp_path):
    """backend "nccl" (= RCCL) on the library's device buffer and stream; a one-rank sum must leave the solve bit-identical."""
    import os
    import subprocess
    import sys
    out = str(tmp_path / "rccl.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_rccl_worker.py"), out], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    assert np.array_equal(d["c0"], d["c1"]) and np.array_equal(d["cp0"], d["cp1"])
    assert np.array_equal(d["c0"], d["c2"]) and np.array_equal(d["cp0"], d["cp2"])   # hs_rccl_init path (no Python hook)
