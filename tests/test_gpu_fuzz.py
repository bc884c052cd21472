"""Randomised parity sweeps in the driver's suite (`-m gpu`): window shapes the fixed tests do not enumerate, NEW ones whenever the kernels change.

tools/fuzz_parity.py (HIP against the oracle on random windows: spline order, length, band width 4 .. window-wide, IMU, frozen prefixes, constant
landmarks, priors, rotation- / translation-only) and tools/fuzz_shards.py (two landmark shards under torch.distributed against the single-process
solve) run as subprocesses under HS_GUARD=1 (every device table at its exact size with a checked pattern behind it). The seed is a hash of the
kernel sources — the GPU box has no .git, and a round that changes a kernel gets shapes no earlier round has seen. Round 5 found two
out-of-bounds writes this way that four rounds of fixed tests had not; those sweeps were builder-run, these are not."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_seed():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "hyperslam_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return int(h.hexdigest()[:6], 16) + 1


def run(cmd, timeout):
    env = dict(os.environ, HS_GUARD="1", MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return out.returncode, out.stdout, out.stderr


@pytest.mark.gpu
def test_random_windows_against_the_oracle():
    """270 windows of up to 150 control points + 30 of BASELINE size (100 .. 512 control points, 500 .. 5 000 landmarks), bars of
    tests/test_gpu_edge_cases.py::compare; the long-double oracle referees ill-conditioned windows by the rule written in tools/fuzz_parity.py."""
    seed = source_seed()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), str(n), str(seed + i)] + extra, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, cwd=ROOT, env=dict(os.environ, HS_GUARD="1"))
             for i, (n, extra) in enumerate(((135, []), (135, []), (30, ["large"])))]
    for p, n in zip(procs, (135, 135, 30)):
        out, _ = p.communicate(timeout=1500)
        assert p.returncode == 0 and f"{n} cases, 0 failures" in out, f"seed {seed}\n" + out[-3000:]


@pytest.mark.gpu
def test_random_windows_on_two_shards():
    """100 random windows sharded by landmark over two ranks (gloo hook, one GPU) against the single-process solve."""
    seed = source_seed()
    rc, out, err = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "tools", "fuzz_shards.py"), "100", str(seed)], 1500)
    assert rc == 0 and "100 cases on 2 ranks, 0 failures" in out, f"seed {seed}\n" + out[-3000:] + err[-1500:]
