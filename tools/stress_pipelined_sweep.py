"""Stress of the pipelined border sweep (k_border_forward2 next to k_band_factor_mx): N solves of a bordered window in the
default arrangement against ONE solve with the sweep behind the factorisation (HS_DEBUG_FLAGS=128) — every output must have the same bits.
usage (GPU box): python tools/stress_pipelined_sweep.py [repeats=40]"""
import os
import subprocess
import sys

sys.path.insert(0, ".")
import numpy as np


def worker(flags, repeats):
    os.environ["HS_DEBUG_FLAGS"] = flags
    import hyperslam_amd as ha
    from hyperslam_amd import synthetic
    out = []
    windows = [synthetic.config2(), synthetic.small_inertial(order=4, n_cp=72, n_landmarks=300, obs_pairs=3, n_inertial=600, seed=33),
               synthetic.small_inertial(order=6, n_cp=96, n_landmarks=400, obs_pairs=4, n_inertial=900, seed=5)]
    for w in windows:
        p = ha.Problem(w)
        p.snapshot()
        for _ in range(repeats):
            p.restore()
            s = p.solve(5)
            bg, ba = p.bias()
            out.append(np.concatenate([p.control_points().ravel(), p.landmarks().ravel(), bg.ravel(), ba.ravel(), p.gravity().ravel(), [s["final_cost"]]]))
        p.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        res = worker(sys.argv[2], int(sys.argv[3]))
        np.save(sys.argv[4], np.array([np.frombuffer(r.tobytes(), np.uint64).sum(dtype=np.uint64) for r in res], np.uint64))  # checksum of the bits of every solve
        sys.exit(0)
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    sums = {}
    for flags, n in (("128", 1), ("0", repeats)):
        f = "/tmp/stress_%s.npy" % flags
        subprocess.check_call([sys.executable, __file__, "--worker", flags, str(n), f])
        sums[flags] = np.load(f)
    ref, got = sums["128"], sums["0"].reshape(3, repeats)
    bad = int((got != ref[:, None]).sum())
    print("windows 3, solves per window", repeats, " solves that differ from the sequential arrangement:", bad)
    sys.exit(1 if bad else 0)
