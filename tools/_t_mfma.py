import sys, os
sys.path.insert(0, ".")
import numpy as np
import hyperslam_amd as ha
from hyperslam_amd import synthetic, _lib
orc = _lib.Library("oracle/liboracle.so", "hso_")
def rel(a, b): return np.abs(a - b).max() / max(1e-300, np.abs(b).max())
cases = [("small k4 n16", synthetic.small_visual(order=4, n_cp=16, n_landmarks=60, obs_pairs=3)),
         ("two-ended k4 n60", synthetic.small_visual(order=4, n_cp=60, n_landmarks=150, obs_pairs=3, seed=13, span=0.5)),
         ("k6 n20", synthetic.small_visual(order=6, n_cp=20, n_landmarks=50, obs_pairs=3, seed=8)),
         ("medium tracks", synthetic.small_visual(order=4, n_cp=30, n_landmarks=80, obs_pairs=5, seed=15, span=1.6))]
for name, w in cases:
    with ha.Problem(w) as g, ha.Problem(w, lib=orc) as c:
        sg, sc = g.solve(5), c.solve(5)
        print(name, "bw", g.lib.band_blocks(g.h), "iters", sg["num_iterations"], sc["num_iterations"], "ok", sg["num_successful_steps"], sc["num_successful_steps"],
              "cost", sg["final_cost"], sc["final_cost"], "cp rel", rel(g.control_points(), c.control_points()), flush=True)
