import os
"""Times optimize() on a BASELINE.json config (device stage times from hs_summary). usage: python tools/time_config.py 1|2|3"""
import sys, time
sys.path.insert(0, ".")
os.environ.setdefault("HS_STAGE_TIMING", "1")
import hyperslam_amd as ha
from hyperslam_amd import synthetic
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = {0: synthetic.config0, 1: synthetic.config1, 2: synthetic.config2, 3: synthetic.config3}[cfg]()
p = ha.Problem(w)
p.snapshot()
for _ in range(3):
    p.restore()
    s = p.solve(5)
t = time.perf_counter()
for _ in range(5):
    p.restore()
    s = p.solve(5)
dt = (time.perf_counter() - t) / 5
print(f"config {cfg}: {w.num_residual_blocks()} residual blocks, n_cp {w.n_cp}; {1e3*dt/5:.3f} ms/LM iteration wall; "
      f"cost {s['initial_cost']:.6g} -> {s['final_cost']:.6g} in {s['num_iterations']} it ({s['num_successful_steps']} ok)")
print({k: round(s[k] / 5, 4) for k in ("linearize_ms", "schur_ms", "solve_ms", "update_ms", "total_ms")})
