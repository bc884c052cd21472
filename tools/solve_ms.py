"""Median solve-stage time (factor + sweeps, device timestamps of hs_solve) for a bench configuration: python tools/solve_ms.py [config] [flags...]
Runs once per HS_DEBUG_FLAGS value given (default: the current environment)."""
import os, subprocess, sys
if len(sys.argv) > 2:
    for f in sys.argv[2:]:
        env = dict(os.environ, HS_DEBUG_FLAGS=f)
        subprocess.run([sys.executable, __file__, sys.argv[1]], env=env)
    sys.exit(0)
sys.path.insert(0, ".")
import numpy as np
os.environ.setdefault("HS_STAGE_TIMING", "1")
import hyperslam_amd as ha
from hyperslam_amd import synthetic
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = {1: synthetic.config1, 2: synthetic.config2, 3: synthetic.config3}[cfg]()
p = ha.Problem(w); p.snapshot()
v = []
for i in range(12):
    p.restore(); s = p.solve(3); v.append(s["solve_ms"])
print("config", cfg, "HS_DEBUG_FLAGS", os.environ.get("HS_DEBUG_FLAGS", "0"), "solve_ms median", float(np.median(v[2:])), "min", min(v), "ms_per_iter", s.get("ms_per_iter"))
