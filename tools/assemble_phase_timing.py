"""Phase timestamps of k_assemble (HS_DEBUG_FLAGS=256, profiling build: tools/build_profiling_lib.sh).
usage (GPU box): python tools/assemble_phase_timing.py [config=1|2|3|r]   (r: the replay's steady state, window-wide band: k_assemble_wide)"""
import os
import sys, ctypes as C; sys.path.insert(0, ".")
import numpy as np
os.environ["HS_DEBUG_FLAGS"] = str(256 | int(os.environ.get("HS_DEBUG_FLAGS", "0")))
import hyperslam_amd as ha
from hyperslam_amd import synthetic, _lib
_lib.PRODUCT_LIB = os.path.join("tools", "libhyperslam_hip_prof.so")
cfg = sys.argv[1] if len(sys.argv) > 1 else "1"
w = synthetic.small_visual(order=4, n_cp=36, n_landmarks=480, obs_pairs=18, span=3.0) if cfg == "r" else \
    {1: synthetic.config1, 2: synthetic.config2, 3: synthetic.config3}[int(cfg)]()
p = ha.Problem(w); p.snapshot()
for i in range(3): p.restore(); s = p.solve(1)
lib = _lib.load().cdll
n_wg = 6 * w.n_cp
n = 128 * 1024 + 8 * n_wg
buf = np.zeros(n, np.int64)
lib.hs_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.hs_debug_read(p.h, buf.ctypes.data, n)
t = buf[128 * 1024:].reshape(n_wg, 8)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
rel = (t[:, :7] - t0) * 0.01
names = ["start", "state word", "work-list ranges", "work-list entries", "partial values", "combined (barrier)", "row written"]
print("workgroups stamped:", len(t), " kernel span first start -> last end [us]:", rel[:, 6].max())
print("start quantiles [us]", np.percentile(rel[:, 0], [0, 25, 50, 75, 100]).round(2), " end quantiles", np.percentile(rel[:, 6], [0, 25, 50, 75, 100]).round(2))
for i in range(1, 7):
    d = rel[:, i] - rel[:, i - 1]
    print(f"{names[i]:20s} after the previous stamp [us]: median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f}")
print("workgroup duration [us]: median", np.median(rel[:, 6] - rel[:, 0]).round(2), " max", (rel[:, 6] - rel[:, 0]).max().round(2))
