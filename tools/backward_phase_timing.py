"""Phase timestamps of k_band_backward_w (HS_DEBUG_FLAGS=16): python tools/backward_phase_timing.py [config]"""
import os, sys, ctypes as C; sys.path.insert(0, ".")
os.environ["HS_DEBUG_FLAGS"] = os.environ.get("HS_DEBUG_FLAGS", "16")
import numpy as np
os.environ.setdefault("HS_STAGE_TIMING", "1")
import hyperslam_amd as ha
from hyperslam_amd import synthetic, _lib
_lib.PRODUCT_LIB = os.path.join("tools", "libhyperslam_hip_prof.so")  # profiling build (tools/build_profiling_lib.sh): the product library has no timing hooks
os.environ.setdefault("HS_DEBUG_FLAGS", "16")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = {1: synthetic.config1, 2: synthetic.config2, 3: synthetic.config3}[cfg]()
p = ha.Problem(w); p.snapshot()
for i in range(2): p.restore(); s = p.solve(1)
lib = _lib.load().cdll
n = 64 * 1024 + 4096
buf = np.zeros(n, np.int64)
lib.hs_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.hs_debug_read(p.h, buf.ctypes.data, n)
for b in range(2):
    t = buf[64 * 1024 + 2048 * b:64 * 1024 + 2048 * (b + 1)]
    if t[0] == 0: continue
    steps = t[16:16 + 1100]; steps = steps[steps > 0]
    d = -np.diff(np.sort(steps)[::-1]) if len(steps) > 1 else np.zeros(1)
    print(f"block {b}: init {t[1]-t[0]} flag-wait {t[2]-t[1]} loop {t[3]-t[2]} fence+join {t[4]-t[3]} outputs {t[5]-t[4] if t[5] else 0}  (units of 10 ns); "
          f"{len(steps)} steps, median step {np.median(d)}, mean {d.mean():.1f}, max {d.max()}")
print("solve_ms", s["solve_ms"])
