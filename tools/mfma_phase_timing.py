"""Phase timestamps of k_band_factor_mfma (HS_DEBUG_FLAGS=16, 100 MHz clock): python tools/mfma_phase_timing.py [config]"""
import os, sys, ctypes as C; sys.path.insert(0, ".")
os.environ["HS_DEBUG_FLAGS"] = str(16 | int(os.environ.get("HS_DEBUG_FLAGS", "0")))
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
n = 8 * 1024 + 8 * 600
buf = np.zeros(n, np.int64)
lib.hs_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.hs_debug_read(p.h, buf.ctypes.data, n)
t = buf[:8 * 600].reshape(-1, 8); q = buf[8 * 1024:].reshape(-1, 8)
r = slice(6, 50, 2)
print("units of 10 ns. step:", np.median(np.diff(t[5:50, 0])), " tile update", np.median(t[r, 1] - t[r, 0]))
print("loader (even steps), relative to the start of phase B of the same step: put start", np.median(t[r, 4] - t[r, 0]), " put done",
      np.median(t[r, 5] - t[r, 0]), " fetch issued", np.median(t[r, 6] - t[r, 0]))
r = slice(6, 50)
print("panel row r, relative to the start of phase B of step r - 1: read start", np.median(q[r, 0] - t[5:49, 0]), " read done",
      np.median(q[r, 4] - t[5:49, 0]), " phase B start", np.median(q[r, 1] - t[5:49, 0]), " factor done", np.median(q[r, 2] - t[5:49, 0]),
      " row written", np.median(q[r, 3] - t[5:49, 0]))
print("solve_ms", s["solve_ms"])
