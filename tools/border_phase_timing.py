"""Phase timestamps of k_border_solve_reg (HS_DEBUG_FLAGS=16, profiling build: tools/build_profiling_lib.sh) on configs[2].
Per pair of columns (8 slots at xpart[8 (700 + pair)]): [0] start, [1] columns published, [2] barrier passed, [3] pivots + operands ready,
[4] trailing update done, [5] scaled columns stored."""
import os, sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
import hyperslam_amd as ha
from hyperslam_amd import synthetic, _lib
_lib.PRODUCT_LIB = os.path.join("tools", "libhyperslam_hip_prof.so")
os.environ.setdefault("HS_DEBUG_FLAGS", "16")
w = synthetic.config2()
p = ha.Problem(w); p.snapshot()
for i in range(2): p.restore(); s = p.solve(1)
lib = _lib.load().cdll
buf = np.zeros(8 * 1024, np.int64)
lib.hs_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.hs_debug_read(p.h, buf.ctypes.data, len(buf))
t = buf[8 * 700:8 * 700 + 8 * 49].reshape(49, 8)
r = slice(5, 45)
print("units of 10 ns, medians over pairs 5..44: pair", np.median(np.diff(t[r, 0])), " publish", np.median(t[r, 1] - t[r, 0]), " barrier", np.median(t[r, 2] - t[r, 1]),
      " pivots + operands", np.median(t[r, 3] - t[r, 2]), " update", np.median(t[r, 4] - t[r, 3]), " store columns", np.median(t[r, 5] - t[r, 4]))
for k in (5, 20, 40): print(k, [int(v - t[k, 0]) for v in t[k, :6]], "next", int(t[k + 1, 0] - t[k, 0]))
