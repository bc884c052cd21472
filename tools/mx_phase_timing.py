"""Phase timestamps of k_band_factor_mx (profiling build, HS_DEBUG_FLAGS = 16, 100 MHz clock): python tools/mx_phase_timing.py [config]"""
import os, sys, ctypes as C; sys.path.insert(0, ".")
os.environ["HS_DEBUG_FLAGS"] = str(16 | int(os.environ.get("HS_DEBUG_FLAGS", "0")))
import numpy as np
os.environ.setdefault("HS_STAGE_TIMING", "1")
import hyperslam_amd as ha
from hyperslam_amd import synthetic, _lib
_lib.PRODUCT_LIB = os.path.join("tools", "libhyperslam_hip_prof.so")  # profiling build (tools/build_profiling_lib.sh): the product library has no timing hooks
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
r = slice(8, 56)
t0 = t[r, 0]
med = lambda a: float(np.median(a))
print("units of 10 ns, medians over block rows 8 .. 55 of job 0, relative to the start of the iteration in MFMA wave 0")
print("step:", med(np.diff(t[8:57, 0])))
print("MFMA waves done: ", med(t[r, 2] - t0), med(t[r, 3] - t0), med(t[r, 4] - t0), " loader done:", med(t[r, 5] - t0), " storer done:", med(t[r, 1] - t0))
print("panel wave 3 (SIMD 3, next to the loader): start", med(q[r, 0] - t0), " own column updated", med(q[r, 1] - t0), " diagonal block factored", med(q[r, 2] - t0), " solved + published", med(q[r, 3] - t0))
print("panel wave 2 (SIMD 2, next to the storer): start", med(q[r, 4] - t0), " own column updated", med(q[r, 5] - t0), " diagonal block factored", med(q[r, 6] - t0), " solved + published", med(q[r, 7] - t0))
print("raw, block rows 20 .. 27 (MFMA group 0 start, groups 0 1 2 done, loader, storer | wave 3: start upd chol end | wave 2: start upd chol end):")
for i in range(20, 28):
    b = t[i, 0]
    print(i, [int(x - b) for x in t[i, [0, 2, 3, 4, 5, 1]]], [int(x - b) for x in q[i, :4]], [int(x - b) for x in q[i, 4:8]])
names = ["prologue done", "junction reached", "partner arrived", "merged", "X_m published", "last block row", "window handed over"]
for job in (0, 1):
    c = t[590 + 4 * job]
    print(f"job {job} relative to job 0's start [10 ns]:", " ".join(f"{n} {int(c[k + 1] - t[590, 0])}" for k, n in enumerate(names) if c[k + 1] > 0), " (start", int(c[0] - t[590, 0]), ")")
print("solve_ms", s["solve_ms"])
