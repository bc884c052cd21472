"""Phase timestamps of k_dense_solve_mx (kernels_dense_mx.hpp) on a replay-shaped window: n free block rows with window-wide bands, with or
without an IMU (border unknowns). Profiling build (tools/build_profiling_lib.sh), HS_DEBUG_FLAGS=16; wall_clock64 ticks are 10 ns.
usage (GPU box): python tools/dense_mx_phase_timing.py [n_free=33] [imu=0]"""
import os, sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["HS_DEBUG_FLAGS"] = str(16 | int(os.environ.get("HS_DEBUG_FLAGS", "0")))
os.environ.setdefault("HS_STAGE_TIMING", "1")
import numpy as np
import hyperslam_amd as ha
from hyperslam_amd import _lib
_lib.PRODUCT_LIB = os.environ.get("HS_PROF_LIB", os.path.join("tools", "libhyperslam_hip_prof.so"))
from test_gpu_edge_cases import window_with_band

n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
imu = len(sys.argv) > 2 and sys.argv[2] != "0"
w = window_with_band(4, n, n_cp=n + 4, imu=imu)
p = ha.Problem(w); p.snapshot()
for i in range(3):
    p.restore(); s = p.solve(1)
lib = _lib.load().cdll
buf = np.zeros(8 * 300 + 8 * 32, np.int64)
lib.hs_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.hs_debug_read(p.h, buf.ctypes.data, len(buf))
t0 = buf[8 * 300 - 1]
t = buf[8 * 300:].reshape(32, 8) - t0
print("band blocks", lib.hs_band_blocks(p.h), " border unknowns", (6 * len(w.imu["bias_g"]) + 2) if imu else 0, " solve_ms", s["solve_ms"])
print("[10 ns after kernel start]  tiles loaded %d" % t[20][3])
print(" k:  U1 (row k+1) + barrier  panel k+1 (load, pivots, store)  own U2  barrier | step      [panel wave 0]")
for k in range(16):
    r, q = t[k], t[k + 1]
    if r[3] <= 0:
        break
    print(f"{k:3d}: {q[0] - r[2]:8d} {q[1] - q[0]:10d} ({q[4] - q[0]:4d} {q[5] - q[4]:4d} {q[1] - q[5]:4d}) {r[3] - q[1]:8d} {(q[2] if q[3] > 0 else t[20][0]) - r[3]:8d} | {(q[2] if q[3] > 0 else t[20][0]) - r[2]:6d}")
print("factorisation done %d, sweep done %d (+%d), outputs done %d (+%d)" % (t[20][0], t[20][1], t[20][1] - t[20][0], t[20][2], t[20][2] - t[20][1]))
print("sweep blocks from the last: (fetch + pend -> LDS + barrier, W pend, apply U(:, K) x_K) | block")
for i in range(8):
    r = t[21 + i]
    if r[3] <= 0 or r[0] <= 0:
        break
    print(f"{i:3d}: {r[1] - r[0]:6d} {r[2] - r[1]:6d} {r[3] - r[2]:6d} | {r[3] - r[0]:6d}")
