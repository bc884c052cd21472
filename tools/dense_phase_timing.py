"""Phase timestamps of k_dense_factor on a replay-shaped window (33 free block rows, band = window). HS_DEBUG_FLAGS=16.
usage (GPU box): HS_DEBUG_FLAGS=16 python tools/dense_phase_timing.py [n_free=33] [imu=0]"""
import os, sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ.setdefault("HS_DEBUG_FLAGS", "16")
os.environ.setdefault("HS_STAGE_TIMING", "1")
import numpy as np
import hyperslam_amd as ha
from hyperslam_amd import _lib
_lib.PRODUCT_LIB = os.path.join("tools", "libhyperslam_hip_prof.so")  # profiling build (tools/build_profiling_lib.sh): the product library has no timing hooks
os.environ.setdefault("HS_DEBUG_FLAGS", "16")
from test_gpu_edge_cases import window_with_band

n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
imu = len(sys.argv) > 2 and sys.argv[2] != "0"
w = window_with_band(4, n, n_cp=n + 4, imu=imu)
p = ha.Problem(w); p.snapshot()
for i in range(3):
    p.restore(); s = p.solve(1)
lib = _lib.load().cdll
buf = np.zeros(8 * 300 + 8 * 128, np.int64)
lib.hs_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.hs_debug_read(p.h, buf.ctypes.data, len(buf))
t0 = buf[8 * 300 - 1]
t = buf[8 * 300:].reshape(128, 8)
print("band blocks", lib.hs_band_blocks(C.c_void_p(p.h) if not isinstance(p.h, C.c_void_p) else p.h), " solve_ms", s["solve_ms"])
print("units of 10 ns after kernel start;  k: A-barrier B-barrier | row owner solve, stores | diagonal owner update, factor | step")
prev = None
for k in range(n):
    r = t[k] - t0
    step = "" if prev is None else str(r[1] - prev)
    print(f"{k:3d}: {r[0]:6d} {r[1]:6d} | {r[2]:6d} {r[3]:6d} | {r[4]:6d} {r[5]:6d} | {step}")
    prev = r[1]
