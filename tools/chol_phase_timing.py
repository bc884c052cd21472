import os
import sys, ctypes as C; sys.path.insert(0,".")
import numpy as np
os.environ.setdefault("HS_STAGE_TIMING", "1")
import hyperslam_amd as ha
from hyperslam_amd import synthetic, _lib
_lib.PRODUCT_LIB = os.path.join("tools", "libhyperslam_hip_prof.so")  # profiling build (tools/build_profiling_lib.sh): the product library has no timing hooks
os.environ.setdefault("HS_DEBUG_FLAGS", "16")
w=synthetic.config1()
p=ha.Problem(w); p.snapshot()
for i in range(2): p.restore(); s=p.solve(1)
lib=_lib.load().cdll
buf=np.zeros(16*1024+8*256, np.int64)
lib.hs_debug_read.argtypes=[C.c_void_p, C.c_void_p, C.c_int]
lib.hs_debug_read(p.h, buf.ctypes.data, len(buf))
t=buf[:8*128].reshape(128,8); t2=buf[8*1024:8*1024+8*128].reshape(128,8); t3=buf[16*1024:16*1024+8*128].reshape(128,8)
# compute wave 0: [0] step start, [1] rank-6 update + publish done; panel wave (row r): [2] start, [3] row updated, [4] factored, [5] X written
r = slice(10, 110)
print("units of 10 ns. step:", np.median(np.diff(t[r, 0])), " compute P2+publish:", np.median(t[r, 1] - t[r, 0]))
print("panel: update", np.median(t[r, 3] - t[r, 2]), " factor", np.median(t[r, 4] - t[r, 3]), " solve+write", np.median(t[r, 5] - t[r, 4]),
      " panel start after step start (row r vs step r-1):", np.median(t[11:111, 2] - t[10:110, 0]))
ev = slice(10, 110, 2)
print("arrival at the step barrier relative to step start: compute", np.median(t[ev, 1] - t[ev, 0]), " panel(row r+1)", np.median(t[11:111:2, 5] - t[ev, 0]),
      " storer", np.median(t[ev, 6] - t[ev, 0]), " loader", np.median(t[ev, 7] - t[ev, 0]))
print("after the barrier: next step start - compute arrival", np.median(t[11:111, 0] - t[10:110, 1]), " next step start - panel(r+1) end", np.median(t[11:111, 0] - t[11:111, 5]),
      " next step start - storer arrival", np.median(t[11:111, 0] - t[10:110, 6]), " - loader arrival", np.median(t[12:110:2, 0] - t[11:109:2, 7]),
      " panel(r+1) start - step r start", np.median(t[11:111, 2] - t[10:110, 0]))
print("raw, steps 20..31, relative to the start of step 20 [10 ns]: columns = step start, compute done, panel(r) start, row updated, factored, X written, storer arrival, loader arrival (even steps)")
tw = buf[8 * 512:8 * 512 + 8 * 128].reshape(128, 8)
for r in range(20, 32):
    print(r, [int(v - t[20, 0]) if v else None for v in t[r]], " compute waves 1, 2 (start, done):", [int(v - t[20, 0]) for v in tw[r, 2:6]])
print("solve_ms", s["solve_ms"])

for job in (0, 1):
    c = buf[8 * (200 + 10 * job): 8 * (200 + 10 * job) + 9]
    b = buf[8 * 200]
    print("job", job, "relative to job 0's start [10 ns]: tiles requested", c[0] - b, " loaded", c[1] - b, " prologue done", c[2] - b,
          " junction reached", c[3] - b, " partner arrived", c[4] - b, " merged", c[5] - b, " X_m published", c[6] - b, " last row", c[7] - b,
          " window handed over", c[8] - b)

for blk in (0, 1):
    c = buf[8 * (230 + 10 * blk): 8 * (230 + 10 * blk) + 8]
    b = buf[8 * 230]
    print("backward block", blk, "[10 ns after block 0's start]: started", c[0] - b, " operands staged", c[1] - b, " middle arrived", c[2] - b, " sweep starts", c[3] - b,
          " middle solved", c[4] - b, " published", c[5] - b, " sweep done", c[6] - b, " outputs", c[7] - b)

c = buf[8 * 260:8 * 260 + 4]; b = buf[8 * 230]
print("builder of the last super-block of job 0 [10 ns after sweep block 0's start]: start", c[0] - b, " block in LDS", c[1] - b, " inverse done", c[2] - b, " flag raised", c[3] - b)
