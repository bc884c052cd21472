"""Decision folded into k_build_visual (Tables::fold_decision): when the decision workgroup publishes and how long the chunk workgroups wait
(HS_DEBUG_FLAGS=32, profiling build: tools/build_profiling_lib.sh). usage (GPU box): python tools/fold_phase_timing.py [config=1]"""
import os
import sys, ctypes as C; sys.path.insert(0, ".")
import numpy as np
os.environ["HS_DEBUG_FLAGS"] = str(32 | int(os.environ.get("HS_DEBUG_FLAGS", "0")))
import hyperslam_amd as ha
from hyperslam_amd import synthetic, _lib
_lib.PRODUCT_LIB = os.path.join("tools", "libhyperslam_hip_prof.so")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = {1: synthetic.config1, 2: synthetic.config2, 3: synthetic.config3}[cfg]()
p = ha.Problem(w); p.snapshot()
for i in range(3): p.restore(); s = p.solve(2)  # the second iteration's build carries the first iteration's decision
lib = _lib.load().cdll
n = 48 * 1024 + 64 * 1024
buf = np.zeros(n, np.int64)
lib.hs_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.hs_debug_read(p.h, buf.ctypes.data, n)
t = buf[48 * 1024:].reshape(1024, 4, 16)
d = t[1023, 0, :5].copy()
t = t[:1023]
t = t[t[:, 0, 0] > 0]
t0 = min(t[:, :, 0].min(), d[0] if d[0] > 0 else t[:, :, 0].min())
print("decision workgroup [us after the first stamp of the launch]: entry, partials loaded, summed, decided, published:", ((d - t0) * 0.01).round(2).tolist())
start = (t[:, :, 0].min(axis=1) - t0) * 0.01
inputs = (t[:, :, 1].max(axis=1) - t0) * 0.01  # inputs + keys, behind the first barrier
ready = (t[:, :, 2].max(axis=1) - t0) * 0.01   # slots sorted, decision in hand, point in LDS (fold mode: the wait sits in front of this stamp)
print("chunk workgroups:", len(t), " start quantiles [us]", np.percentile(start, [0, 25, 50, 75, 100]).round(2), " inputs in LDS", np.percentile(inputs, [0, 25, 50, 75, 100]).round(2),
      " sorted + decision in hand", np.percentile(ready, [0, 25, 50, 75, 100]).round(2), " of which after the inputs", np.percentile(ready - inputs, [0, 25, 50, 75, 100]).round(2))
end = (t[:, :, 12].max(axis=1) - t0) * 0.01
print("end quantiles", np.percentile(end, [0, 25, 50, 75, 100]).round(2))
