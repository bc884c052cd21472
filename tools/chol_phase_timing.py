import sys, ctypes as C; sys.path.insert(0,".")
import numpy as np
import hyperslam_amd as ha
from hyperslam_amd import synthetic, _lib
w=synthetic.config1()
p=ha.Problem(w); p.snapshot()
for i in range(2): p.restore(); s=p.solve(1)
lib=_lib.load().cdll
buf=np.zeros(8*128, np.int64)
lib.hs_debug_read.argtypes=[C.c_void_p, C.c_void_p, C.c_int]
lib.hs_debug_read(p.h, buf.ctypes.data, 8*128)
t=buf.reshape(128,8)
d=np.diff(t[:, :6], axis=1)
print("wall_clock units; median per phase [P1, barA, P2, P3, barB]:", np.median(d[10:110],axis=0), "step:", np.median(np.diff(t[10:110,0])))
print("solve_ms", s["solve_ms"])
