#!/bin/bash
# measurement batch: SIMD placement of the build kernel's waves, uncontended times of the inertial-branch kernels (one stream), the
# free-running estimation dump's HIP-vs-oracle difference
out=gpurun_out; mkdir -p $out
{ echo "== build phases + SIMD placement"; timeout 120 python tools/build_phase_timing.py 1
  echo "== estimation dump difference (0.6 s free-running replay, stereo-inertial, order 4)"
  cd hyperslam_amd/host && ./replay 0.6 1 4 /tmp/a.hyper > /dev/null && ./replay_oracle 0.6 1 4 /tmp/b.hyper > /dev/null; cd ../..
  python - <<'PY'
import numpy as np
a, b = np.loadtxt("/tmp/a.hyper", delimiter=","), np.loadtxt("/tmp/b.hyper", delimiter=",")
print("rows", a.shape, "max abs diff cols 5+:", np.abs(a[:, 5:] - b[:, 5:]).max(), " cols 1-4:", np.abs(a[:, 1:5] - b[:, 1:5]).max(), " per column", np.abs(a - b).max(axis=0))
PY
} > $out/r04h_misc.txt 2>&1
HS_DEBUG_FLAGS=1048576 timeout 300 bash tools/kernel_stats.sh $out/r04h_config2_one_stream.csv python tools/time_config.py 2 > $out/r04h_config2_one_stream.txt 2>&1
cat $out/r04h_misc.txt $out/r04h_config2_one_stream.txt
