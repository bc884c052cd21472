#!/bin/bash
# usage (GPU box): bash tools/r06_build_streams.sh <tag> — k_build_visual with the streams of its J'J phase dealt per chunk: GPU suite, configs[1..3],
# phase stamps of the build (profiling build) A = per chunk / B = fixed per diagonal (HS_DEBUG_FLAGS sign bit), alternating timings of configs[1]
tag=${1:-r06st}
out=gpurun_out; mkdir -p $out
(time python -m pytest tests -x -q -m gpu -n 2) > $out/${tag}_tests.log 2>&1
tail -4 $out/${tag}_tests.log
for c in 1 2 3; do HS_STAGE_TIMING=0 python tools/time_config.py $c | head -1; done > $out/${tag}_configs.txt 2>&1
cat $out/${tag}_configs.txt
if [ -f tools/libhyperslam_hip_prof.so ]; then
  for c in 1 2 3; do
    echo "== config $c, streams per chunk"; python tools/build_phase_timing.py $c | head -34
    echo "== config $c, fixed streams per diagonal"; HS_DEBUG_FLAGS=-2147483648 python tools/build_phase_timing.py $c | head -34
  done > $out/${tag}_build_phase_timing.txt 2>&1
  grep -E "^==|J'J tiles|cost summed|kernel span|duration quantiles" $out/${tag}_build_phase_timing.txt
  for c in 1 2 3; do for f in 0 -2147483648 0 -2147483648; do echo -n "flags $f: "; HS_STAGE_TIMING=0 HS_LIBRARY=tools/libhyperslam_hip_prof.so HS_DEBUG_FLAGS=$f python tools/time_config.py $c | head -1; done; done > $out/${tag}_ab_configs.txt 2>&1
  cat $out/${tag}_ab_configs.txt
fi
