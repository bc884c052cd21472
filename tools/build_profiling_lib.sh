#!/bin/bash
# Profiling build of the HIP library: the same sources with the phase-timestamp hooks compiled in (HS_PROFILE_HOOKS=1) and
# hs_debug_read exported. The product library (hyperslam_amd/libhyperslam_hip.so, __graft_entry__.build()) contains neither.
# Used by tools/{chol,dense,mfma,backward}_phase_timing.py through HS_LIBRARY=<path>.
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DHS_PROFILE_HOOKS=1 -Wl,-soname,libhyperslam_hip_prof.so \
  -o tools/libhyperslam_hip_prof.so hyperslam_amd/csrc/capi.hip
echo tools/libhyperslam_hip_prof.so
