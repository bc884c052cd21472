#!/bin/bash
out=gpurun_out; mkdir -p $out
{ for RL in "128 12" "256 24" "192 18" "256 20" "224 22"; do set -- $RL; echo "== R=$1 L=$2"; HS_BUILD_R=$1 HS_BUILD_L=$2 timeout 120 python tools/build_phase_timing.py 1 | tail -22 | grep -v "^ [0-9] \|^1[0-9] " ; HS_BUILD_R=$1 HS_BUILD_L=$2 HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py 1 | head -1; HS_BUILD_R=$1 HS_BUILD_L=$2 HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py 3 | head -1; done
} > $out/r04h_geometry.txt 2>&1
cat $out/r04h_geometry.txt
