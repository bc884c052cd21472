#!/bin/bash
out=gpurun_out; mkdir -p $out
{
echo "== default"; timeout 120 python tools/build_phase_timing.py 1
echo "== R=128 L=12"; HS_BUILD_R=128 HS_BUILD_L=12 timeout 120 python tools/build_phase_timing.py 1
echo "== config 2"; timeout 120 python tools/build_phase_timing.py 2
} > $out/r04b_build_phases.txt 2>&1
cat $out/r04b_build_phases.txt
