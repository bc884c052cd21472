#!/bin/bash
out=gpurun_out; mkdir -p $out
{ echo "== default"; timeout 120 python tools/build_phase_timing.py 1
} > $out/r04d_build_phases.txt 2>&1
cat $out/r04d_build_phases.txt
