#!/bin/bash
out=gpurun_out; mkdir -p $out
{
echo "== default"; timeout 120 python tools/build_phase_timing.py 1
echo "== config 2"; timeout 120 python tools/build_phase_timing.py 2
for c in 1 2 3; do echo "== config $c fused"; HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py $c | head -1; done
for L in 8 12 16; do for R in 96 128 160; do echo "== config 1 L=$L R=$R"; HS_BUILD_L=$L HS_BUILD_R=$R HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py 1 | head -1; done; done
for L in 8 10 12; do for R in 64 96 128; do echo "== config 2 L=$L R=$R"; HS_BUILD_L=$L HS_BUILD_R=$R HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py 2 | head -1; done; done
} > $out/r04c_build_phases.txt 2>&1
timeout 300 bash tools/kernel_stats.sh $out/r04c_bench_kernel_stats.csv python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $out/r04c_kernel_stats.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $out/r04c_pytest.log 2>&1; echo "pytest rc $?" >> $out/r04c_pytest.log
tail -4 $out/r04c_pytest.log
cat $out/r04c_build_phases.txt $out/r04c_kernel_stats.txt
