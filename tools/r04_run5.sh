#!/bin/bash
out=gpurun_out; mkdir -p $out
{ echo "== default"; timeout 120 python tools/build_phase_timing.py 1

  for c in 1 2 3; do echo "== config $c fused"; HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py $c | head -1; done
} > $out/r04g_build_phases.txt 2>&1
timeout 300 bash tools/kernel_stats.sh $out/r04g_bench_kernel_stats.csv python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $out/r04g_kernel_stats.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $out/r04g_pytest.log 2>&1; echo "pytest rc $?" >> $out/r04g_pytest.log
tail -4 $out/r04g_pytest.log
cat $out/r04g_build_phases.txt $out/r04g_kernel_stats.txt
