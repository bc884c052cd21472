#!/bin/bash
out=gpurun_out; mkdir -p $out
{ for c in 1 2 3; do echo "== config $c"; HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py $c | head -1; done; } > $out/r04o_configs.txt 2>&1
timeout 300 bash tools/kernel_stats.sh $out/r04o_bench_kernel_stats.csv python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $out/r04o_kernel_stats.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "parity or edge or solve_golden or distributed or inertial" > $out/r04o_pytest.log 2>&1; echo "pytest rc $?" >> $out/r04o_pytest.log
tail -4 $out/r04o_pytest.log
cat $out/r04o_configs.txt; head -8 $out/r04o_kernel_stats.txt
timeout 120 python tools/chol_phase_timing.py 2>&1 | tail -8
