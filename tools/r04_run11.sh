#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-r04m}
timeout 900 python -m pytest tests/test_gpu_inertial.py tests/test_sensor_blocks.py tests/test_solve_golden.py "tests/test_gpu_parity.py::test_baseline_configs_at_full_size" tests/test_host_driver.py -m gpu -q --maxfail=30 -p no:cacheprovider > $out/${tag}_pytest.log 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest.log
tail -6 $out/${tag}_pytest.log
echo "== config 2"; HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py 2 | head -1; HS_STAGE_TIMING=1 timeout 120 python tools/time_config.py 2 | tail -1
timeout 300 bash tools/kernel_stats.sh $out/${tag}_config2_kernel_stats.csv python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline
