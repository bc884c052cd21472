#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-r04l}
for path in fused records; do
  HS_BUILD_PATH=$path timeout 300 bash tools/kernel_stats.sh $out/${tag}_config2_${path}_kernel_stats.csv python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_config2_${path}.txt 2>&1
  echo "== config 2 $path"; HS_BUILD_PATH=$path HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py 2 | head -1; HS_BUILD_PATH=$path HS_STAGE_TIMING=1 timeout 120 python tools/time_config.py 2 | tail -1
  cat $out/${tag}_config2_${path}.txt
done
for path in fused records; do echo "== config 3 $path"; HS_BUILD_PATH=$path timeout 300 bash tools/kernel_stats.sh $out/${tag}_config3_${path}_kernel_stats.csv python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline | head -9; done
