#!/bin/bash
# round 4, GPU call 1: the fused build — parity suite, timing against the record path, chunk geometry sweep, kernel trace
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $out/r04a_pytest.log 2>&1; echo "pytest rc $?" >> $out/r04a_pytest.log
tail -5 $out/r04a_pytest.log
{
for c in 1 2 3; do
  echo "== config $c fused"; HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py $c
  echo "== config $c records"; HS_BUILD_PATH=records HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py $c
done
for L in 6 12 18 24 32; do for R in 128 256; do echo "== config 1 L=$L R=$R"; HS_BUILD_L=$L HS_BUILD_R=$R HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py 1 | head -1; done; done
for L in 8 12 18; do for R in 128 192; do echo "== config 2 L=$L R=$R"; HS_BUILD_L=$L HS_BUILD_R=$R HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py 2 | head -1; done; done
echo "== staged config 1"; HS_STAGE_TIMING=1 timeout 120 python tools/time_config.py 1
echo "== staged config 2"; HS_STAGE_TIMING=1 timeout 120 python tools/time_config.py 2
} > $out/r04a_timing.txt 2>&1
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline"
timeout 300 bash tools/kernel_stats.sh $out/r04a_bench_kernel_stats.csv $B > $out/r04a_kernel_stats.txt 2>&1
timeout 300 bash tools/kernel_stats.sh $out/r04a_config2_kernel_stats.csv python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline >> $out/r04a_kernel_stats.txt 2>&1
timeout 300 bash tools/kernel_stats.sh $out/r04a_config3_kernel_stats.csv python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline >> $out/r04a_kernel_stats.txt 2>&1
timeout 200 $B > $out/r04a_bench.json 2> $out/r04a_bench.err
cat $out/r04a_timing.txt | grep -v "^{" | head -60
