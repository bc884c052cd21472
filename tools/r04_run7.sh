#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-r04i}
{ for c in 1 2 3; do echo "== config $c fused"; HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py $c | head -1; done
} > $out/${tag}_timing.txt 2>&1
timeout 300 bash tools/kernel_stats.sh $out/${tag}_bench_kernel_stats.csv python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $out/${tag}_kernel_stats.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $out/${tag}_pytest.log 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest.log
tail -25 $out/${tag}_pytest.log
cat $out/${tag}_timing.txt $out/${tag}_kernel_stats.txt
