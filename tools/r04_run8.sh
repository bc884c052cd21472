#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-r04j}
{ for c in 1 2 3; do echo "== config $c fused"; HS_STAGE_TIMING=0 timeout 120 python tools/time_config.py $c | head -1; done
} > $out/${tag}_timing.txt 2>&1
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline"
timeout 300 bash tools/kernel_stats.sh $out/${tag}_bench_kernel_stats.csv $B > $out/${tag}_kernel_stats.txt 2>&1
timeout 300 bash tools/pmc_traffic.sh $out/${tag}_pmc_hbm_traffic.json $B > $out/${tag}_pmc.txt 2>&1
timeout 300 $B > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -q --maxfail=30 -p no:cacheprovider > $out/${tag}_pytest.log 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest.log
tail -5 $out/${tag}_pytest.log
cat $out/${tag}_timing.txt $out/${tag}_kernel_stats.txt $out/${tag}_pmc.txt; cat $out/${tag}_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_gn_iteration','device_ms_per_iteration')}); print(d['roofline'])"
