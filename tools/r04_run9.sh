#!/bin/bash
out=gpurun_out; mkdir -p $out; tag=${1:-r04k}
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $out/${tag}_pytest.log 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest.log
tail -30 $out/${tag}_pytest.log
