#!/bin/bash
# usage (GPU box): bash tools/r06_quick.sh <tag> — the dense-solve kernel's tests, its phase stamps (profiling build) and the three 6 s replays
tag=${1:-r06q}
out=gpurun_out; mkdir -p $out
(time python -m pytest tests -x -q -m gpu -n 2) > $out/${tag}_tests.log 2>&1
tail -4 $out/${tag}_tests.log
{ python tools/dense_mx_phase_timing.py 33 1; python tools/dense_mx_phase_timing.py 33 0; } > $out/${tag}_dense_mx_phase_timing.txt 2>&1
cat $out/${tag}_dense_mx_phase_timing.txt
( cd hyperslam_amd/host; for a in "6.0 0 4" "6.0 1 4" "6.0 1 6"; do echo "replay $a"; ./replay $a 2>/dev/null | tail -1; done ) > $out/${tag}_replay.txt 2>&1
python - <<'PY' $out/${tag}_replay.txt
import sys,json
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   device %.4f  wall %.4f  rmse %.4f cost %s' % (d['mean_solve_ms'], d['mean_host_wall_ms']['hs_solve'], d['position_rmse_m'], d['last_cost']))
    else: print(l)
PY
