#!/bin/bash
# usage (GPU box): bash tools/r06_ab.sh <HS_DEBUG_FLAGS value of the B side> [repetitions] — the three 6 s replays, A (product) and B alternating on the same box
flags=${1:-134217728}; reps=${2:-2}
cd hyperslam_amd/host
for r in $(seq $reps); do for a in "6.0 0 4" "6.0 1 4" "6.0 1 6"; do for f in 0 $flags; do
  HS_DEBUG_FLAGS=$f ./replay $a 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('replay $a flags %-10s device %.4f  wall %.4f' % ('$f', d['mean_solve_ms'], d['mean_host_wall_ms']['hs_solve']))"
done; done; done
