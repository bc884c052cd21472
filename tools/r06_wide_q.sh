#!/bin/bash
# usage (GPU box): bash tools/r06_wide_q.sh <tag> — window-wide bands: the landmark term once per window (k_landmark_gram_wide, default) against
# once per chunk (HS_WIDE_Q=0): GPU suite, the three 6 s replays alternating on one box, kernel averages of the stereo replay
tag=${1:-r06wq}
out=gpurun_out; mkdir -p $out
(time python -m pytest tests -x -q -m gpu -n 2) > $out/${tag}_tests.log 2>&1
grep -h "passed\|failed" $out/${tag}_tests.log | tail -1
( cd hyperslam_amd/host; for r in 1 2; do for a in "6.0 0 4" "6.0 1 4" "6.0 1 6"; do for q in 1 0; do
  HS_WIDE_Q=$q ./replay $a 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('replay $a HS_WIDE_Q=$q device %.4f  wall %.4f  rmse %.4f  cost %s' % (d['mean_solve_ms'], d['mean_host_wall_ms']['hs_solve'], d['position_rmse_m'], d['last_cost']))"
done; done; done ) > $out/${tag}_replay_ab.txt 2>&1
cat $out/${tag}_replay_ab.txt
bash tools/kernel_stats.sh $out/${tag}_replay_stereo_kernel_stats.csv hyperslam_amd/host/replay 6.0 0 4 > $out/${tag}_kernel_stats.txt 2>&1
head -8 $out/${tag}_replay_stereo_kernel_stats.csv | cut -c1-120
