#!/bin/bash
# usage (GPU box): bash tools/r06_run.sh <tag> — GPU suite, replays (device time, host split, kernel trace), A/B of the dense solve
tag=${1:-r06}
out=gpurun_out; mkdir -p $out
(time python -m pytest tests -x -q -m gpu -n 2) > $out/${tag}_gpu_suite.log 2>&1
tail -6 $out/${tag}_gpu_suite.log
bash tools/r06_replay_host.sh $tag
grep -h "replay\|mean_solve" $out/${tag}_replay_host_split.txt | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   device %.4f  wall %.4f  stage %.4f  rmse %.4f cost %s' % (d['mean_solve_ms'], d['mean_host_wall_ms']['hs_solve'], d['mean_host_wall_ms']['stage_between_solves'], d['position_rmse_m'], d['last_cost']))
    else: print(l)
"
( cd hyperslam_amd/host; for a in "6.0 0 4" "6.0 1 4"; do echo "replay $a HS_DEBUG_FLAGS=8 (one-ended band kernels + border chain)"; HS_DEBUG_FLAGS=8 ./replay $a 2>/dev/null | tail -1; done ) > $out/${tag}_replay_dense_ab.txt 2>&1
python - <<'PY' $out/${tag}_replay_dense_ab.txt
import sys,json
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   device %.4f  wall %.4f' % (d['mean_solve_ms'], d['mean_host_wall_ms']['hs_solve']))
    else: print(l)
PY
bash tools/kernel_stats.sh $out/${tag}_replay_kernel_stats.csv hyperslam_amd/host/replay 6.0 1 4 > $out/${tag}_kernel_stats.txt 2>&1
bash tools/kernel_stats.sh $out/${tag}_replay_stereo_kernel_stats.csv hyperslam_amd/host/replay 6.0 0 4 >> $out/${tag}_kernel_stats.txt 2>&1
cat $out/${tag}_kernel_stats.txt
