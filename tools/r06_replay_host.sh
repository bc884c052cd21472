#!/bin/bash
# usage (GPU box): bash tools/r06_replay_host.sh <tag>   — the 6 s replays with the delta interface (default) and with every table rebuilt inside
# optimize() (HS_REPLAY_FULL_TABLES=1, rounds 1-5): device time per optimize(), hs_solve wall, host split of prepare()
tag=${1:-r06}
out=gpurun_out; mkdir -p $out
cd hyperslam_amd/host
{ for a in "6.0 0 4" "6.0 1 4" "6.0 1 6"; do for full in 0 1; do
    echo "replay $a  HS_REPLAY_FULL_TABLES=$full"
    HS_REPLAY_FULL_TABLES=$full ./replay $a 2>/dev/null | tail -1
    HS_REPLAY_FULL_TABLES=$full HS_HOST_TIMING=1 ./replay $a 2>&1 >/dev/null | grep "host timing"
  done; done; } > ../../$out/${tag}_replay_host_split.txt 2>&1
