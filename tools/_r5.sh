set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 700 python -m pytest tests -x -q -m gpu > gpurun_out/r5_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_pytest.log
tail -5 gpurun_out/r5_pytest.log
cd hyperslam_amd/host
( for fl in 0 4194304 8388608 16777216 29360128; do for a in "6.0 1 4" "6.0 0 4"; do echo "flags=$fl args=$a"; HS_DEBUG_FLAGS=$fl timeout 120 ./replay $a 2>/dev/null | tail -1; done; done ) > ../../gpurun_out/r5_replay_ab.txt 2>&1
cd ../..
for a in "3.6 1 4" "6.0 1 6"; do timeout 200 hyperslam_amd/host/replay_lockstep hyperslam_amd/libhyperslam_hip.so $a > gpurun_out/r5_lockstep_$(echo $a | tr ' .' '__').jsonl 2>&1; done
tail -2 gpurun_out/r5_lockstep_*.jsonl | cut -c1-600
cat gpurun_out/r5_replay_ab.txt | cut -c1-330
