cd $GRAFT_REPO_ROOT
timeout 200 python tools/dense_phase_timing.py 33 0 2>&1 | sed -n 1,12p
timeout 200 python tools/dense_phase_timing.py 33 0 2>&1 | tail -4
timeout 400 python -m pytest tests/test_gpu_edge_cases.py tests/test_host_driver.py -x -q -m gpu 2>&1 | tail -3
( cd hyperslam_amd/host; for a in "6.0 1 4" "6.0 0 4"; do timeout 120 ./replay $a 2>/dev/null | tail -1 | cut -c1-300; done )
