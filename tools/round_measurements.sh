#!/bin/bash
# usage (on the GPU box): bash tools/round_measurements.sh <tag>    e.g. r02_v2  -> gpurun_out/<tag>_*
# Everything DESIGN.md quotes for a round: smoke, the bench line (with the CPU baseline), kernel traces, PMC passes, replays, lock-step logs.
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline"
bash tools/kernel_stats.sh $out/${tag}_bench_kernel_stats.csv $B > $out/${tag}_kernel_stats.txt 2>&1
bash tools/kernel_stats.sh $out/${tag}_config2_kernel_stats.csv python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline >> $out/${tag}_kernel_stats.txt 2>&1
bash tools/kernel_stats.sh $out/${tag}_replay_kernel_stats.csv hyperslam_amd/host/replay 6.0 1 4 >> $out/${tag}_kernel_stats.txt 2>&1
bash tools/pmc_traffic.sh $out/${tag}_pmc_hbm_traffic.json $B > $out/${tag}_pmc.txt 2>&1
bash tools/pmc_sq.sh $out/${tag}_pmc_sq.json $B >> $out/${tag}_pmc.txt 2>&1
# (counter passes serialise the kernels of a process: the border sweep that follows the factorisation row by row must not wait for a kernel that
#  cannot run next to it — HS_DEBUG_FLAGS=128, the sequential arrangement, for configs[2] under --pmc)
HS_DEBUG_FLAGS=128 bash tools/pmc_sq.sh $out/${tag}_config2_pmc_sq.json python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline >> $out/${tag}_pmc.txt 2>&1
# matrix-core counters of the factorisation: k_band_factor_mx (default since round 5: trailing window in f64-MFMA accumulators) and the VALU
# look-ahead kernel k_band_factor_la (A/B switch 64); configs[2] runs the WIDE instance
bash tools/pmc_mfma.sh $out/${tag}_pmc_mfma_mx_factor.json $B >> $out/${tag}_pmc.txt 2>&1
HS_DEBUG_FLAGS=64 bash tools/pmc_mfma.sh $out/${tag}_pmc_mfma_valu_factor.json $B >> $out/${tag}_pmc.txt 2>&1
HS_DEBUG_FLAGS=128 bash tools/pmc_mfma.sh $out/${tag}_config2_pmc_mfma_mx_factor.json python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline >> $out/${tag}_pmc.txt 2>&1
bash tools/kernel_stats.sh $out/${tag}_config3_kernel_stats.csv python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline >> $out/${tag}_kernel_stats.txt 2>&1
[ -f tools/libhyperslam_hip_prof.so ] && { python tools/mx_phase_timing.py 1; python tools/mx_phase_timing.py 3 | tail -4; echo "--- k_band_factor_la (A/B switch 64)"; HS_DEBUG_FLAGS=80 python tools/chol_phase_timing.py; } > $out/${tag}_chol_phase_timing.txt 2>&1
[ -f tools/libhyperslam_hip_prof.so ] && python tools/build_phase_timing.py 1 > $out/${tag}_build_phase_timing.txt 2>&1
[ -f tools/libhyperslam_hip_prof.so ] && { python tools/fold_phase_timing.py 1; HS_DEBUG_FLAGS=32768 python tools/fold_phase_timing.py 1; python tools/fold_phase_timing.py 3; } > $out/${tag}_fold_phase_timing.txt 2>&1
for c in 0 1 2 3; do HS_STAGE_TIMING=0 python tools/time_config.py $c; HS_STAGE_TIMING=1 python tools/time_config.py $c; done > $out/${tag}_configs.txt 2>&1
python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_config3.json 2>/dev/null
python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_config2.json 2>/dev/null
( cd hyperslam_amd/host
  for a in "6.0 1 4" "6.0 0 4" "6.0 1 6" "8.0 1 4"; do ./replay $a 2>/dev/null | tail -1; HS_STAGE_TIMING=1 ./replay $a 2>/dev/null | tail -1; done
  ./replay_oracle 6.0 1 4 2>/dev/null | tail -1 ) > $out/${tag}_replay.txt 2>&1
( cd hyperslam_amd/host; for a in "6.0 1 4" "6.0 0 4"; do echo "replay $a"; HS_HOST_TIMING=1 ./replay $a 2>&1 >/dev/null | grep "host timing"; done ) > $out/${tag}_replay_host_split.txt 2>&1
for a in "3.6 0 4" "3.6 1 4" "6.0 1 6" "6.0 1 4"; do tests/harness/replay_lockstep hyperslam_amd/libhyperslam_hip.so $a > $out/${tag}_lockstep_$(echo $a | tr ' .' '__').jsonl 2>&1; done
# round 5 additions: oracle(double) / oracle(long double) / HIP from identical tables; the pipelined border sweep against the sequential one
# (bit identity under repetition, configs[2] timing); the fused build on window-wide bands against the record path on the replays
python tools/lockstep_three_way.py 3.6 0 4 --out $out/${tag}_three_way_3_6_0_4.txt > /dev/null 2>&1
python tools/stress_pipelined_sweep.py 40 > $out/${tag}_stress_pipelined_sweep.txt 2>&1
{ for i in 1 2; do echo "pipelined (default)"; HS_STAGE_TIMING=0 python tools/time_config.py 2 | head -1; echo "sweep behind the factorisation (HS_DEBUG_FLAGS=128)"; HS_DEBUG_FLAGS=128 HS_STAGE_TIMING=0 python tools/time_config.py 2 | head -1; done; } > $out/${tag}_config2_sweep_ab.txt 2>&1
( cd hyperslam_amd/host
  for a in "6.0 0 4" "6.0 1 4" "6.0 1 6"; do for pth in narrow default; do echo "replay $a build path: $pth (narrow = record path above 256 window tiles, round 4's rule)"
    if [ $pth = narrow ]; then export HS_BUILD_PATH=narrow; else unset HS_BUILD_PATH; fi; ./replay $a 2>/dev/null | tail -1; ./replay $a 2>/dev/null | tail -1; done; done; unset HS_BUILD_PATH ) > $out/${tag}_replay_build_path_ab.txt 2>&1
# round 6 additions: the dense solve of small systems (k_dense_solve_mx) — phase stamps, A/B against the round-5 chain — and the delta interface
# (tables kept between solves: hs_append_* / hs_retire_* / hs_stage) against whole-table uploads inside optimize()
[ -f tools/libhyperslam_hip_prof.so ] && { python tools/dense_mx_phase_timing.py 33 1; python tools/dense_mx_phase_timing.py 33 0; } > $out/${tag}_dense_mx_phase_timing.txt 2>&1
bash tools/r06_replay_host.sh $tag
( cd hyperslam_amd/host; for a in "6.0 0 4" "6.0 1 4" "6.0 1 6"; do echo "replay $a HS_DEBUG_FLAGS=8 (round 5: one-ended band kernels + border chain + k_band_backward)"; HS_DEBUG_FLAGS=8 ./replay $a 2>/dev/null | tail -1; echo "replay $a (k_dense_solve_mx)"; ./replay $a 2>/dev/null | tail -1; done ) > $out/${tag}_replay_dense_ab.txt 2>&1
bash tools/kernel_stats.sh $out/${tag}_replay_stereo_kernel_stats.csv hyperslam_amd/host/replay 6.0 0 4 >> $out/${tag}_kernel_stats.txt 2>&1
# later in round 6: A/B switches of the launch structure on the replays, each against the product on the same box (tools/r06_ab.sh) —
# 134217728 prior / inertial candidate costs as launches of their own; 16384 border gathers joined by an event instead of the device flag;
# 67108864 no deferred commit (small visual-only windows: k_pack_decision in every iteration instead of the decision folded into the next build)
{ for f in 134217728 16384 67108864; do echo "--- A/B switch $f"; bash tools/r06_ab.sh $f 2; done; } > $out/${tag}_replay_launch_ab.txt 2>&1
{ for i in 1 2; do echo "gather flag (default)"; HS_STAGE_TIMING=0 python tools/time_config.py 2 | head -1; echo "event between the streams (HS_DEBUG_FLAGS=16384)"; HS_DEBUG_FLAGS=16384 HS_STAGE_TIMING=0 python tools/time_config.py 2 | head -1; done; } > $out/${tag}_config2_gather_ab.txt 2>&1
bash tools/iteration_timeline.sh 2 > $out/${tag}_timeline_config2.txt 2>&1
hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_probe tools/microbench/dpp_f64_probe.hip > /dev/null 2>&1 && /tmp/dpp_probe > $out/${tag}_dpp_f64_probe.txt 2>&1

# last session of round 6: window-wide bands with the landmark term once per window (Tables::wide_q) against once per chunk (HS_WIDE_Q=0), alternating
( cd hyperslam_amd/host; for r in 1 2; do for a in "6.0 0 4" "6.0 1 4" "6.0 1 6"; do for q in 1 0; do echo -n "replay $a HS_WIDE_Q=$q: "; HS_WIDE_Q=$q ./replay $a 2>/dev/null | tail -1 | cut -c1-160; done; done; done ) > $out/${tag}_replay_wide_q_ab.txt 2>&1
[ -f tools/libhyperslam_hip_prof.so ] && { python tools/assemble_phase_timing.py r; python tools/build_phase_timing.py r | head -18; } > $out/${tag}_wide_band_phase_timing.txt 2>&1
echo done
