// replay.cpp — synthetic EuRoC-shaped sliding-window replay through the AbstractOptimizer-style driver (BASELINE.json configs[4]).
//
// MH_01 itself cannot be replayed (no rosbags, ground-truth blob missing, the IMU path aborts upstream — SURVEY.md §0), so the
// stream is synthetic with the real window shape: separation 0.1 s, max_window 3.0 s (settings.yaml:145,148), <= 150 stereo tracks
// at 20 Hz through the EuRoC cameras (settings.yaml:20-72), optional IMU at 200 Hz, one optimize() per separation of data exactly
// as AbstractOptimizer::submit drives it (abstract.cpp:74-147).  Prints one JSON line.
//   usage: replay [seconds=6] [imu=0|1] [order=4] [estimation.hyper]   (4th argument: write the 100 Hz trajectory dump of main.cpp:52-80)
#include <algorithm>
#include <chrono>
#include <cstring>

#include "replay_stream.hpp"

using namespace hyper_hip;

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? std::atof(argv[1]) : 6.0;
  const bool with_imu = argc > 2 && std::atoi(argv[2]) != 0;
  Options opt;
  opt.order = argc > 3 ? std::atoi(argv[3]) : 4;
  const std::vector<Camera> cams = euroc_cameras();
  IMU imu;
  Optimizer optimizer(opt, cams, with_imu ? &imu : nullptr);
  // the per-stage split printed below needs the stage events (four barrier packets per iteration, off by default): HS_REPLAY_STAGES=1
  const bool stages = std::getenv("HS_REPLAY_STAGES") && std::atoi(std::getenv("HS_REPLAY_STAGES")) != 0;
  if (stages && HSF(set_stage_timing)(optimizer.handle(), 1) != HS_OK) return 1;
  double total_solve_ms = 0, max_solve_ms = 0, stage_ms[4] = {0, 0, 0, 0};
  long total_blocks = 0;
  int solves_seen = 0;
  std::vector<double> solve_ms;  // per optimize(): the median and the full-window mean do not see the one-off first-launch cost
  const auto wall0 = std::chrono::steady_clock::now();
  feed_stream(optimizer, cams, seconds, with_imu, [&] {
    if (optimizer.numOptimizations() > solves_seen) {
      solves_seen = optimizer.numOptimizations();
      const hs_summary& s = optimizer.lastSummary();
      total_solve_ms += s.total_ms, max_solve_ms = std::max(max_solve_ms, s.total_ms);
      solve_ms.push_back(s.total_ms);
      total_blocks += long(s.num_residual_blocks) * s.num_iterations;
      if (stages) stage_ms[0] += s.linearize_ms, stage_ms[1] += s.schur_ms, stage_ms[2] += s.solve_ms, stage_ms[3] += s.update_ms;
      if (std::getenv("HS_REPLAY_TRACE"))
        std::fprintf(stderr, "opt %3d  cps %3zu  lms %4zu  blocks %6d  iters %d  ok %d  term %d  cost %.12g -> %.12g\n", solves_seen, optimizer.numControlPoints(),
                     optimizer.numLandmarks(), s.num_residual_blocks, s.num_iterations, s.num_successful_steps, s.termination, s.initial_cost, s.final_cost);
    }
  });
  const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  if (argc > 4) {
    const int n_samples = optimizer.writeEstimation(argv[4]);
    std::fprintf(stderr, "wrote %d samples to %s\n", n_samples, argv[4]);
  }
  // accuracy: control points vs ground truth after aligning the first frozen control point (gauge)
  const SE3 A = groupPlus(gt_pose(optimizer.controlPointStamp(0)), groupInverse(optimizer.controlPoint(0)));
  double se = 0;
  int n = 0;
  for (size_t j = 0; j + opt.order < optimizer.numControlPoints(); ++j) {  // control points the data already constrains
    const Vec3 e = vectorPlus(A, optimizer.controlPoint(j).p), g = gt_position(optimizer.controlPointStamp(j));
    se += (e[0] - g[0]) * (e[0] - g[0]) + (e[1] - g[1]) * (e[1] - g[1]) + (e[2] - g[2]) * (e[2] - g[2]);
    ++n;
  }
  // steady state: the second half of the calls (full 3 s windows, everything loaded)
  double steady = 0, median = 0;
  if (!solve_ms.empty()) {
    const size_t h = solve_ms.size() / 2;
    for (size_t i = h; i < solve_ms.size(); ++i) steady += solve_ms[i] / double(solve_ms.size() - h);
    std::vector<double> sorted = solve_ms;
    std::sort(sorted.begin(), sorted.end());
    median = sorted[sorted.size() / 2];
  }
  std::printf("{\"replay_seconds\": %.2f, \"imu\": %d, \"order\": %d, \"optimizations\": %d, \"control_points\": %zu, \"landmarks\": %zu, "
              "\"mean_solve_ms\": %.4f, \"median_solve_ms\": %.4f, \"full_window_mean_solve_ms\": %.4f, \"max_solve_ms\": %.4f, \"residual_blocks_per_s_in_solve\": %.1f, \"wall_ms\": %.1f, "
              "\"window\": [%.2f, %.2f], \"state_range\": [%.6f, %.6f], \"position_rmse_m\": %.4f, \"last_cost\": [%.6g, %.6g], "
              "\"mean_stage_ms\": {\"linearize\": %.4f, \"schur\": %.4f, \"solve\": %.4f, \"update\": %.4f}, "
              "\"mean_host_wall_ms\": {\"tables\": %.4f, \"hs_solve\": %.4f, \"readback\": %.4f, \"stage_between_solves\": %.4f}, \"delta_interface\": %d}\n",
              seconds, int(with_imu), opt.order, optimizer.numOptimizations(), optimizer.numControlPoints(), optimizer.numLandmarks(),
              total_solve_ms / std::max(1, solves_seen), median, steady, max_solve_ms, total_solve_ms > 0 ? 1e3 * total_blocks / total_solve_ms : 0.0, wall_ms,
              optimizer.window().lower, optimizer.window().upper, optimizer.stateRange().lower, optimizer.stateRange().upper, std::sqrt(se / std::max(1, n)), optimizer.lastSummary().initial_cost,
              optimizer.lastSummary().final_cost, stages ? stage_ms[0] / std::max(1, solves_seen) : -1.0, stages ? stage_ms[1] / std::max(1, solves_seen) : -1.0,
              stages ? stage_ms[2] / std::max(1, solves_seen) : -1.0, stages ? stage_ms[3] / std::max(1, solves_seen) : -1.0, optimizer.wallSplitMs()[0] / std::max(1, solves_seen),
              optimizer.wallSplitMs()[1] / std::max(1, solves_seen), optimizer.wallSplitMs()[2] / std::max(1, solves_seen), optimizer.wallStageMs() / std::max(1, solves_seen),
              int(optimizer.deltaInterface()));
  return 0;
}
