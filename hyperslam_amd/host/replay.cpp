// replay.cpp — synthetic EuRoC-shaped sliding-window replay through the AbstractOptimizer-style driver (BASELINE.json configs[4]).
//
// MH_01 itself cannot be replayed (no rosbags, ground-truth blob missing, the IMU path aborts upstream — SURVEY.md §0), so the
// stream is synthetic with the real window shape: separation 0.1 s, max_window 3.0 s (settings.yaml:145,148), <= 150 stereo tracks
// at 20 Hz through the EuRoC cameras (settings.yaml:20-72), optional IMU at 200 Hz, one optimize() per separation of data exactly
// as AbstractOptimizer::submit drives it (abstract.cpp:74-147).  Prints one JSON line.
//   usage: replay [seconds=6] [imu=0|1] [order=4] [estimation.hyper]   (4th argument: write the 100 Hz trajectory dump of main.cpp:52-80)
#include <chrono>
#include <cstring>
#include <random>

#include "optimizer.hpp"

using namespace hyper_hip;

static Vec3 gt_position(double t) { return {2 * std::sin(0.8 * t), 2 * std::cos(0.6 * t), std::sin(0.4 * t)}; }
static Quat gt_rotation(double t) {
  const Vec3 phi = {0.5 * std::sin(0.5 * t), 0.5 * std::cos(0.3 * t), 0.5 * std::sin(0.7 * t)};
  const double th = std::sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
  const double s = th < 1e-9 ? 0.5 : std::sin(0.5 * th) / th;
  return {s * phi[0], s * phi[1], s * phi[2], std::cos(0.5 * th)};
}
static SE3 gt_pose(double t) { return {gt_rotation(t), gt_position(t)}; }

static std::array<double, 2> project(const Camera& c, const Vec3& ps) {
  const double x = ps[0] / ps[2], y = ps[1] / ps[2], r2 = x * x + y * y;
  const double k1 = c.distortion[0], k2 = c.distortion[1], p1 = c.distortion[2], p2 = c.distortion[3];
  const double rad = 1 + k1 * r2 + k2 * r2 * r2;
  const double xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
  return {c.intrinsics[0] + c.intrinsics[2] * xd, c.intrinsics[1] + c.intrinsics[3] * yd};
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? std::atof(argv[1]) : 6.0;
  const bool with_imu = argc > 2 && std::atoi(argv[2]) != 0;
  Options opt;
  opt.order = argc > 3 ? std::atoi(argv[3]) : 4;
  std::vector<Camera> cams(2);
  cams[0].transformation = {{-0.007707179755532, 0.010499323370595, 0.701752800292141, 0.712301460668946}, {-0.0216401454975, -0.064676986768, 0.00981073058949}};
  cams[0].intrinsics = {367.215, 248.375, 458.654, 457.296};
  cams[0].distortion = {-0.28340811, 0.07395907, 1.76187114e-05, 0.00019359};
  cams[1].transformation = {{-0.002550236745188, 0.015323927487975, 0.702486685782579, 0.711527321918909}, {-0.0198435579556, 0.0453689425024, 0.00786212447038}};
  cams[1].intrinsics = {379.999, 255.238, 457.587, 456.134};
  cams[1].distortion = {-0.28368365, 0.07451284, -3.55590700e-05, -0.00010473};
  IMU imu;
  Optimizer optimizer(opt, cams, with_imu ? &imu : nullptr);

  std::mt19937_64 rng(0x48595045ull ^ 4);
  std::uniform_real_distribution<double> U(0, 1);
  std::normal_distribution<double> Nrm(0, 1);
  struct Track {
    int64_t id;
    Vec3 p_w;
    int age;
  };
  std::vector<Track> tracks;
  int64_t next_id = 0;
  const int max_tracks = 150;  // settings.yaml:118
  const double t_start = 10.0;  // arbitrary root stamp (the optimizer subtracts it)
  double total_solve_ms = 0, max_solve_ms = 0, stage_ms[4] = {0, 0, 0, 0};
  long total_blocks = 0;
  int solves_seen = 0;
  const auto wall0 = std::chrono::steady_clock::now();
  const double frame_dt = 0.05, imu_dt = 0.005;
  double next_frame = 0, next_imu = 0;
  const Vec3 g_true = {0, 0, -9.80665};
  if (with_imu) optimizer.setGravity(g_true);
  for (double t = 0; t < seconds;) {
    const bool do_frame = next_frame <= next_imu || !with_imu;
    t = do_frame ? next_frame : next_imu;
    if (t >= seconds) break;
    if (do_frame) {
      next_frame += frame_dt;
      const SE3 T_wb = gt_pose(t);
      const SE3 T_w0 = groupPlus(T_wb, cams[0].transformation), T_w1 = groupPlus(T_wb, cams[1].transformation);
      const SE3 T_0w = groupInverse(T_w0), T_1w = groupInverse(T_w1);
      VisualTracks m;
      m.stamp = t_start + t;
      std::vector<Track> alive;
      for (Track& tr : tracks) {
        const Vec3 p0 = vectorPlus(T_0w, tr.p_w), p1 = vectorPlus(T_1w, tr.p_w);
        if (p0[2] < 0.5 || p1[2] < 0.5 || tr.age > 60) continue;
        const auto a = project(cams[0], p0), b = project(cams[1], p1);
        if (a[0] < 0 || a[0] >= 752 || a[1] < 0 || a[1] >= 480 || b[0] < 0 || b[0] >= 752 || b[1] < 0 || b[1] >= 480) continue;
        m.identifiers.push_back(tr.id);
        m.P0.push_back({a[0] + 0.5 * Nrm(rng), a[1] + 0.5 * Nrm(rng)});
        m.P1.push_back({b[0] + 0.5 * Nrm(rng), b[1] + 0.5 * Nrm(rng)});
        tr.age++;
        alive.push_back(tr);
      }
      tracks.swap(alive);
      while (int(tracks.size()) < max_tracks) {  // new features: random pixel in camera 0, depth 2..10 m
        const double u = 752 * U(rng), v = 480 * U(rng), depth = 2 + 8 * U(rng);
        const Vec3 bdir = cams[0].pixelToBearing(u, v);
        const Vec3 p0 = {bdir[0] / bdir[2] * depth, bdir[1] / bdir[2] * depth, depth};
        const Vec3 pw = vectorPlus(T_w0, p0);
        const Vec3 p1 = vectorPlus(T_1w, pw);
        const auto b = project(cams[1], p1);
        if (p1[2] < 0.5 || b[0] < 0 || b[0] >= 752 || b[1] < 0 || b[1] >= 480) continue;
        tracks.push_back({next_id, pw, 1});
        m.identifiers.push_back(next_id++);
        m.P0.push_back({u + 0.5 * Nrm(rng), v + 0.5 * Nrm(rng)});
        m.P1.push_back({b[0] + 0.5 * Nrm(rng), b[1] + 0.5 * Nrm(rng)});
      }
      optimizer.submit(m);
    } else {
      next_imu += imu_dt;
      const double h = 1e-4;
      const Quat q0 = gt_rotation(t), qp = gt_rotation(t + h), qm = gt_rotation(t - h);
      const Quat dq = mul(conj(qm), qp);  // ~ Exp(2h w_b)
      const Vec3 w_b = {dq.x / h, dq.y / h, dq.z / h};
      const Vec3 pp = gt_position(t + h), p0 = gt_position(t), pm = gt_position(t - h);
      const Vec3 a_w = {(pp[0] - 2 * p0[0] + pm[0]) / (h * h) - g_true[0], (pp[1] - 2 * p0[1] + pm[1]) / (h * h) - g_true[1],
                        (pp[2] - 2 * p0[2] + pm[2]) / (h * h) - g_true[2]};
      const Vec3 a_b = rotate(conj(q0), a_w);
      InertialMeasurement m;
      m.stamp = t_start + t;
      const double sg = 1.6968e-04 * std::sqrt(200.0), sa = 2.0e-3 * std::sqrt(200.0);
      for (int c = 0; c < 3; ++c) m.value[c] = w_b[c] + sg * Nrm(rng), m.value[3 + c] = a_b[c] + sa * Nrm(rng);
      optimizer.submit(m);
    }
    if (optimizer.numOptimizations() > solves_seen) {
      solves_seen = optimizer.numOptimizations();
      const hs_summary& s = optimizer.lastSummary();
      total_solve_ms += s.total_ms, max_solve_ms = std::max(max_solve_ms, s.total_ms);
      total_blocks += long(s.num_residual_blocks) * s.num_iterations;
      stage_ms[0] += s.linearize_ms, stage_ms[1] += s.schur_ms, stage_ms[2] += s.solve_ms, stage_ms[3] += s.update_ms;
      if (std::getenv("HS_REPLAY_TRACE"))
        std::fprintf(stderr, "opt %3d  cps %3zu  lms %4zu  blocks %6d  iters %d  ok %d  term %d  cost %.12g -> %.12g\n", solves_seen, optimizer.numControlPoints(),
                     optimizer.numLandmarks(), s.num_residual_blocks, s.num_iterations, s.num_successful_steps, s.termination, s.initial_cost, s.final_cost);
    }
  }
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
  std::printf("{\"replay_seconds\": %.2f, \"imu\": %d, \"order\": %d, \"optimizations\": %d, \"control_points\": %zu, \"landmarks\": %zu, "
              "\"mean_solve_ms\": %.4f, \"max_solve_ms\": %.4f, \"residual_blocks_per_s_in_solve\": %.1f, \"wall_ms\": %.1f, "
              "\"window\": [%.2f, %.2f], \"position_rmse_m\": %.4f, \"last_cost\": [%.6g, %.6g], "
              "\"mean_stage_ms\": {\"linearize\": %.4f, \"schur\": %.4f, \"solve\": %.4f, \"update\": %.4f}, "
              "\"mean_host_wall_ms\": {\"tables\": %.4f, \"hs_solve\": %.4f, \"readback\": %.4f}}\n",
              seconds, int(with_imu), opt.order, optimizer.numOptimizations(), optimizer.numControlPoints(), optimizer.numLandmarks(),
              total_solve_ms / std::max(1, solves_seen), max_solve_ms, total_solve_ms > 0 ? 1e3 * total_blocks / total_solve_ms : 0.0, wall_ms,
              optimizer.window().lower, optimizer.window().upper, std::sqrt(se / std::max(1, n)), optimizer.lastSummary().initial_cost,
              optimizer.lastSummary().final_cost, stage_ms[0] / std::max(1, solves_seen), stage_ms[1] / std::max(1, solves_seen),
              stage_ms[2] / std::max(1, solves_seen), stage_ms[3] / std::max(1, solves_seen), optimizer.wallSplitMs()[0] / std::max(1, solves_seen),
              optimizer.wallSplitMs()[1] / std::max(1, solves_seen), optimizer.wallSplitMs()[2] / std::max(1, solves_seen));
  return 0;
}
