// replay_stream.hpp — the synthetic EuRoC-shaped message stream shared by replay.cpp and tests/harness/replay_lockstep.cpp (BASELINE.json configs[4]).
//
// MH_01 itself cannot be replayed (no rosbags, ground-truth blob missing, the IMU path aborts upstream — SURVEY.md §0), so the
// stream is synthetic with the real window shape: separation 0.1 s, max_window 3.0 s (settings.yaml:145,148), <= 150 stereo tracks
// at 20 Hz through the EuRoC cameras (settings.yaml:20-72), optional IMU at 200 Hz, one optimize() per separation of data exactly
// as AbstractOptimizer::submit drives it (abstract.cpp:74-147).
#pragma once
#include <cstdlib>
#include <random>

#include "optimizer.hpp"

namespace hyper_hip {

inline Vec3 gt_position(double t) { return {2 * std::sin(0.8 * t), 2 * std::cos(0.6 * t), std::sin(0.4 * t)}; }
inline Quat gt_rotation(double t) {
  const Vec3 phi = {0.5 * std::sin(0.5 * t), 0.5 * std::cos(0.3 * t), 0.5 * std::sin(0.7 * t)};
  const double th = std::sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
  const double s = th < 1e-9 ? 0.5 : std::sin(0.5 * th) / th;
  return {s * phi[0], s * phi[1], s * phi[2], std::cos(0.5 * th)};
}
inline SE3 gt_pose(double t) { return {gt_rotation(t), gt_position(t)}; }

inline std::array<double, 2> project(const Camera& c, const Vec3& ps) {
  const double x = ps[0] / ps[2], y = ps[1] / ps[2], r2 = x * x + y * y;
  const double k1 = c.distortion[0], k2 = c.distortion[1], p1 = c.distortion[2], p2 = c.distortion[3];
  const double rad = 1 + k1 * r2 + k2 * r2 * r2;
  const double xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
  return {c.intrinsics[0] + c.intrinsics[2] * xd, c.intrinsics[1] + c.intrinsics[3] * yd};
}


inline std::vector<Camera> euroc_cameras() {  // settings.yaml:20-72
  std::vector<Camera> cams(2);
  cams[0].transformation = {{-0.007707179755532, 0.010499323370595, 0.701752800292141, 0.712301460668946}, {-0.0216401454975, -0.064676986768, 0.00981073058949}};
  cams[0].intrinsics = {367.215, 248.375, 458.654, 457.296};
  cams[0].distortion = {-0.28340811, 0.07395907, 1.76187114e-05, 0.00019359};
  cams[1].transformation = {{-0.002550236745188, 0.015323927487975, 0.702486685782579, 0.711527321918909}, {-0.0198435579556, 0.0453689425024, 0.00786212447038}};
  cams[1].intrinsics = {379.999, 255.238, 457.587, 456.134};
  cams[1].distortion = {-0.28368365, 0.07451284, -3.55590700e-05, -0.00010473};
  return cams;
}

/// Feeds `seconds` of the stream into the optimizer; `after_message()` runs after every submit().
template <class F>
void feed_stream(Optimizer& optimizer, const std::vector<Camera>& cams, double seconds, bool with_imu, F&& after_message) {
  // (HS_REPLAY_SEED: another stream of tracks and noise over the same trajectory — the lock-step harness on more window shapes; default 4)
  const char* seed_env = std::getenv("HS_REPLAY_SEED");
  std::mt19937_64 rng(0x48595045ull ^ (seed_env ? std::strtoull(seed_env, nullptr, 10) : 4ull));
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
    after_message();
  }
}

}  // namespace hyper_hip
