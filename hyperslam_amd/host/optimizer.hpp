// optimizer.hpp — C++ host-side mirror of HyperSLAM's optimizer plugin surface on top of the C ABI.
//
// Restates the *structure-producing* behaviour of
//   AbstractOptimizer::submit / setWindow / process   /root/reference/internal/hyper/optimizers/abstract.cpp:40-292
//   CeresOptimizer::updateState / updateLandmarks     /root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:286-382
// with the same names and argument meaning, using plain structs instead of the (absent) HyperVariables / HyperSensors types:
// which residuals exist, which control points are frozen, when optimize() runs and how the window grows / slides.
// All arithmetic of optimize() happens behind hs_solve (gfx950 kernels); this file is host bookkeeping only.
//
// The C ABI prefix is a macro so that the identical driver can be linked against the oracle (prefix hso_) in CPU tests.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iomanip>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hyperslam_hip.h"

#ifndef HS_ABI_PREFIX
#define HS_ABI_PREFIX hs_
#endif
#define HS_CAT2(a, b) a##b
#define HS_CAT(a, b) HS_CAT2(a, b)
#define HSF(name) HS_CAT(HS_ABI_PREFIX, name)

// the oracle exports the same entry points under its own prefix
extern "C" {
int HSF(create)(int, void*, hs_problem**);
int HSF(destroy)(hs_problem*);
const char* HSF(last_error)(const hs_problem*);
int HSF(set_spline)(hs_problem*, int, double, double, int, const double*, const uint8_t*, int, int);
int HSF(set_cameras)(hs_problem*, int, const double*, const double*, const double*);
int HSF(set_sensors)(hs_problem*, int, const double*);
int HSF(set_landmarks)(hs_problem*, int, const double*, const uint8_t*);
int HSF(set_imu)(hs_problem*, const double*, const double*, const double*, const double*, const double*, int, double, double, int, const double*,
                 const double*, int);
int HSF(set_gravity)(hs_problem*, const double*, int);
int HSF(set_bearing_residuals)(hs_problem*, int, const double*, const double*, const int32_t*, const int32_t*);
int HSF(set_pixel_residuals)(hs_problem*, int, const double*, const double*, const int32_t*, const int32_t*);
int HSF(set_prior_residuals)(hs_problem*, int, const double*, const double*, const int32_t*);
int HSF(set_inertial_residuals)(hs_problem*, int, const double*, const double*);
int HSF(solve)(hs_problem*, int, hs_summary*, hs_iteration*);
int HSF(get_control_points)(hs_problem*, double*);
int HSF(get_landmarks)(hs_problem*, double*);
int HSF(get_bias)(hs_problem*, double*, double*);
int HSF(get_gravity)(hs_problem*, double*);
int HSF(sample_trajectory)(hs_problem*, int, const double*, double*, double*, double*);
int HSF(process_tracks)(hs_problem*, double, int, const double*, const double*, double*, double*, double*);
int HSF(set_stage_timing)(hs_problem*, int);
int HSF(append_landmarks)(hs_problem*, int, const double*, const uint8_t*, int32_t*);
int HSF(append_bearing_residuals)(hs_problem*, int, const double*, const double*, const int32_t*, const int32_t*);
int HSF(append_prior_residuals)(hs_problem*, int, const double*, const double*, const int32_t*);
int HSF(append_inertial_residuals)(hs_problem*, int, const double*, const double*);
int HSF(retire_landmarks)(hs_problem*, int, const int32_t*, int32_t*);
int HSF(retire_residuals_before)(hs_problem*, int, double);
int HSF(stage)(hs_problem*);
}

namespace hyper_hip {

using Stamp = double;
using Scalar = double;

// ---- minimal value types (stand-ins for hyper::SE3 / Bearing / ...) ------------------------------------------------------
struct Quat {
  double x = 0, y = 0, z = 0, w = 1;
};
inline Quat mul(const Quat& a, const Quat& b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
          a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Quat conj(const Quat& a) { return {-a.x, -a.y, -a.z, a.w}; }
using Vec3 = std::array<double, 3>;
inline Vec3 rotate(const Quat& q, const Vec3& v) {
  const Quat r = mul(mul(q, Quat{v[0], v[1], v[2], 0}), conj(q));
  return {r.x, r.y, r.z};
}
struct SE3 {
  Quat q;
  Vec3 p{0, 0, 0};
};
inline SE3 groupPlus(const SE3& a, const SE3& b) {  // a o b
  const Vec3 t = rotate(a.q, b.p);
  return {mul(a.q, b.q), {t[0] + a.p[0], t[1] + a.p[1], t[2] + a.p[2]}};
}
inline SE3 groupInverse(const SE3& a) {
  const Quat qi = conj(a.q);
  const Vec3 t = rotate(qi, a.p);
  return {qi, {-t[0], -t[1], -t[2]}};
}
inline Vec3 vectorPlus(const SE3& a, const Vec3& v) {
  const Vec3 t = rotate(a.q, v);
  return {t[0] + a.p[0], t[1] + a.p[1], t[2] + a.p[2]};
}

struct Camera {  // sensors[] entry of settings.yaml (transformation, intrinsics [cx cy fx fy], radtan distortion)
  SE3 transformation;
  std::array<double, 4> intrinsics{0, 0, 1, 1};
  std::array<double, 4> distortion{0, 0, 0, 0};

  /// EXTERNAL Camera::convertPixelsToBearings (call site abstract.cpp:222-223): undistort (fixed point) and normalise.
  Vec3 pixelToBearing(double u, double v) const {
    const double xd = (u - intrinsics[0]) / intrinsics[2], yd = (v - intrinsics[1]) / intrinsics[3];
    double x = xd, y = yd;
    const double k1 = distortion[0], k2 = distortion[1], p1 = distortion[2], p2 = distortion[3];
    for (int it = 0; it < 20; ++it) {
      const double r2 = x * x + y * y, rad = 1 + k1 * r2 + k2 * r2 * r2;
      const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x), dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
      x = (xd - dx) / rad, y = (yd - dy) / rad;
    }
    const double n = std::sqrt(x * x + y * y + 1);
    return {x / n, y / n, 1 / n};
  }
  /// EXTERNAL Camera::Triangulate(T_01, b0, b1) (call site abstract.cpp:252): midpoint of the two rays, in frame 0.
  static Vec3 Triangulate(const SE3& T_01, const Vec3& b0, const Vec3& b1) {
    const Vec3 d1 = rotate(T_01.q, b1), o = T_01.p;
    const double a = b0[0] * b0[0] + b0[1] * b0[1] + b0[2] * b0[2], b = b0[0] * d1[0] + b0[1] * d1[1] + b0[2] * d1[2];
    const double c = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
    const double e = b0[0] * o[0] + b0[1] * o[1] + b0[2] * o[2], f = d1[0] * o[0] + d1[1] * o[1] + d1[2] * o[2];
    const double den = a * c - b * b;
    const double s0 = den > 1e-12 ? (c * e - b * f) / den : 1.0, s1 = den > 1e-12 ? (b * e - a * f) / den : 1.0;
    return {0.5 * (s0 * b0[0] + o[0] + s1 * d1[0]), 0.5 * (s0 * b0[1] + o[1] + s1 * d1[1]), 0.5 * (s0 * b0[2] + o[2] + s1 * d1[2])};
  }
};

struct IMU {  // IMU entry of settings.yaml:74-109 + bias splines (imu.cpp:64-81)
  SE3 transformation;
  std::array<double, 6> gyroscope_intrinsics{1, 1, 1, 0, 0, 0}, accelerometer_intrinsics{1, 1, 1, 0, 0, 0};
  std::array<double, 9> gyroscope_sensitivity{}, accelerometer_axes_offsets{};
  int bias_order = 4;
  double bias_separation = 1.0;
};

// ---- messages (hyper/messages/**, EXTERNAL) ------------------------------------------------------------------------------
struct VisualTracks {  // one stereo frame: identifiers[i] seen at pixel P0[i] in camera 0 and P1[i] in camera 1
  Stamp stamp = 0;
  std::vector<int64_t> identifiers;
  std::vector<std::array<double, 2>> P0, P1;
};
struct InertialMeasurement {
  Stamp stamp = 0;
  std::array<double, 6> value{};  // [gyroscope; accelerometer]
};
struct ManifoldMeasurement {  // pose prior
  Stamp stamp = 0;
  std::array<double, 7> value{};
};

struct Range {
  Stamp lower = 0, upper = 0;  // LOWER_INCLUSIVE_ONLY (abstract.hpp:29-30)
  bool contains(Stamp t) const { return lower <= t && t < upper; }
  Stamp size() const { return upper - lower; }
};

struct Options {  // backend YAML keys actually read: separation, max_window, constancy flags
  Stamp separation = 0.1, max_window = 3.0;
  int order = 4;  // EXTERNAL BasisInterpolator() default is not visible; BASELINE configs use 4 and 6
  bool rotation_constant = false, translation_constant = false;
  int max_num_iterations = 5;  // optimizer.cpp:40
};

/// Everything optimize() hands to the library for one solve, as flat arrays in the C ABI's table layouts.
struct WindowTables {
  int order = 4;
  double t0 = 0, dt = 0.1;
  int rotation_constant = 0, translation_constant = 0;
  std::vector<double> cp;              // n_cp x 8
  std::vector<uint8_t> cp_constant;    // n_cp
  std::vector<double> cam_T, cam_I, cam_D;
  std::vector<double> landmarks;       // n x 3
  std::vector<double> br_stamp, br_bearing;
  std::vector<int32_t> br_lm, br_cam;
  std::vector<double> pr_stamp, pr_pose;  // pose priors against the identity sensor (ManifoldMeasurement)
  bool has_imu = false;
  double imu_T[7] = {0, 0, 0, 1, 0, 0, 0}, imu_i_g[6] = {1, 1, 1, 0, 0, 0}, imu_i_a[6] = {1, 1, 1, 0, 0, 0}, imu_S_g[9] = {0}, imu_X_a[9] = {0};
  int bias_order = 4;
  double bias_t0 = 0, bias_dt = 1;
  std::vector<double> bias_g, bias_a;  // n_bias x 4
  double gravity[3] = {0, 0, -9.80665};
  int gravity_constant = 1;
  std::vector<double> in_stamp, in_meas;

  int numControlPoints() const { return int(cp.size() / 8); }
  int numFrozen() const { return int(std::count_if(cp_constant.begin(), cp_constant.end(), [](uint8_t c) { return c != 0; })); }
  int numResidualBlocks() const { return int(br_stamp.size() + pr_stamp.size() + in_stamp.size()); }

  /// Hands the tables to a library through its C ABI entry points (template arguments: the library's functions).
  template <auto SetSpline, auto SetCameras, auto SetSensors, auto SetLandmarks, auto SetImu, auto SetGravity, auto SetBearing, auto SetPixel, auto SetPrior,
            auto SetInertial, class Check>
  void upload(hs_problem* h, Check&& check) const {
    uploadWith(h, SetSpline, SetCameras, SetSensors, SetLandmarks, SetImu, SetGravity, SetBearing, SetPixel, SetPrior, SetInertial, check);
  }
  /// Same with run-time function pointers (a library resolved with dlsym).
  template <class F1, class F2, class F3, class F4, class F5, class F6, class F7, class F8, class F9, class F10, class Check>
  void uploadWith(hs_problem* h, F1 set_spline, F2 set_cameras, F3 set_sensors, F4 set_landmarks, F5 set_imu, F6 set_gravity, F7 set_bearing, F8 set_pixel,
                  F9 set_prior, F10 set_inertial, Check&& check) const {
    check(set_spline(h, order, t0, dt, numControlPoints(), cp.data(), cp_constant.data(), rotation_constant, translation_constant), "set_spline");
    check(set_cameras(h, int(cam_T.size() / 7), cam_T.data(), cam_I.data(), cam_D.data()), "set_cameras");
    static const double identity[7] = {0, 0, 0, 1, 0, 0, 0};
    check(set_sensors(h, 1, identity), "set_sensors");
    check(set_landmarks(h, int(landmarks.size() / 3), landmarks.data(), nullptr), "set_landmarks");
    check(set_bearing(h, int(br_stamp.size()), br_stamp.data(), br_bearing.data(), br_lm.data(), br_cam.data()), "set_bearing_residuals");
    check(set_pixel(h, 0, nullptr, nullptr, nullptr, nullptr), "set_pixel_residuals");
    const std::vector<int32_t> sensor(pr_stamp.size(), 0);
    check(set_prior(h, int(pr_stamp.size()), pr_stamp.data(), pr_pose.data(), sensor.data()), "set_prior_residuals");
    if (has_imu) {
      check(set_imu(h, imu_T, imu_i_g, imu_i_a, imu_S_g, imu_X_a, bias_order, bias_t0, bias_dt, int(bias_g.size() / 4), bias_g.data(), bias_a.data(), 0), "set_imu");
      check(set_gravity(h, gravity, gravity_constant), "set_gravity");
    }
    check(set_inertial(h, int(in_stamp.size()), in_stamp.data(), in_meas.data()), "set_inertial_residuals");
  }
};

/// Mirror of `Optimizer<OptimizerSuite::CERES>` + the non-virtual logic of `AbstractOptimizer` behind the C ABI.
class Optimizer {
  struct ControlPoint {
    Stamp stamp;
    SE3 T;
    uint8_t constant = 0;
  };
  struct Observation {
    Stamp stamp;
    int32_t camera;
    Vec3 bearing;
    uint64_t seq = 0;  // order of arrival (= row order of the library's table under the delta interface)
  };
  struct Landmark {
    Vec3 position{0, 0, 0};
    std::vector<Observation> observations;
    Stamp lower = 0, upper = 0;  // AbstractLandmark::range() (landmarks/abstract.cpp:62-99)
    int32_t index = -1;          // row of the library's landmark table (delta interface)
  };
  struct BiasPoint {
    Stamp stamp;
    Vec3 g{0, 0, 0}, a{0, 0, 0};
  };

 public:
  Optimizer(const Options& options, const std::vector<Camera>& cameras, const IMU* imu = nullptr, int device = 0)
      : opt_(options), cameras_(cameras), has_imu_(imu != nullptr) {
    if (imu) imu_ = *imu;
    if (HSF(create)(device, nullptr, &handle_) != HS_OK) throw std::runtime_error("hs_create failed (no usable GPU?)");
    // HS_REPLAY_FULL_TABLES=1 (A/B and test switch): every table rebuilt and re-sent inside optimize(), as up to round 5
    if (const char* e = std::getenv("HS_REPLAY_FULL_TABLES")) delta_ = std::atoi(e) == 0;
  }
  ~Optimizer() {
    if (handle_) HSF(destroy)(handle_);
  }
  Optimizer(const Optimizer&) = delete;
  Optimizer& operator=(const Optimizer&) = delete;

  const Stamp& root() const { return root_stamp_; }
  const Range& window() const { return window_; }
  int numOptimizations() const { return num_optimizations_; }
  const hs_summary& lastSummary() const { return last_summary_; }
  size_t numLandmarks() const { return landmarks_.size(); }
  size_t numControlPoints() const { return cp_.size(); }

  /// State range = stamps for which all k control points exist (EXTERNAL AbstractState::range()): with the uniform basis of
  /// order k the segment [t_i, t_i+1) uses control points i - (k-1)/2 .. i - (k-1)/2 + k - 1, so the range runs from the stamp of
  /// control point (k-1)/2 to the stamp of control point n - k/2 (the same bound as the library's own validity check,
  /// host_structure.hpp n_seg, and as the bootstrap window [0, separation) of k control points, abstract.cpp:76-96).
  Range stateRange() const {
    const int k = opt_.order;
    return {cp_[(k - 1) / 2].stamp, cp_[cp_.size() - k / 2].stamp};
  }

  // ---- AbstractOptimizer::submit (abstract.cpp:74-147) ----
  void submit(const VisualTracks& m) { submitImpl(m.stamp, [&](Stamp s) { process(m, s); }), stage(); }
  void submit(const InertialMeasurement& m) { submitImpl(m.stamp, [&](Stamp s) { process(m, s); }), stage(); }
  void submit(const ManifoldMeasurement& m) { submitImpl(m.stamp, [&](Stamp s) { process(m, s); }), stage(); }

  /// End of a message (Backend::spin between two submits, backend.cpp:143-145): whatever the message changed is sorted and sent now, so that the
  /// optimize() a later message triggers finds the tables resident.
  void stage() {
    if (!delta_ || !staged_dirty_) return;
    const auto w0 = std::chrono::steady_clock::now();
    check(HSF(stage)(handle_), "stage");
    staged_dirty_ = false;
    wall_stage_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  }

  // ---- AbstractOptimizer::setWindow (abstract.cpp:40-62) ----
  void setWindow(const Range& window) {
    // abstract.cpp:42 reads state().range() before anything is pruned, and upstream never removes elements from the state itself
    // (updateState only removes ceres parameter blocks, optimizer.cpp:331-341): the range keeps its original lower bound. This
    // mirror drops unreachable control points from cp_, so the unpruned lower bound is remembered separately.
    const Range range{unpruned_lower_, stateRange().upper};
    window_ = window;
    updateLandmarks(window_);
    updateState(window_);
    gravity_constant_ = window_.size() < range.size();  // abstract.cpp:57-61
    if (delta_) pushState();
  }

  /// Delta interface: the variables and flags a window change touches (control points + constancy, sensors, bias points, gravity) — small
  /// tables, sent whole; the residual and landmark tables follow the messages row by row (process(), updateLandmarks(), updateState()).
  void pushState() {
    staged_dirty_ = true;
    uploadState();
    static const double identity[7] = {0, 0, 0, 1, 0, 0, 0};
    check(HSF(set_sensors)(handle_, 1, identity), "set_sensors");
    if (has_imu_) {
      const SE3& x = imu_.transformation;
      const double Tb[7] = {x.q.x, x.q.y, x.q.z, x.q.w, x.p[0], x.p[1], x.p[2]};
      std::vector<double> bg(4 * bias_.size()), ba(4 * bias_.size());
      for (size_t j = 0; j < bias_.size(); ++j)
        for (int c = 0; c < 4; ++c) bg[4 * j + c] = c < 3 ? bias_[j].g[c] : bias_[j].stamp, ba[4 * j + c] = c < 3 ? bias_[j].a[c] : bias_[j].stamp;
      check(HSF(set_imu)(handle_, Tb, imu_.gyroscope_intrinsics.data(), imu_.accelerometer_intrinsics.data(), imu_.gyroscope_sensitivity.data(),
                         imu_.accelerometer_axes_offsets.data(), imu_.bias_order, bias_.front().stamp, imu_.bias_separation, int(bias_.size()), bg.data(), ba.data(), 0),
            "set_imu");
      check(HSF(set_gravity)(handle_, gravity_.data(), gravity_constant_), "set_gravity");
    }
  }

  /// Builds the flat tables of the current window: the content upstream keeps in ceres::Problem (parameter blocks + residual
  /// blocks of the window, optimizer.cpp:189-382). `order` receives the landmark behind every row of the landmark table.
  WindowTables buildTables(std::vector<Landmark*>* order = nullptr) {
    WindowTables t;
    const int n_cp = int(cp_.size());
    t.order = opt_.order, t.t0 = cp_.front().stamp, t.dt = opt_.separation;
    t.rotation_constant = opt_.rotation_constant, t.translation_constant = opt_.translation_constant;
    t.cp.resize(size_t(8) * n_cp), t.cp_constant.resize(n_cp);
    for (int j = 0; j < n_cp; ++j) {
      const ControlPoint& c = cp_[j];
      double* o = &t.cp[8 * j];
      o[0] = c.T.q.x, o[1] = c.T.q.y, o[2] = c.T.q.z, o[3] = c.T.q.w, o[4] = c.T.p[0], o[5] = c.T.p[1], o[6] = c.T.p[2], o[7] = c.stamp;
      t.cp_constant[j] = c.constant;
    }
    t.cam_T.resize(7 * cameras_.size()), t.cam_I.resize(4 * cameras_.size()), t.cam_D.resize(4 * cameras_.size());
    for (size_t c = 0; c < cameras_.size(); ++c) {
      const SE3& x = cameras_[c].transformation;
      const double v[7] = {x.q.x, x.q.y, x.q.z, x.q.w, x.p[0], x.p[1], x.p[2]};
      std::copy(v, v + 7, &t.cam_T[7 * c]);
      std::copy(cameras_[c].intrinsics.begin(), cameras_[c].intrinsics.end(), &t.cam_I[4 * c]);
      std::copy(cameras_[c].distortion.begin(), cameras_[c].distortion.end(), &t.cam_D[4 * c]);
    }
    // landmarks + bearing residuals (the runtime uses bearing factors, abstract.cpp:243-260)
    int32_t li = 0;
    std::vector<Landmark*> rows;  // (delta interface: the library's row order = order of arrival minus the retired ones; identifier order otherwise)
    for (auto& [id, lm] : landmarks_) rows.push_back(&lm);
    if (delta_) std::sort(rows.begin(), rows.end(), [](const Landmark* a, const Landmark* b) { return a->index < b->index; });
    struct Row {
      uint64_t seq;
      int32_t lm;
      const Observation* ob;
    };
    std::vector<Row> obs;
    for (Landmark* row : rows) {
      Landmark& lm = *row;
      if (order) order->push_back(&lm);
      t.landmarks.insert(t.landmarks.end(), lm.position.begin(), lm.position.end());
      for (const Observation& ob : lm.observations) obs.push_back({ob.seq, li, &ob});
      ++li;
    }
    // residual rows: landmark by landmark, or — delta interface — in the order the library holds them (arrival), so that a harness that
    // hands these tables to a second library gives it bit for bit what the first one accumulated
    if (delta_) std::sort(obs.begin(), obs.end(), [](const Row& a, const Row& b) { return a.seq < b.seq; });
    for (const Row& r : obs) {
      t.br_stamp.push_back(r.ob->stamp), t.br_lm.push_back(r.lm), t.br_cam.push_back(r.ob->camera);
      t.br_bearing.insert(t.br_bearing.end(), r.ob->bearing.begin(), r.ob->bearing.end());
    }
    for (const ManifoldMeasurement& m : priors_) t.pr_stamp.push_back(m.stamp), t.pr_pose.insert(t.pr_pose.end(), m.value.begin(), m.value.end());
    t.has_imu = has_imu_;
    if (has_imu_) {
      for (const InertialMeasurement& m : inertials_) t.in_stamp.push_back(m.stamp), t.in_meas.insert(t.in_meas.end(), m.value.begin(), m.value.end());
      const SE3& x = imu_.transformation;
      const double Tb[7] = {x.q.x, x.q.y, x.q.z, x.q.w, x.p[0], x.p[1], x.p[2]};
      std::copy(Tb, Tb + 7, t.imu_T);
      std::copy(imu_.gyroscope_intrinsics.begin(), imu_.gyroscope_intrinsics.end(), t.imu_i_g);
      std::copy(imu_.accelerometer_intrinsics.begin(), imu_.accelerometer_intrinsics.end(), t.imu_i_a);
      std::copy(imu_.gyroscope_sensitivity.begin(), imu_.gyroscope_sensitivity.end(), t.imu_S_g);
      std::copy(imu_.accelerometer_axes_offsets.begin(), imu_.accelerometer_axes_offsets.end(), t.imu_X_a);
      t.bias_order = imu_.bias_order, t.bias_t0 = bias_.front().stamp, t.bias_dt = imu_.bias_separation;
      t.bias_g.resize(4 * bias_.size()), t.bias_a.resize(4 * bias_.size());
      for (size_t j = 0; j < bias_.size(); ++j)
        for (int c = 0; c < 4; ++c) t.bias_g[4 * j + c] = c < 3 ? bias_[j].g[c] : bias_[j].stamp, t.bias_a[4 * j + c] = c < 3 ? bias_[j].a[c] : bias_[j].stamp;
      std::copy(gravity_.begin(), gravity_.end(), t.gravity);
      t.gravity_constant = gravity_constant_;
    }
    return t;
  }

  /// CeresOptimizer::optimize (optimizer.cpp:276-280): flat tables -> hs_solve -> write back in place.
  void optimize() {
    const auto wall0 = std::chrono::steady_clock::now();
    // Delta interface (default): the tables are already resident — appended row by row in process(), retired in setWindow(), staged at the end
    // of submit() — and this function only solves and reads back. The flat tables are still built when a test harness observes the call.
    const bool hooks = bool(before_solve) || bool(after_solve);
    std::vector<Landmark*> order;
    WindowTables t;
    if (!delta_ || hooks) t = buildTables(&order);
    if (!delta_)
      t.upload<&HSF(set_spline), &HSF(set_cameras), &HSF(set_sensors), &HSF(set_landmarks), &HSF(set_imu), &HSF(set_gravity), &HSF(set_bearing_residuals),
               &HSF(set_pixel_residuals), &HSF(set_prior_residuals), &HSF(set_inertial_residuals)>(handle_, [&](int rc, const char* what) { check(rc, what); });
    if (num_observations_ == 0 && inertials_.empty() && priors_.empty()) return;  // nothing to optimise yet
    if (before_solve) before_solve(t, num_optimizations_);
    const auto wall1 = std::chrono::steady_clock::now();
    last_iterations_.assign(size_t(opt_.max_num_iterations) + 1, hs_iteration{});
    check(HSF(solve)(handle_, opt_.max_num_iterations, &last_summary_, last_iterations_.data()), "solve");
    const auto wall2 = std::chrono::steady_clock::now();
    ++num_optimizations_;
    // write back in place (the reference's solver mutates the variables through raw double*, optimizer.cpp:299-305,354-356)
    const int n_cp = int(cp_.size());
    std::vector<double>& cpv = readback_[0];
    std::vector<double>& lmv = readback_[1];
    std::vector<double>& bgv = readback_[2];
    std::vector<double>& bav = readback_[3];
    cpv.resize(size_t(8) * n_cp), lmv.resize(3 * landmarks_.size()), bgv.resize(4 * bias_.size()), bav.resize(4 * bias_.size());
    check(HSF(get_control_points)(handle_, cpv.data()), "get_control_points");
    for (int j = 0; j < n_cp; ++j) {
      const double* o = &cpv[8 * j];
      cp_[j].T = SE3{Quat{o[0], o[1], o[2], o[3]}, {o[4], o[5], o[6]}};
    }
    if (!landmarks_.empty()) check(HSF(get_landmarks)(handle_, lmv.data()), "get_landmarks");
    if (delta_) {
      for (auto& [id, lm] : landmarks_) lm.position = {lmv[3 * size_t(lm.index)], lmv[3 * size_t(lm.index) + 1], lmv[3 * size_t(lm.index) + 2]};
    } else {
      for (size_t l = 0; l < order.size(); ++l) order[l]->position = {lmv[3 * l], lmv[3 * l + 1], lmv[3 * l + 2]};
    }
    double grav[3] = {gravity_[0], gravity_[1], gravity_[2]};
    if (has_imu_) {
      check(HSF(get_bias)(handle_, bgv.data(), bav.data()), "get_bias");
      for (size_t j = 0; j < bias_.size(); ++j)
        for (int c = 0; c < 3; ++c) bias_[j].g[c] = bgv[4 * j + c], bias_[j].a[c] = bav[4 * j + c];
      check(HSF(get_gravity)(handle_, grav), "get_gravity");
      std::copy(grav, grav + 3, gravity_.begin());
    }
    const auto wall3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    wall_tables_ms_ += ms(wall0, wall1), wall_solve_ms_ += ms(wall1, wall2), wall_readback_ms_ += ms(wall2, wall3);
    if (after_solve) {
      WindowTables r = t;  // result tables: the same window at the solver's final point
      r.cp = cpv, r.landmarks = lmv;
      if (has_imu_) r.bias_g = bgv, r.bias_a = bav, std::copy(grav, grav + 3, r.gravity);
      after_solve(t, r, last_summary_, last_iterations_, num_optimizations_ - 1);
    }
  }
  /// Observers of optimize() (test harnesses only): `before_solve` sees the tables of the window right after they were handed to
  /// the library, `after_solve` additionally the same tables at the solver's final point with its summary and iteration records.
  std::function<void(const WindowTables&, int)> before_solve;
  std::function<void(const WindowTables&, const WindowTables&, const hs_summary&, const std::vector<hs_iteration>&, int)> after_solve;
  hs_problem* handle() const { return handle_; }
  /// Host wall-clock split of all optimize() calls so far: building + uploading the tables, hs_solve (structure, launches, sync), read-back.
  std::array<double, 3> wallSplitMs() const { return {wall_tables_ms_, wall_solve_ms_, wall_readback_ms_}; }
  /// Host wall-clock of all hs_stage calls (between solves, off optimize()'s clock) and whether the delta interface is in use.
  double wallStageMs() const { return wall_stage_ms_; }
  bool deltaInterface() const { return delta_; }

  /// The SIGUSR1 dump of apps/hyperslam/main.cpp:52-80: the state sampled at `rate` Hz over its range, one line per sample
  /// `stamp, qx, qy, qz, qw, px, py, pz` (scientific, 20 digits, stamp = root + sample). The samples are evaluated by the
  /// library (batched spline evaluation on the device), not on the host. Returns the number of samples written.
  int writeEstimation(const std::string& path, double rate = 100.0) {
    const int k = opt_.order, n_cp = int(cp_.size());
    std::vector<double> cp(size_t(8) * n_cp);
    std::vector<uint8_t> frozen(n_cp, 0);
    for (int j = 0; j < n_cp; ++j) {
      const ControlPoint& c = cp_[j];
      double* o = &cp[8 * j];
      o[0] = c.T.q.x, o[1] = c.T.q.y, o[2] = c.T.q.z, o[3] = c.T.q.w, o[4] = c.T.p[0], o[5] = c.T.p[1], o[6] = c.T.p[2], o[7] = c.stamp;
    }
    check(HSF(set_spline)(handle_, k, cp_.front().stamp, opt_.separation, n_cp, cp.data(), frozen.data(), opt_.rotation_constant, opt_.translation_constant),
          "set_spline");
    const Range r = stateRange();
    std::vector<double> stamps;
    for (int i = 0; r.lower + i / rate < r.upper - 1e-9; ++i) stamps.push_back(r.lower + i / rate);  // EXTERNAL Range::sample(rate)
    std::vector<double> pose(7 * stamps.size());
    if (!stamps.empty()) check(HSF(sample_trajectory)(handle_, int(stamps.size()), stamps.data(), pose.data(), nullptr, nullptr), "sample_trajectory");
    std::ofstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    f << std::scientific << std::setprecision(20);
    for (size_t i = 0; i < stamps.size(); ++i) {
      f << root_stamp_ + stamps[i];
      for (int c = 0; c < 7; ++c) f << ", " << pose[7 * i + c];
      f << "\n";
    }
    return int(stamps.size());
  }

  /// Pose of control point j (for tests / trajectory dumps).
  const SE3& controlPoint(size_t j) const { return cp_[j].T; }
  Stamp controlPointStamp(size_t j) const { return cp_[j].stamp; }
  bool controlPointConstant(size_t j) const { return cp_[j].constant; }
  void setGravity(const Vec3& g) { gravity_ = g; }
  const Vec3& gravity() const { return gravity_; }

  /// EXTERNAL AbstractState::evaluate(StateQuery{stamp}) restricted to what process(VisualTracks) needs (abstract.cpp:197-198):
  /// the value of the spline. The control points are still at their held/extrapolated values when a frame arrives, so the value is
  /// computed on the host with the same cumulative formulation (only used to place new landmarks).
  SE3 evaluate(Stamp stamp) const {
    const int k = opt_.order;
    const double x = (stamp - cp_.front().stamp) / opt_.separation;
    const int first = int(std::floor(x)) - (k - 1) / 2;
    const double u = x - std::floor(x);
    std::vector<double> lam(k);
    cumulativeBasis(k, u, lam.data());
    Quat q = cp_[first].T.q;
    Vec3 p = cp_[first].T.p;
    for (int j = 1; j < k; ++j) {
      const Quat rel = mul(conj(cp_[first + j - 1].T.q), cp_[first + j].T.q);
      q = mul(q, quatPow(rel, lam[j]));
      for (int c = 0; c < 3; ++c) p[c] += lam[j] * (cp_[first + j].T.p[c] - cp_[first + j - 1].T.p[c]);
    }
    return {q, p};
  }

 private:

  void check(int rc, const char* what) const {
    if (rc != HS_OK) throw std::runtime_error(std::string(what) + " failed: " + HSF(last_error)(handle_));
  }

  static void cumulativeBasis(int k, double u, double* lam) {  // SURVEY.md A.1 via Cox-de Boor (host, tiny)
    std::vector<double> N(k, 0.0);
    N[k - 1] = 1.0;  // order 1 on the segment; raise the order
    for (int ord = 2; ord <= k; ++ord)
      for (int j = k - ord; j < k; ++j) {
        const double jj = double(j - (k - 1));  // knot index of basis j
        const double left = (u - jj) / (ord - 1) * N[j];
        const double right = j + 1 < k ? (jj + ord - u) / (ord - 1) * N[j + 1] : 0.0;
        N[j] = left + right;
      }
    for (int j = 0; j < k; ++j) {
      double s = 0;
      for (int m = j; m < k; ++m) s += N[m];
      lam[j] = s;
    }
  }
  static Quat quatPow(Quat q, double t) {  // Exp(t Log(q)), principal branch
    if (q.w < 0) q = {-q.x, -q.y, -q.z, -q.w};
    const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (n < 1e-15) return {0, 0, 0, 1};
    const double half = std::atan2(n, q.w) * t, s = std::sin(half) / n;
    return {s * q.x, s * q.y, s * q.z, std::cos(half)};
  }

  template <class F>
  void submitImpl(Stamp raw_stamp, F&& do_process) {
    if (cp_.empty()) {  // first message: bootstrap (abstract.cpp:76-96)
      root_stamp_ = raw_stamp;
      const int k = opt_.order;
      for (int i = 0; i < k; ++i) cp_.push_back({0.0 + (i - (k - 1) / 2) * opt_.separation, SE3{}, 0});
      unpruned_lower_ = stateRange().lower;
      if (has_imu_) extendBias(Range{0, opt_.separation});
      setWindow(Range{0, opt_.separation});
    }
    const Stamp stamp = raw_stamp - root_stamp_;
    const Range state_range = stateRange();
    // state_range.contains(stamp), evaluated with the library's own segment arithmetic (uniform knots t0 + j * separation,
    // host_structure.hpp h_segment_first): the control-point stamps are accumulated sums (abstract.cpp:128) and differ from
    // t0 + j * separation in the last bits, so comparing against them could admit a stamp the spline lookup then rejects.
    const int seg = int(std::floor((stamp - cp_.front().stamp) / opt_.separation)) - (opt_.order - 1) / 2;
    if (seg >= 0 && seg < int(cp_.size()) - opt_.order + 1) {
      // window_.contains(stamp): the window's upper bound coincides with the state's by construction (both advance together,
      // abstract.cpp:139-144), so inside the state only the lower bound remains to be checked
      if (stamp >= window_.lower) do_process(stamp);
      else throw std::runtime_error("message inside the state but before the window: not implemented (abstract.cpp:109-112)");
      return;
    }
    if (seg < 0) return;  // "Discarding out-of-scope message." (abstract.cpp:116)
    (void)state_range;
    optimize();                             // abstract.cpp:119
    const Stamp delta = stamp - window_.upper;
    // abstract.cpp:124: n = ceil(delta), delta in seconds as written upstream (one new control point for any gap up to 1 s).
    // delta == 0 (a stamp exactly on the end of the state) gives n = 0 upstream, after which process() evaluates the state outside
    // its range (undefined upstream); here that single case extends by one control point instead.
    const int n_upstream = int(std::ceil(delta));
    const int n = n_upstream < 1 ? 1 : n_upstream;
    for (int i = 1; i <= n; ++i) {                       // abstract.cpp:127-137: hold the second-to-last pose
      // abstract.cpp:128 writes rbegin()->stamp() + i * separation with rbegin() re-read after every insertion: for n > 1 that
      // spaces the new stamps by 1, 2, 3 ... separations, which a uniform basis cannot represent; the knots stay uniform here
      // (identical for n = 1, the only case the 10 Hz cadence of the front-ends produces).
      const Stamp new_stamp = cp_.back().stamp + opt_.separation;
      const SE3 held = cp_[cp_.size() - 2].T;
      cp_.back().T = held;
      cp_.push_back({new_stamp, held, 0});
    }
    const Stamp x = n * opt_.separation, upper = window_.upper + x, size = window_.size();
    if (has_imu_) extendBias(Range{window_.lower, upper});
    setWindow(size + x <= opt_.max_window + 1e-12 ? Range{window_.lower, upper} : Range{upper - size, upper});  // abstract.cpp:139-143
    do_process(stamp);
  }

  // ---- process(VisualTracks) (abstract.cpp:186-264): pixel -> bearing conversion and triangulation of new tracks run in the
  //      library (hs_process_tracks, batched on the device); this function only keeps the books ----
  void process(const VisualTracks& m, Stamp stamp) {
    if (cameras_.size() != 2) throw std::runtime_error("Unsupported camera configuration.");
    const int n = int(m.identifiers.size());
    if (n == 0) return;
    if (!delta_) uploadState();  // (delta interface: the state was sent when the window last changed, pushState())
    std::vector<double> p0(2 * size_t(n)), p1(2 * size_t(n)), b0(3 * size_t(n)), b1(3 * size_t(n)), pw(3 * size_t(n));
    for (int i = 0; i < n; ++i) p0[2 * i] = m.P0[i][0], p0[2 * i + 1] = m.P0[i][1], p1[2 * i] = m.P1[i][0], p1[2 * i + 1] = m.P1[i][1];
    check(HSF(process_tracks)(handle_, stamp, n, p0.data(), p1.data(), b0.data(), b1.data(), pw.data()), "process_tracks");
    std::vector<Landmark*> fresh, seen(n);
    std::vector<double> fresh_xyz;
    for (int i = 0; i < n; ++i) {
      auto [it, inserted] = landmarks_.try_emplace(m.identifiers[i]);
      Landmark& lm = it->second;
      if (inserted) {
        lm.position = {pw[3 * i], pw[3 * i + 1], pw[3 * i + 2]};
        lm.lower = lm.upper = stamp;
        fresh.push_back(&lm), fresh_xyz.insert(fresh_xyz.end(), lm.position.begin(), lm.position.end());
      }
      lm.observations.push_back({stamp, 0, {b0[3 * i], b0[3 * i + 1], b0[3 * i + 2]}, next_seq_++});
      lm.observations.push_back({stamp, 1, {b1[3 * i], b1[3 * i + 1], b1[3 * i + 2]}, next_seq_++});
      lm.lower = std::min(lm.lower, stamp), lm.upper = std::max(lm.upper, stamp);
      seen[i] = &lm;
    }
    num_observations_ += 2 * size_t(n);
    if (delta_) {  // addLandmark + add(VisualBearingObservation&) x 2 per track (abstract.cpp:243-260)
      int32_t first = 0;
      check(HSF(append_landmarks)(handle_, int(fresh.size()), fresh_xyz.data(), nullptr, &first), "append_landmarks");
      for (size_t l = 0; l < fresh.size(); ++l) fresh[l]->index = first + int32_t(l);
      std::vector<double> st(2 * size_t(n), stamp), br(6 * size_t(n));
      std::vector<int32_t> li(2 * size_t(n)), cam(2 * size_t(n));
      for (int i = 0; i < n; ++i) {
        li[2 * i] = li[2 * i + 1] = seen[i]->index, cam[2 * i] = 0, cam[2 * i + 1] = 1;
        for (int c = 0; c < 3; ++c) br[6 * i + c] = b0[3 * i + c], br[6 * i + 3 + c] = b1[3 * i + c];
      }
      check(HSF(append_bearing_residuals)(handle_, 2 * n, st.data(), br.data(), li.data(), cam.data()), "append_bearing_residuals");
      staged_dirty_ = true;
    }
  }
  /// Control points and cameras as the library needs them for evaluation-only calls (tracks, trajectory samples).
  void uploadState() {
    const int k = opt_.order, n_cp = int(cp_.size());
    std::vector<double> cp(size_t(8) * n_cp);
    std::vector<uint8_t> frozen(n_cp);
    for (int j = 0; j < n_cp; ++j) {
      const ControlPoint& c = cp_[j];
      double* o = &cp[8 * j];
      o[0] = c.T.q.x, o[1] = c.T.q.y, o[2] = c.T.q.z, o[3] = c.T.q.w, o[4] = c.T.p[0], o[5] = c.T.p[1], o[6] = c.T.p[2], o[7] = c.stamp;
      frozen[j] = c.constant;
    }
    check(HSF(set_spline)(handle_, k, cp_.front().stamp, opt_.separation, n_cp, cp.data(), frozen.data(), opt_.rotation_constant, opt_.translation_constant),
          "set_spline");
    std::vector<double> T(7 * cameras_.size()), I(4 * cameras_.size()), D(4 * cameras_.size());
    for (size_t c = 0; c < cameras_.size(); ++c) {
      const SE3& t = cameras_[c].transformation;
      const double v[7] = {t.q.x, t.q.y, t.q.z, t.q.w, t.p[0], t.p[1], t.p[2]};
      std::copy(v, v + 7, &T[7 * c]);
      std::copy(cameras_[c].intrinsics.begin(), cameras_[c].intrinsics.end(), &I[4 * c]);
      std::copy(cameras_[c].distortion.begin(), cameras_[c].distortion.end(), &D[4 * c]);
    }
    check(HSF(set_cameras)(handle_, int(cameras_.size()), T.data(), I.data(), D.data()), "set_cameras");
  }
  void process(const InertialMeasurement& m, Stamp stamp) {  // abstract.cpp:272-292
    InertialMeasurement c = m;
    c.stamp = stamp;
    inertials_.push_back(c);
    if (delta_) check(HSF(append_inertial_residuals)(handle_, 1, &c.stamp, c.value.data()), "append_inertial_residuals"), staged_dirty_ = true;
  }
  /// abstract.cpp:266-270: a pose measurement becomes a ManifoldObservation of the (identity-extrinsics) sensor -> pose-prior
  /// residual block (optimizer.cpp:234-251).
  void process(const ManifoldMeasurement& m, Stamp stamp) {
    ManifoldMeasurement c = m;
    c.stamp = stamp;
    priors_.push_back(c);
    const int32_t sensor = 0;
    if (delta_) check(HSF(append_prior_residuals)(handle_, 1, &c.stamp, c.value.data(), &sensor), "append_prior_residuals"), staged_dirty_ = true;
  }

  // ---- CeresOptimizer::updateLandmarks (optimizer.cpp:360-382): retire landmarks whose observation range left the window ----
  void updateLandmarks(const Range& range) {
    std::vector<int32_t> retired;
    for (auto it = landmarks_.begin(); it != landmarks_.end();) {
      const bool intersects = it->second.upper >= range.lower && it->second.lower < range.upper;
      if (!intersects) retired.push_back(it->second.index), num_observations_ -= it->second.observations.size();
      it = intersects ? std::next(it) : landmarks_.erase(it);
    }
    if (delta_ && !retired.empty()) {  // RemoveParameterBlock(landmark), its residual blocks with it (optimizer.cpp:365-371)
      std::vector<int32_t> remap(landmarks_.size() + retired.size());
      check(HSF(retire_landmarks)(handle_, int(retired.size()), retired.data(), remap.data()), "retire_landmarks");
      for (auto& [id, lm] : landmarks_) lm.index = remap[size_t(lm.index)];
      staged_dirty_ = true;
    }
  }
  // ---- CeresOptimizer::updateState (optimizer.cpp:286-345): freeze control points at or before the window's lower bound;
  //      drop the ones no residual can reach any more ----
  void updateState(const Range& range) {
    for (ControlPoint& c : cp_) c.constant = c.stamp <= range.lower ? 1 : 0;  // optimizer.cpp:323-328
    // oldest stamp any retained residual refers to
    Stamp oldest = range.lower;
    for (const auto& [id, lm] : landmarks_) oldest = std::min(oldest, lm.lower);
    if (has_imu_) {
      // Ceres never removes inertial / prior residual blocks (only landmark removal takes residual blocks along, optimizer.cpp:365-371),
      // so upstream the problem grows for as long as the IMU runs. Retirement rule here: residuals older than the window and than every
      // retained landmark go — the same kind of information loss as landmark retirement — which keeps the tables bounded.
      inertials_.erase(std::remove_if(inertials_.begin(), inertials_.end(), [&](const InertialMeasurement& m) { return m.stamp < oldest; }),
                       inertials_.end());
    }
    priors_.erase(std::remove_if(priors_.begin(), priors_.end(), [&](const ManifoldMeasurement& m) { return m.stamp < oldest; }), priors_.end());
    if (delta_) {
      if (has_imu_) check(HSF(retire_residuals_before)(handle_, HS_INERTIAL, oldest), "retire_residuals_before");
      check(HSF(retire_residuals_before)(handle_, HS_PRIOR, oldest), "retire_residuals_before");
      staged_dirty_ = true;
    }
    if (has_imu_ && !inertials_.empty() && !bias_.empty()) {
      // bias points no retained inertial residual reads are not part of the problem (Ceres only holds parameter blocks some residual
      // block refers to, exteroceptive.cpp:64-76): drop them from the front so that the border of the reduced system stays as wide
      // as the window, not as long as the run. Same arithmetic as the library's segment lookup (floor((t - t0) / dt) - (kb - 1) / 2).
      Stamp t_min = inertials_.front().stamp;
      for (const InertialMeasurement& m : inertials_) t_min = std::min(t_min, m.stamp);
      const int kb = imu_.bias_order;
      while (int(bias_.size()) > kb && int(std::floor((t_min - bias_[1].stamp) / imu_.bias_separation)) - (kb - 1) / 2 >= 0) bias_.erase(bias_.begin());
    }
    const int k = opt_.order;
    size_t drop = 0;  // control points entirely before the segment of `oldest` (optimizer.cpp:331-341); the margin keeps a stamp that
                      // sits exactly on a knot inside the valid range whatever the rounding of (stamp - t0) / separation
    while (drop + k < cp_.size() && cp_[drop + (k - 1) / 2 + 1].stamp <= oldest - 1e-6 * opt_.separation) ++drop;
    if (drop) cp_.erase(cp_.begin(), cp_.begin() + drop);
  }
  // ---- updateSensor(IMU&, Range) is CHECK(false) upstream (optimizer.cpp:384-386); here: keep the bias splines covering the range ----
  void extendBias(const Range& range) {
    const int kb = imu_.bias_order;
    const double sep = imu_.bias_separation;
    if (bias_.empty()) {
      const Stamp first = std::floor(range.lower / sep) * sep - ((kb - 1) / 2) * sep;
      for (int j = 0; j < kb; ++j) bias_.push_back({first + j * sep});
    }
    while (bias_[bias_.size() - 1 - kb / 2].stamp <= range.upper + opt_.separation) {
      BiasPoint b = bias_.back();
      b.stamp += sep;
      bias_.push_back(b);
    }
  }

  Options opt_;
  std::vector<Camera> cameras_;
  bool has_imu_ = false;
  IMU imu_;
  hs_problem* handle_ = nullptr;
  Stamp root_stamp_ = 0;
  Range window_;
  Stamp unpruned_lower_ = 0;  // lower bound of state().range() upstream (elements are never removed from the state there)
  bool gravity_constant_ = false;
  Vec3 gravity_{0, 0, -9.80665};
  std::vector<ControlPoint> cp_;
  std::map<int64_t, Landmark> landmarks_;
  std::vector<InertialMeasurement> inertials_;
  std::vector<ManifoldMeasurement> priors_;
  std::vector<BiasPoint> bias_;
  hs_summary last_summary_{};
  std::vector<hs_iteration> last_iterations_;
  int num_optimizations_ = 0;
  double wall_tables_ms_ = 0, wall_solve_ms_ = 0, wall_readback_ms_ = 0, wall_stage_ms_ = 0;
  bool delta_ = true;             // tables kept in the library through hs_append_* / hs_retire_* / hs_stage (false: rebuilt inside optimize())
  bool staged_dirty_ = false;     // rows were appended / retired since the last hs_stage
  size_t num_observations_ = 0;   // visual residual blocks of the window
  uint64_t next_seq_ = 0;
  std::vector<double> readback_[4];
};

}  // namespace hyper_hip
