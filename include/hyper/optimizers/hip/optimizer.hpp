/// Optimizer<kOptimizerSuiteHIP> ("suite: hip") — the reference-side plugin that puts libhyperslam_hip.so behind HyperSLAM's optimizer interface.
///
/// WHERE THIS FILE LIVES. It is written against the HyperSLAM tree (Eigen, glog, yaml-cpp, HyperVariables / HyperSensors / HyperState),
/// none of which exist in this repository's image: it is shipped for the maintainer who adds the backend and is not part of this
/// repository's build. What this repository tests: (i) the header is type-checked (`g++ -std=c++20 -fsyntax-only`) against the
/// reference's own in-tree headers (AbstractOptimizer, Environment, observations, landmarks) with hand-written declaration-only
/// stand-ins for the EXTERNAL ones (tests/stubs/, tests/test_plugin_header.py; build container only), so every override, signature and
/// include is checked by a compiler; (ii) everything underneath it — the C ABI it calls (include/hyperslam_hip.h: syntax, exported
/// symbol set, behaviour); (iii) the same window logic restated without the external dependencies (hyperslam_amd/host/optimizer.hpp,
/// tests/test_host_driver.py), which runs against both libraries.
///
/// What a maintainer changes upstream (two places, everything else — front-end, Backend::spin, AbstractOptimizer::submit / setWindow /
/// process, the YAML keys — stays as it is):
///   1. internal/hyper/system/components/backend.cpp:37    `} else if (suite == "hip") {` branch, see make_hip_optimizer() at the end
///   2. settings.yaml:137                                  suite: hip
/// and adds this header + include/hyperslam_hip.h to the include path and -lhyperslam_hip to the link line. The enum of
/// include/hyper/optimizers/forward.hpp:14-17 needs no edit: see kOptimizerSuiteHIP below.
///
/// How it maps onto the Ceres backend it replaces (internal/hyper/optimizers/ceres/optimizer.cpp, "cc" below):
///   Ceres keeps the problem structure incrementally (AddParameterBlock / AddResidualBlock / RemoveParameterBlock, cc:286-382) and
///   mutates the variables in place through the registered double* (cc:299-305,354-356). So does this plugin, through the library's DELTA
///   interface (hyperslam_hip.h: hs_append_* / hs_retire_* / hs_stage): add(observation) appends one row to the resident residual table,
///   addLandmark one row to the landmark table, updateLandmarks retires rows (the landmark's residual rows with it), and optimize() sends
///   what a window change touches — control points with their constancy mask, the bias points in use, gravity — solves and WRITES the result
///   back into the same variables, so that every caller above (abstract.cpp:74-147) observes exactly the side effects it observes with
///   Ceres. The residual and landmark tables are sorted and uploaded between solves (hs_stage from add(InertialObservation&) /
///   add(ManifoldObservation&): the last call of the messages that arrive one observation at a time; a stereo frame's rows are staged by the
///   next of those, or by hs_solve itself): optimize() finds them resident, as ceres::Solve finds its problem built.
///     parameter blocks   state elements in variables_ (cc:296-306)      -> hs_set_spline rows [q(4) p(3) t], constancy mask (cc:319-328)
///                        sensor.parameters() (cc:143-155)               -> hs_set_cameras / hs_set_sensors / hs_set_imu (all constant, camera.hpp:18, imu.hpp:18)
///                        imu bias elements (imu.cpp:64-81)              -> hs_set_imu bias tables (variable, cc:65-66)
///                        landmarks_ (cc:347-358)                        -> hs_set_landmarks, dense id = position in the table
///                        gravity (cc:84-108,130-141)                    -> hs_set_gravity
///     residual blocks    add(VisualBearingObservation&) cc:189-210      -> hs_set_bearing_residuals  (AngularMetric, Huber 1.6e-3)
///                        add(VisualPixelObservation&)   cc:212-232      -> hs_set_pixel_residuals    (CartesianMetric, Huber 0.5)
///                        add(ManifoldObservation&)      cc:234-251      -> hs_set_prior_residuals    (ManifoldMetric, no loss)
///                        add(InertialObservation&)      cc:253-274      -> hs_set_inertial_residuals (CartesianMetric<6>, Scaled 1.6e-5)
///     removal            RemoveParameterBlock(landmark) drops its residual blocks (cc:365-371, enable_fast_removal) -> observation lists are
///                        pruned with the landmark in updateLandmarks(); state elements without residuals leave variables_ (cc:330-341)
///     bias splines       updateSensor(imu, range) — CHECK(false) upstream (cc:384-386) although abstract.cpp:278-286 relies on it to create
///                        and extend imu.gyroscopeBias() / accelerometerBias() — is implemented here (extendBias)
///     bounded window     Ceres never removes inertial / prior residual blocks, so upstream every control point an IMU sample ever touched
///                        stays a (constant) parameter block for ever; here residuals older than the window and every retained landmark are
///                        retired in updateLandmarks() (the same kind of information loss as landmark retirement, cc:365-371), which keeps
///                        variables_ and the tables bounded (the library holds at most 1024 control points)
///     solve              ceres::Solve(kDefaultSolverOptions) cc:38-54,276-280 -> hs_solve(handle, 5, ...)
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <iterator>
#include <limits>
#include <memory>
#include <typeindex>
#include <typeinfo>
#include <unordered_map>
#include <utility>
#include <vector>

#include <glog/logging.h>
#include <yaml-cpp/yaml.h>

#include "hyper/environment/observations/inertial.hpp"
#include "hyper/environment/observations/manifold.hpp"
#include "hyper/environment/observations/visual.hpp"
#include "hyper/optimizers/abstract.hpp"
#include "hyper/sensors/camera.hpp"
#include "hyper/sensors/imu.hpp"
#include "hyper/state/interpolators/abstract.hpp"
#include "hyper/yaml/yaml.hpp"

#include "hyperslam_hip.h"

namespace hyper {

/// Key of the specialisation. The reference's `enum class OptimizerSuite { CERES, DEFAULT = CERES }` (forward.hpp:14-17) has no HIP
/// enumerator; a scoped enum admits every value of its underlying type, so the key is a constant of the enum type and forward.hpp
/// stays untouched (a tree that adds `HIP = 1` to the enum names the same specialisation).
inline constexpr auto kOptimizerSuiteHIP = static_cast<OptimizerSuite>(1);

template <>
class Optimizer<kOptimizerSuiteHIP> final : public AbstractOptimizer {
 public:
  /// Same signature as Optimizer<CERES> (ceres/optimizer.hpp:28; backend.cpp:46). `device` / the in-tree solver options of cc:38-54
  /// (max_num_iterations = 5) are fixed here as they are fixed there.
  explicit Optimizer(const YAML::Node& yaml_node = {}, const std::vector<Sensor*>& sensors = {}) : AbstractOptimizer{yaml_node} {
    CHECK_EQ(hs_create(/*device=*/0, /*stream=*/nullptr, &handle_), HS_OK) << "no usable gfx950 device";
    if (!yaml_node.IsNull())
      for (auto* sensor : sensors) {  // cc:76-83 setSensorManifold(createSensorManifold(...))
        DCHECK(sensor != nullptr);
        addSensor(*sensor);
      }
  }
  ~Optimizer() final { hs_destroy(handle_); }

  /// cc:84-108. The gravity block is read from the environment at optimize(); nothing to register.
  auto swapEnvironment(std::unique_ptr<Environment<Manifold>>& environment) -> void final { std::swap(environment_, environment); }

  /// cc:110-128.
  auto swapState(std::unique_ptr<AbstractState>& state) -> void final {
    variables_.clear();
    std::swap(state_, state);
    updateState(window_);
  }

  /// Constancy flags of Manifold<Stamped<SE3>, CERES>{time_constant, rotation_constant, translation_constant} (backend.cpp:52-55,
  /// setStateManifold cc:168-182). Time is always constant in the library (stamped.hpp:35-36 with time_constant = true, settings.yaml).
  auto setStateConstancy(const bool rotation_constant, const bool translation_constant) -> void {
    rotation_constant_ = rotation_constant;
    translation_constant_ = translation_constant;
  }

  /// cc:189-232. A residual block keeps its landmark's parameter block alive in Ceres (AddResidualBlock registers an unknown block): an
  /// observation of a landmark that was retired earlier (addLandmark is only called for NEW landmarks, abstract.cpp:252-257) makes
  /// it active again, like there.
  auto add(VisualBearingObservation& observation) -> void final {
    landmarks_.insert(&observation.landmark());
    bearings_.push_back(&observation);
    const auto& m = observation.measurement();
    const double stamp = admitted(m.stamp());
    const std::int32_t row = landmarkRow(observation.landmark()), camera = camera_index_.at(&m.sensor());
    check(hs_append_bearing_residuals(handle_, 1, &stamp, m.variable().asVector().data(), &row, &camera));
  }
  auto add(VisualPixelObservation& observation) -> void final {
    landmarks_.insert(&observation.landmark());
    pixels_.push_back(&observation);
    const auto& m = observation.measurement();
    const double stamp = admitted(m.stamp());
    const std::int32_t row = landmarkRow(observation.landmark()), camera = camera_index_.at(&m.sensor());
    check(hs_append_pixel_residuals(handle_, 1, &stamp, m.variable().asVector().data(), &row, &camera));
  }
  auto add(ManifoldObservation<Manifold>& observation) -> void final {  // cc:234-251
    priors_.push_back(&observation);
    const auto& m = observation.measurement();
    const double stamp = admitted(m.stamp());
    const std::int32_t sensor = pose_sensor_index_.at(&m.sensor());
    check(hs_append_prior_residuals(handle_, 1, &stamp, m.variable().asVector().data(), &sensor));  // SE3 [q(4) p(3)]
    stage();
  }
  auto add(InertialObservation<Manifold>& observation) -> void final {  // cc:253-274
    inertials_.push_back(&observation);
    const auto& m = observation.measurement();
    const double stamp = admitted(m.stamp());
    check(hs_append_inertial_residuals(handle_, 1, &stamp, m.variable().asVector().data()));  // Tangent<SE3> [angular(3) linear(3)]
    stage();
  }

  [[nodiscard]] auto hasSensor(const Sensor& sensor) const -> bool final {  // cc:157-159
    return camera_index_.contains(&sensor) || pose_sensor_index_.contains(&sensor) || imu_ == &sensor;
  }

  auto setGravityConstant(const bool set_constant) -> void final { gravity_constant_ = set_constant; }  // cc:130-141

  /// Inertial Jacobian: HS_INERTIAL_AS_REFERENCE reproduces evaluators/inertial.cpp:131-198 as written (default), HS_INERTIAL_EXACT the
  /// derivative of the prediction (see include/hyperslam_hip.h).
  auto setInertialJacobian(const int mode) -> void { CHECK_EQ(hs_set_inertial_jacobian(handle_, mode), HS_OK) << hs_last_error(handle_); }

  /// Retirement of inertial / pose-prior observations older than the window and every retained landmark (updateLandmarks). DEVIATION from
  /// the Ceres backend, which never removes those residual blocks (and so never lets their control points go): default on, because the
  /// library holds a bounded window (<= 1024 control points); off reproduces upstream's unbounded growth for as long as the tables fit.
  /// Pose priors on control points that are still free are kept either way (they may be what fixes the gauge).
  auto setRetireOldObservations(const bool retire) -> void { retire_old_observations_ = retire; }

  /// Windows optimize() could not hand to the library because the control-point stamps between the oldest and the newest parameter block are
  /// not uniform (HS_ERR_KNOTS; upstream's multi-state extension, abstract.cpp:127-137). 0 on a front-end that delivers at its own cadence.
  [[nodiscard]] auto skippedWindows() const -> std::size_t { return skipped_windows_; }

  /// Knot spacing of the two bias splines updateSensor() creates (the reference has no YAML key for it; its test uses 10 state
  /// separations, tests/internal/tests/optimizers/evaluators/inertial.cpp:48-49). Takes effect for splines that are still empty.
  auto setBiasSeparation(const Stamp separation) -> void {
    CHECK_GT(separation, 0);
    bias_separation_ = separation;
  }

  /// cc:276-280. What a window change touches (control points, sensors, bias points in use, gravity) -> hs_solve -> write-back in place; the
  /// residual and landmark tables are resident (add / addLandmark / updateLandmarks).
  auto optimize() -> void final {
    if (bearings_.empty() && pixels_.empty() && priors_.empty() && inertials_.empty()) return;
    const auto order = state().interpolator()->layout().outer.size;  // control points per segment (k)

    // ---- control points: the state elements Ceres holds parameter blocks for (variables_, cc:296-306), in stamp order. The library's
    //      basis is uniform — control point j sits at t0 + j * separation and hs_set_spline rejects a table whose stamps say otherwise —
    //      so the table is the CONTIGUOUS run of state elements between the oldest and the newest parameter block. updateState() only
    //      ever drops a prefix / suffix of that run; should an element in between not be a parameter block (it is not in variables_), it
    //      rides along as a constant row instead of silently re-indexing every later control point. ----
    CHECK(!variables_.empty());
    auto oldest_stamp = std::numeric_limits<Stamp>::max(), newest_stamp = std::numeric_limits<Stamp>::lowest();
    for (const auto* variable : variables_) oldest_stamp = std::min(oldest_stamp, variable->stamp()), newest_stamp = std::max(newest_stamp, variable->stamp());
    std::vector<StampedManifold*> cps;
    std::vector<std::uint8_t> cp_constant;
    const auto& elements = state().elements();
    for (auto itr = elements.lower_bound(oldest_stamp); itr != elements.end() && (*itr)->stamp() <= newest_stamp; ++itr) {
      cps.push_back(static_cast<StampedManifold*>(itr->get()));
      const auto stamp = (*itr)->stamp();
      cp_constant.push_back((!variables_.contains(itr->get()) || stamp <= window_.lowerBound() || newest_stamp < stamp) ? 1 : 0);  // cc:319-328
    }
    CHECK_GE(cps.size(), static_cast<std::size_t>(order));
    std::vector<double> cp(8 * cps.size());
    for (std::size_t j = 0; j < cps.size(); ++j) std::copy_n(cps[j]->asVector().data(), 8, &cp[8 * j]);  // [q(4) p(3) t], stamped.hpp:35-36
    const auto t0 = cps.front()->stamp();
    // ONE refusal is survivable: HS_ERR_KNOTS. Upstream's extension by more than one state (abstract.cpp:127-137 computes the new stamps as
    // rbegin()->stamp() + i * separation with rbegin() re-read after every insertion, i.e. spaced by 1, 2, 3 ... separations) leaves a
    // PERMANENT hole in the knots, which no uniform basis can represent: every window whose table spans the hole is skipped — the state stays
    // as it is, estimation resumes once the hole has slid out of the window — and counted (skippedWindows()), so that the caller can see
    // that it happened instead of reading it out of the log. Every other failure of hs_set_spline (order, sizes, null table) is a bug of
    // this file and aborts through check(), like everywhere else.
    if (const auto rc = hs_set_spline(handle_, order, t0, separation_, static_cast<int>(cps.size()), cp.data(), cp_constant.data(), rotation_constant_, translation_constant_);
        rc == HS_ERR_KNOTS) {
      ++skipped_windows_;
      LOG(ERROR) << "hip optimizer: window " << window_.lowerBound() << " .. " << window_.upperBound() << " skipped (" << skipped_windows_
                 << " so far): " << hs_last_error(handle_);
      return;
    } else {
      check(rc);
    }
    // ---- sensors: sensor.parameters() in Traits order (cc:143-155); constant blocks (camera.hpp:18, imu.hpp:18) ----
    std::vector<double> cam_T(7 * cameras_.size()), cam_i(4 * cameras_.size()), cam_d(4 * cameras_.size());
    for (std::size_t c = 0; c < cameras_.size(); ++c) {
      const auto parameters = cameras_[c]->parameters();  // {transformation, intrinsics [cx cy fx fy], distortion [k1 k2 p1 p2]}
      std::copy_n(parameters[Traits<Camera>::kTransformationOffset]->asVector().data(), 7, &cam_T[7 * c]);
      std::copy_n(parameters[Traits<Camera>::kIntrinsicsOffset]->asVector().data(), 4, &cam_i[4 * c]);
      std::copy_n(parameters[Traits<Camera>::kDistortionOffset]->asVector().data(), 4, &cam_d[4 * c]);
    }
    check(hs_set_cameras(handle_, static_cast<int>(cameras_.size()), cam_T.data(), cam_i.data(), cam_d.data()));
    std::vector<double> sensor_T(7 * pose_sensors_.size());
    for (std::size_t s = 0; s < pose_sensors_.size(); ++s) std::copy_n(pose_sensors_[s]->parameters()[0]->asVector().data(), 7, &sensor_T[7 * s]);
    check(hs_set_sensors(handle_, static_cast<int>(pose_sensors_.size()), sensor_T.data()));

    // ---- landmarks and residual tables: resident in the library, row by row (addLandmark / add(...) -> hs_append_*, updateLandmarks ->
    //      hs_retire_*). Nothing to send: the library's landmark values are those its last solve left, which are the variables' (write-back). ----

    // ---- IMU: static blocks {T_bs, i_g, i_a, S_g, X_a} (inertial.cpp:36-49) + the two R^3 bias splines (imu.cpp:64-81) + gravity.
    //      Ceres only sees the bias elements some residual block refers to (exteroceptive.cpp:64-76); the table holds exactly that range:
    //      elements [bias_first, bias_first + n_bias) of both splines. ----
    std::vector<Traits<IMU>::GyroscopeBias*> bias_g_elements;
    std::vector<Traits<IMU>::AccelerometerBias*> bias_a_elements;
    if (imu_ != nullptr && !inertials_.empty()) {
      const auto parameters = imu_->parameters();
      for (const auto& element : imu_->gyroscopeBias().elements()) bias_g_elements.push_back(static_cast<Traits<IMU>::GyroscopeBias*>(element.get()));
      for (const auto& element : imu_->accelerometerBias().elements()) bias_a_elements.push_back(static_cast<Traits<IMU>::AccelerometerBias*>(element.get()));
      CHECK_EQ(bias_g_elements.size(), bias_a_elements.size());  // both splines are created and extended together (updateSensor)
      const auto bias_layout = imu_->gyroscopeBias().interpolator()->layout();
      const auto bias_order = bias_layout.outer.size;
      CHECK_GE(bias_g_elements.size(), static_cast<std::size_t>(bias_order));
      const auto bias_dt = bias_g_elements[1]->stamp() - bias_g_elements[0]->stamp();
      auto [oldest, newest] = std::minmax_element(inertials_.begin(), inertials_.end(),
                                                  [](const auto* a, const auto* b) { return a->measurement().stamp() < b->measurement().stamp(); });
      // first element a residual at `stamp` reads, counted from element `from`, with the library's own arithmetic (hyperslam_hip.h:
      // control point j at bias_t0 + j * bias_dt, segment floor((stamp - t0) / dt), elements segment - (order - 1) / 2 ... + order - 1)
      const auto first_read = [&](const Stamp stamp, const std::size_t from) {
        return static_cast<std::ptrdiff_t>(std::floor((stamp - bias_g_elements[from]->stamp()) / bias_dt)) - (bias_order - 1) / 2;
      };
      auto bias_first = static_cast<std::size_t>(std::max<std::ptrdiff_t>(0, first_read((*oldest)->measurement().stamp(), 0)));
      while (bias_first > 0 && first_read((*oldest)->measurement().stamp(), bias_first) < 0) --bias_first;  // a stamp within rounding of a knot
      const auto bias_end = std::min<std::ptrdiff_t>(static_cast<std::ptrdiff_t>(bias_g_elements.size()),
                                                     static_cast<std::ptrdiff_t>(bias_first) + first_read((*newest)->measurement().stamp(), bias_first) + bias_order + 1);
      bias_g_elements = {bias_g_elements.begin() + static_cast<std::ptrdiff_t>(bias_first), bias_g_elements.begin() + bias_end};
      bias_a_elements = {bias_a_elements.begin() + static_cast<std::ptrdiff_t>(bias_first), bias_a_elements.begin() + bias_end};
      const auto n_bias = bias_g_elements.size();
      std::vector<double> bias_g(4 * n_bias), bias_a(4 * n_bias);
      for (std::size_t j = 0; j < n_bias; ++j) {
        std::copy_n(bias_g_elements[j]->asVector().data(), 4, &bias_g[4 * j]);  // Stamped<R3> [b(3) t]
        std::copy_n(bias_a_elements[j]->asVector().data(), 4, &bias_a[4 * j]);
      }
      check(hs_set_imu(handle_, parameters[0]->asVector().data(), parameters[1]->asVector().data(), parameters[2]->asVector().data(),
                       parameters[3]->asVector().data(), parameters[4]->asVector().data(), bias_order, bias_g[3], bias_dt, static_cast<int>(n_bias), bias_g.data(),
                       bias_a.data(), /*bias_constant=*/0));  // cc:62-63 set*BiasConstant(false)
      check(hs_set_gravity(handle_, mutableEnvironment().gravity().data(), gravity_constant_ ? 1 : 0));
      // (the inertial rows themselves: appended by add(InertialObservation&), retired in updateLandmarks(). When every sample has left the window the
      //  bias and gravity tables the handle still holds are unknowns without residuals and come back untouched,
      //  tests/test_gpu_inertial.py::test_imu_tables_without_inertial_residuals)
    }

    // ---- solve (cc:38-54: trust-region LM, 5 iterations, monotonic steps) ----
    hs_summary summary;
    check(hs_solve(handle_, /*max_iterations=*/5, &summary, nullptr));
    LOG(INFO) << "hip: " << summary.num_iterations << " iterations, cost " << summary.initial_cost << " -> " << summary.final_cost;  // cc:279

    // ---- write back in place: the variables own the memory, the solver's result must be visible through them (cc:299-305,354-356) ----
    check(hs_get_control_points(handle_, cp.data()));
    for (std::size_t j = 0; j < cps.size(); ++j)
      if (!cp_constant[j]) std::copy_n(&cp[8 * j], 7, cps[j]->asVector().data());  // stamp untouched (time is constant)
    if (!landmark_rows_.empty()) {
      std::vector<double> lm(3 * landmark_rows_.size());
      check(hs_get_landmarks(handle_, lm.data()));
      for (std::size_t l = 0; l < landmark_rows_.size(); ++l) landmark_rows_[l]->variable() = Position<Scalar>{lm[3 * l], lm[3 * l + 1], lm[3 * l + 2]};
    }
    if (!bias_g_elements.empty()) {
      std::vector<double> bias_g(4 * bias_g_elements.size()), bias_a(4 * bias_a_elements.size());
      check(hs_get_bias(handle_, bias_g.data(), bias_a.data()));
      for (std::size_t j = 0; j < bias_g_elements.size(); ++j) {
        std::copy_n(&bias_g[4 * j], 3, bias_g_elements[j]->asVector().data());
        std::copy_n(&bias_a[4 * j], 3, bias_a_elements[j]->asVector().data());
      }
      check(hs_get_gravity(handle_, mutableEnvironment().gravity().data()));
    }
  }

 private:
  auto check(const int rc) const -> void { CHECK_EQ(rc, HS_OK) << hs_last_error(handle_); }  // the reference aborts through glog CHECK

  /// Sorts and uploads what the messages since the last solve changed (hs_stage: enqueued, nothing is awaited) — from the add() overrides of the
  /// messages that bring ONE observation (inertial sample, pose measurement), i.e. at the end of AbstractOptimizer::process; a stereo frame's
  /// rows (process(VisualTracks) calls add() once per observation, abstract.cpp:246-259, and has no last call to hang this on) are staged with
  /// the next of those, or by hs_solve. A maintainer who prefers one explicit call adds `optimizer_->stage()` behind `submit` in Backend::spin
  /// (backend.cpp:143-145) and makes this public.
  auto stage() -> void { check(hs_stage(handle_)); }

  /// Upstream admits a message with state().range().contains(stamp) on the elements' ACCUMULATED stamps (abstract.cpp:103-106, 127-137);
  /// the library derives the segment from t0 + j * separation. The two agree except in the last bits of a stamp on a knot: a stamp that
  /// upstream admitted and the uniform arithmetic puts one segment outside the state is moved by those last bits (evaluated at admission,
  /// against the state as it is then — the state only grows at its newer end afterwards).
  [[nodiscard]] auto admitted(Stamp stamp) const -> Stamp {
    const auto& elements = state().elements();
    if (elements.empty()) return stamp;
    const auto order = static_cast<std::ptrdiff_t>(state().interpolator()->layout().outer.size);
    const auto t0 = (*elements.begin())->stamp(), newest = (*elements.rbegin())->stamp();
    const auto n_segments = static_cast<std::ptrdiff_t>(elements.size()) - order + 1;
    const auto segment = [&](const Stamp s) { return static_cast<std::ptrdiff_t>(std::floor((s - t0) / separation_)) - (order - 1) / 2; };
    for (auto i = 0; i < 4 && segment(stamp) >= n_segments && stamp - newest < 1e-9 * separation_; ++i) stamp = std::nextafter(stamp, std::numeric_limits<Stamp>::lowest());
    for (auto i = 0; i < 4 && segment(stamp) < 0 && t0 - stamp < 1e-9 * separation_; ++i) stamp = std::nextafter(stamp, std::numeric_limits<Stamp>::max());
    return stamp;
  }

  /// Row of a landmark in the library's table; a landmark the library does not hold (new: addLandmark; retired earlier and observed again:
  /// AddResidualBlock registers an unknown parameter block, cc:203-209) is appended with its present value.
  auto landmarkRow(AbstractLandmark& abstract_landmark) -> std::int32_t {
    if (const auto itr = landmark_row_.find(&abstract_landmark); itr != landmark_row_.end()) return itr->second;
    auto* landmark = static_cast<Landmark<Position<Scalar>>*>(&abstract_landmark);
    const auto& p = landmark->variable();
    const double xyz[3] = {p.x(), p.y(), p.z()};
    std::int32_t row = -1;
    check(hs_append_landmarks(handle_, 1, xyz, /*constant=*/nullptr, &row));
    CHECK_EQ(static_cast<std::size_t>(row), landmark_rows_.size());
    landmark_row_.emplace(&abstract_landmark, row);
    landmark_rows_.push_back(landmark);
    return row;
  }

  /// createSensorManifold + setSensorManifold (cc:56-71,143-155): cameras by pointer -> index into the camera table, the IMU, and
  /// any other pose sensor (ManifoldObservation's sensor) -> index into the extrinsics table.
  auto addSensor(Sensor& sensor) -> void {
    const auto type = std::type_index{typeid(sensor)};
    if (type == std::type_index{typeid(Camera)}) {
      camera_index_.emplace(&sensor, static_cast<std::int32_t>(cameras_.size()));
      cameras_.push_back(&sensor.as<Camera>());
    } else if (type == std::type_index{typeid(IMU)}) {
      CHECK(imu_ == nullptr) << "one IMU (the inertial factor reads a single bias spline pair)";
      imu_ = &sensor.as<IMU>();
    } else {
      pose_sensor_index_.emplace(&sensor, static_cast<std::int32_t>(pose_sensors_.size()));
      pose_sensors_.push_back(&sensor);
    }
  }

  /// cc:286-345 without the Ceres calls: variables_ is the set of state elements that are parameter blocks. New elements of the padded
  /// range join; elements outside it leave once no residual block touches them (cc:330-341). Constancy is decided at optimize().
  /// AbstractOptimizer::setWindow runs updateLandmarks() first (abstract.cpp:52-56), so the observation lists are already pruned.
  auto updateState(const Range& range) -> void final {
    const auto& elements = state().elements();
    const auto [left_padding, right_padding] = state().interpolator()->layout().outerPadding();
    const auto begin = std::prev(elements.upper_bound(range.lowerBound()), left_padding);
    const auto end = std::next(elements.upper_bound(range.upperBound()), right_padding);
    const auto stamp_0 = (*begin)->stamp();
    const auto stamp_n = (*std::prev(end))->stamp();
    for (auto itr = begin; itr != end; ++itr) variables_.insert(itr->get());
    // residual stamps still present, sorted: a control point is in use if a residual's k-neighbourhood contains it
    std::vector<Stamp> stamps;
    for (const auto* o : bearings_) stamps.push_back(o->measurement().stamp());
    for (const auto* o : pixels_) stamps.push_back(o->measurement().stamp());
    for (const auto* o : priors_) stamps.push_back(o->measurement().stamp());
    for (const auto* o : inertials_) stamps.push_back(o->measurement().stamp());
    std::sort(stamps.begin(), stamps.end());
    // A residual at t in [t_i, t_i+1) reads the control points i - left_padding ... i + right_padding, i.e. those with stamps in
    // (t - (left_padding + 1) dt, t + right_padding dt]. One separation of slack on the old side: a control point that stays one window
    // longer is a constant block without residuals (harmless), one that leaves too early would fail the library's range check.
    const auto ahead = separation_ * right_padding;
    const auto behind = separation_ * (left_padding + 2);
    const auto unused = [&](const auto* variable) {
      const auto stamp = variable->stamp();
      if (!(stamp < stamp_0 || stamp_n < stamp)) return false;
      const auto itr = std::lower_bound(stamps.begin(), stamps.end(), stamp - ahead);
      return itr == stamps.end() || *itr >= stamp + behind;  // no residual block left on it (cc:331-341)
    };
    // CONTIGUITY RULE: only a prefix and a suffix (in stamp order) leave. Ceres removes any parameter block without residuals (cc:336-341);
    // the library indexes control points uniformly, so an unused element between two used ones stays (optimize() hands it over as a
    // constant row) and leaves once everything older than it has left.
    std::vector<AbstractStamped<Scalar>*> by_stamp(variables_.begin(), variables_.end());
    std::sort(by_stamp.begin(), by_stamp.end(), [](const auto* a, const auto* b) { return a->stamp() < b->stamp(); });
    auto first_kept = by_stamp.begin(), last_kept = by_stamp.end();
    while (first_kept != last_kept && unused(*first_kept)) ++first_kept;
    while (last_kept != first_kept && unused(*std::prev(last_kept))) --last_kept;
    for (auto itr = by_stamp.begin(); itr != first_kept; ++itr) variables_.erase(*itr);
    for (auto itr = last_kept; itr != by_stamp.end(); ++itr) variables_.erase(*itr);
  }

  auto addLandmark(Landmark<Position<Scalar>>& landmark) -> void final {  // cc:347-358
    DCHECK(!landmarks_.contains(&landmark));
    landmarks_.insert(&landmark);
    (void)landmarkRow(landmark);
  }

  /// cc:360-382: landmarks whose observation range left the window are retired, and RemoveParameterBlock takes their residual blocks
  /// along (enable_fast_removal, cc:365-371) — before updateState() looks for state elements without residuals (abstract.cpp:52-56).
  /// Then the retirement rule for the residual kinds Ceres never removes (inertial, pose prior): everything older than the window and
  /// than every retained landmark goes, so the tables (and variables_) stay bounded while the IMU runs.
  auto updateLandmarks(const Range& range) -> void final {
    std::erase_if(landmarks_, [&](const auto* landmark) { return !landmark->range().intersects(range); });
    const auto retired = [&](const auto* observation) { return !landmarks_.contains(&observation->landmark()); };
    std::erase_if(bearings_, retired);
    std::erase_if(pixels_, retired);
    {  // the same in the library: hs_retire_landmarks takes the rows of the landmarks and their residual rows along and reports where the others moved
      std::vector<std::int32_t> rows, remap(landmark_rows_.size());
      for (std::size_t l = 0; l < landmark_rows_.size(); ++l)
        if (!landmarks_.contains(landmark_rows_[l])) rows.push_back(static_cast<std::int32_t>(l));
      if (!rows.empty()) {
        check(hs_retire_landmarks(handle_, static_cast<int>(rows.size()), rows.data(), remap.data()));
        std::vector<Landmark<Position<Scalar>>*> kept(landmark_rows_.size() - rows.size());
        for (std::size_t l = 0; l < landmark_rows_.size(); ++l) {
          if (remap[l] < 0) {
            landmark_row_.erase(landmark_rows_[l]);
          } else {
            kept[static_cast<std::size_t>(remap[l])] = landmark_rows_[l];
            landmark_row_[landmark_rows_[l]] = remap[l];
          }
        }
        landmark_rows_ = std::move(kept);
      }
    }
    if (!retire_old_observations_) return;
    auto oldest = range.lowerBound();
    for (const auto* landmark : landmarks_) oldest = std::min(oldest, landmark->range().lowerBound());
    const auto expired = [&](const auto* observation) { return observation->measurement().stamp() < oldest; };
    std::erase_if(inertials_, expired);
    check(hs_retire_residuals_before(handle_, HS_INERTIAL, oldest));
    // a pose prior goes once every control point it constrains is constant (stamp <= window lower bound, cc:319-328): until then it may be
    // what anchors the gauge of the free control points
    const auto [left_padding, right_padding] = state().interpolator()->layout().outerPadding();
    (void)left_padding;
    const auto prior_expired = [&, reach = separation_ * (right_padding + 1)](const auto* observation) {
      return expired(observation) && observation->measurement().stamp() + reach <= range.lowerBound();
    };
    std::erase_if(priors_, prior_expired);
    // (stamp < oldest and stamp + reach <= lower bound: one threshold)
    check(hs_retire_residuals_before(handle_, HS_PRIOR, std::min(oldest, std::nextafter(range.lowerBound() - separation_ * (right_padding + 1), std::numeric_limits<Stamp>::max()))));
  }

  /// CHECK(false) upstream (cc:384-386), yet AbstractOptimizer::process(InertialMeasurement) calls it whenever a bias spline is empty
  /// or does not contain the stamp and then DCHECKs that it does (abstract.cpp:278-289): this override is the only place the bias
  /// elements can come from. Both splines get the same knots: created at the first call, extended past the range afterwards.
  auto updateSensor(IMU& imu, const Range& range) -> void final {
    DCHECK(imu_ == &imu);
    extendBias<Traits<IMU>::GyroscopeBias>(imu.gyroscopeBias(), range);
    extendBias<Traits<IMU>::AccelerometerBias>(imu.accelerometerBias(), range);
  }

  /// Uniform knots at multiples of bias_separation_ (shifted by the left padding of the interpolator, like the state's bootstrap at
  /// abstract.cpp:87-93); a new element starts from the value of the last one (the bias random walk's best guess), zero for the first.
  /// The spline is kept valid one state separation beyond the range, so that a sample on the upper boundary of the window — and the
  /// extrapolated control point abstract.cpp:127-137 adds next — stay inside. Same rule as Optimizer::extendBias of the host mirror
  /// (hyperslam_amd/host/optimizer.hpp), which the replay tests run.
  template <typename TElement>
  auto extendBias(AbstractState& bias, const Range& range) -> void {
    auto& elements = bias.elements();
    const auto layout = bias.interpolator()->layout();
    const auto [left_padding, right_padding] = layout.outerPadding();
    if (elements.empty()) {
      const auto first = std::floor(range.lowerBound() / bias_separation_) * bias_separation_ - left_padding * bias_separation_;
      for (auto i = 0; i < layout.outer.size; ++i) {
        auto element = std::make_unique<TElement>();
        element->stamp() = first + i * bias_separation_;
        element->variable().setZero();
        elements.insert(std::move(element));
      }
    }
    // range() of the spline ends at the stamp of its right_padding-th element from the end
    while ((*std::prev(elements.end(), 1 + right_padding))->stamp() <= range.upperBound() + separation_) {
      const auto& last = static_cast<const TElement&>(**elements.rbegin());
      auto element = std::make_unique<TElement>();
      element->stamp() = last.stamp() + bias_separation_;
      element->variable() = last.variable();
      elements.insert(std::move(element));
    }
  }

  hs_problem* handle_{nullptr};
  std::vector<const Camera*> cameras_;
  std::vector<const Sensor*> pose_sensors_;
  std::unordered_map<const Sensor*, std::int32_t> camera_index_, pose_sensor_index_;
  IMU* imu_{nullptr};
  std::vector<VisualBearingObservation*> bearings_;
  std::vector<VisualPixelObservation*> pixels_;
  std::vector<ManifoldObservation<Manifold>*> priors_;
  std::vector<InertialObservation<Manifold>*> inertials_;
  std::unordered_map<const AbstractLandmark*, std::int32_t> landmark_row_;  // rows of the library's landmark table (delta interface)
  std::vector<Landmark<Position<Scalar>>*> landmark_rows_;
  bool rotation_constant_{false}, translation_constant_{false}, gravity_constant_{true}, retire_old_observations_{true};
  Stamp bias_separation_{1.0};
  std::size_t skipped_windows_{0};
};

using HipOptimizer = Optimizer<kOptimizerSuiteHIP>;

/// The branch of Backend::Backend (backend.cpp:37-60) for `suite: hip`: same sequence as the Ceres branch — optimizer, environment,
/// state constancy from the YAML — minus the Ceres manifold objects (the library applies the same retractions).
inline auto make_hip_optimizer(const YAML::Node& node, const std::vector<Sensor*>& sensors) -> std::unique_ptr<AbstractOptimizer> {
  auto optimizer = std::make_unique<HipOptimizer>(node, sensors);
  auto environment = std::make_unique<Environment<SE3<Scalar>>>();
  optimizer->swapEnvironment(environment);
  CHECK(yaml::ReadAs<bool>(node, "time_constant")) << "the HIP backend keeps control-point stamps fixed";
  optimizer->setStateConstancy(yaml::ReadAs<bool>(node, "rotation_constant"), yaml::ReadAs<bool>(node, "translation_constant"));
  return optimizer;
}

}  // namespace hyper
