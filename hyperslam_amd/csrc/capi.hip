// capi.hip — C ABI of libhyperslam_hip.so (include/hyperslam_hip.h): the extern "C" entry points. The one translation unit of the
// library; its host side is split by concern:
//   host_tables.hpp   hs_problem, batched uploads, prepare(): the caller's tables -> the sorted tables in HBM (problem.hpp)
//   host_launch.hpp   the launch sequence of an LM iteration, the RCCL exchange, kernel set-up
//   capi.hip          the ABI functions (setters, hs_solve, hs_linearize, hs_cost_function_evaluate, manifolds, tracks, ...)
//
// Replaces CeresOptimizer::{add(...), updateState, addLandmark, updateLandmarks, optimize}
// (/root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:189-382) behind flat tables. There is no CPU fallback:
// every evaluation entry point runs the gfx950 kernels of kernels.hpp and fails with HS_ERR_DEVICE if no GPU is usable.
#include "host_tables.hpp"
#include "host_launch.hpp"

/// Rows of a residual table that satisfy `drop(i)` leave; the others keep their order.
template <class Drop>
static int drop_rows(int n, Drop&& drop, std::vector<double>* stamp, std::vector<double>* meas, int width, std::vector<int32_t>* a = nullptr, std::vector<int32_t>* b = nullptr) {
  int w = 0;
  for (int i = 0; i < n; ++i) {
    if (drop(i)) continue;
    if (w != i) {
      (*stamp)[w] = (*stamp)[i];
      for (int c = 0; c < width; ++c) (*meas)[size_t(width) * w + c] = (*meas)[size_t(width) * i + c];
      if (a) (*a)[w] = (*a)[i];
      if (b) (*b)[w] = (*b)[i];
    }
    ++w;
  }
  stamp->resize(w), meas->resize(size_t(width) * w);
  if (a) a->resize(w);
  if (b) b->resize(w);
  return n - w;
}

extern "C" {

int hs_version(void) { return 1; }
const char* hs_arch(void) { return "gfx950"; }

int hs_create(int device, void* stream, hs_problem** out) {
  if (!out) return HS_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return HS_ERR_DEVICE;
  hs_problem* p = new hs_problem();
  p->device = device;
  if (const char* e = std::getenv("HS_REFERENCE_LITERAL")) p->inertial_mode = std::atoi(e) ? HS_INERTIAL_AS_REFERENCE : HS_INERTIAL_EXACT;
  if (const char* e = std::getenv("HS_STAGE_TIMING")) p->stage_timing = std::atoi(e) != 0;
  if (const char* e = std::getenv("HS_HOST_TIMING")) p->host_timing = std::atoi(e) != 0;
  if (hipSetDevice(device) != hipSuccess) {
    delete p;
    return HS_ERR_DEVICE;
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) p->n_cu = cus;
  }
  if (stream) {
    p->stream = static_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) {
      delete p;
      return HS_ERR_DEVICE;
    }
    p->own_stream = true;
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&p->h_state), sizeof(DevState), hipHostMallocDefault) != hipSuccess) {
    delete p;
    return HS_ERR_DEVICE;
  }
  warm_kernels(device);
  if (set_func_attributes(p) != HS_OK || ensure_side_stream(p) != HS_OK) {
    const std::string e = p->err;
    std::fprintf(stderr, "hyperslam_hip: %s\n", e.c_str());
    delete p;
    return HS_ERR_DEVICE;
  }
  *out = p;
  return HS_OK;
}

int hs_destroy(hs_problem* p) {
  if (!p) return HS_OK;
  if (p->host_timing && p->host_calls) {
    std::fprintf(stderr, "hs host timing over %d hs_solve calls [ms per call]: structure + tables %.4f, uploads %.4f, launches %.4f, wait %.4f\n", p->host_calls,
                 p->host_ms[0] / p->host_calls, p->host_ms[1] / p->host_calls, p->host_ms[2] / p->host_calls, p->host_ms[3] / p->host_calls);
    const size_t n = p->host_log.size() / 4, h = n / 2;  // second half of the calls: buffers have reached their size, nothing is allocated
    double m[4] = {0, 0, 0, 0};
    for (size_t i = h; i < n; ++i)
      for (int c = 0; c < 4; ++c) m[c] += p->host_log[4 * i + c] / double(n - h);
    std::fprintf(stderr, "hs host timing, second half of the calls: structure + tables %.4f, uploads %.4f, launches %.4f, wait %.4f\n", m[0], m[1], m[2], m[3]);
    if (const char* e = std::getenv("HS_HOST_TIMING"); e && std::atoi(e) >= 2)  // HS_HOST_TIMING=2: every call (where a one-off cost sits)
      for (size_t i = 0; i < n; ++i)
        std::fprintf(stderr, "hs host timing call %zu: structure + tables %.4f, uploads %.4f, launches %.4f, wait %.4f\n", i, p->host_log[4 * i],
                     p->host_log[4 * i + 1], p->host_log[4 * i + 2], p->host_log[4 * i + 3]);
  }
  if (p->scratch) hs_destroy(p->scratch), p->scratch = nullptr;
  (void)hipSetDevice(p->device);
  (void)hipStreamSynchronize(p->stream);
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_join) (void)hipEventDestroy(p->ev_join);
  if (p->ev_irec) (void)hipEventDestroy(p->ev_irec);
  if (p->side) (void)hipStreamDestroy(p->side);
  if (p->h_state) (void)hipHostFree(p->h_state);
  if (p->h_result) (void)hipHostFree(p->h_result);
  if (p->rccl_comm && rccl_api()) (void)rccl_api()->CommDestroy(static_cast<ncclComm_t>(p->rccl_comm));
  if (p->own_stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return HS_OK;
}

const char* hs_last_error(const hs_problem* p) { return p ? p->err.c_str() : "null handle"; }

/// How far a control-point stamp may be from t0 + j dt and still be that knot: 1e-6 of the spacing, or eight ulps of the largest stamp
/// of the table where that is more (epoch-scale stamps). A hole or a non-uniform knot is off by a whole spacing.
static double knot_tolerance(double t0, double dt, int n_cp) {
  return std::max(1e-6 * dt, 8.0 * 2.220446049250313e-16 * std::max(std::fabs(t0), std::fabs(t0 + n_cp * dt)));
}

int hs_set_spline(hs_problem* p, int order, double t0, double dt, int n_cp, const double* cp, const uint8_t* cp_constant, int rc, int tc) {
  if (!p) return HS_ERR_INVALID;
  if (order < 2 || order > hsd::kMaxOrder) HS_FAIL(HS_ERR_INVALID, "spline order out of range");
  if (n_cp < order || !(dt > 0) || !cp) HS_FAIL(HS_ERR_INVALID, "need n_cp >= order, dt > 0 and a control-point table");
  // The basis is uniform: control point j is taken to sit at t0 + j dt, whatever its row says. A table with a hole (upstream prunes state
  // elements one by one, ceres/optimizer.cpp:330-341) or non-uniform knots would silently re-index every later control point: refused.
  // The tolerance has to cover how stamps are made, not only what a knot is: accumulated (t += dt, up to n_cp additions) or converted from
  // integer nanoseconds, at epoch-scale magnitudes (t0 ~ 1.7e9 s: one ulp is 2.4e-7 s) — a few ulps of the largest stamp — and it stays
  // six orders of magnitude below a knot spacing wherever the stamps resolve one.
  const double knot_tol = knot_tolerance(t0, dt, n_cp);
  for (int j = 0; j < n_cp; ++j)
    if (!(std::fabs(cp[8 * j + 7] - (t0 + j * dt)) <= knot_tol))
      HS_FAIL(HS_ERR_KNOTS, "control-point stamps are not t0 + j dt (row " + std::to_string(j) + "): the spline basis is uniform, a table with a hole or non-uniform knots is refused");
  // Same knots and flags as the resident table: only the values (and the constancy mask) changed — the sorted tables, every index and size stand.
  const bool same_structure = p->k == order && p->t0 == t0 && p->dt == dt && p->n_cp == n_cp && p->rot_const == (rc != 0) && p->trans_const == (tc != 0) &&
                              p->cp.size() == size_t(8) * n_cp;
  bool same_values = same_structure && std::memcmp(p->cp.data(), cp, sizeof(double) * 8 * n_cp) == 0;
  for (int j = 0; same_values && j < n_cp; ++j) same_values = p->cp_const[j] == (cp_constant ? cp_constant[j] : 0);
  if (same_values && !p->device_ahead) return HS_OK;  // (nothing to send: the device holds exactly this table)
  p->k = order, p->t0 = t0, p->dt = dt, p->n_cp = n_cp;
  p->cp.assign(cp, cp + size_t(8) * n_cp);
  p->cp_const.assign(n_cp, 0);
  if (cp_constant) p->cp_const.assign(cp_constant, cp_constant + n_cp);
  p->rot_const = rc != 0, p->trans_const = tc != 0;
  p->touch(same_structure ? unsigned(hs_problem::vCp) : unsigned(hs_problem::kAll));
  return HS_OK;
}

int hs_set_cameras(hs_problem* p, int n, const double* T_bs, const double* intr, const double* dist) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || n > 0xffff || (n && (!T_bs || !intr || !dist))) HS_FAIL(HS_ERR_INVALID, "bad camera table");
  const bool same_count = p->n_cam == n && p->cam.size() == size_t(kCamStride) * n;
  const std::vector<double> before = p->cam;
  p->n_cam = n;
  p->cam.assign(size_t(kCamStride) * n, 0.0);
  for (int i = 0; i < n; ++i) {
    double* c = &p->cam[size_t(kCamStride) * i];
    std::memcpy(c, T_bs + 7 * i, 56), std::memcpy(c + 7, intr + 4 * i, 32), std::memcpy(c + 11, dist + 4 * i, 32);
  }
  if (same_count && before == p->cam) return HS_OK;
  p->touch(same_count ? unsigned(hs_problem::vCam) : unsigned(hs_problem::vCam | hs_problem::kVis));  // (the visual section checks the camera indices)
  return HS_OK;
}

int hs_set_sensors(hs_problem* p, int n, const double* T_bs) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && !T_bs)) HS_FAIL(HS_ERR_INVALID, "bad sensor table");
  const bool same_count = p->n_sensor == n && p->sensor.size() == size_t(8) * n;
  const std::vector<double> before = p->sensor;
  p->n_sensor = n;
  p->sensor.assign(size_t(8) * n, 0.0);
  for (int i = 0; i < n; ++i) std::memcpy(&p->sensor[size_t(8) * i], T_bs + 7 * i, 56);
  if (same_count && before == p->sensor) return HS_OK;
  p->touch(same_count ? unsigned(hs_problem::vSensor) : unsigned(hs_problem::vSensor | hs_problem::kPri));  // (the prior section checks the sensor indices)
  return HS_OK;
}

int hs_set_landmarks(hs_problem* p, int n, const double* xyz, const uint8_t* constant) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && !xyz)) HS_FAIL(HS_ERR_INVALID, "bad landmark table");
  bool same_structure = p->n_lm == n && p->lm.size() == size_t(3) * n && p->lm_const.size() == size_t(n);
  for (int i = 0; same_structure && i < n; ++i) same_structure = p->lm_const[i] == (constant ? constant[i] : 0);
  p->n_lm = n;
  p->lm.assign(xyz, xyz + size_t(3) * n);
  p->lm_const.assign(n, 0);
  if (constant) p->lm_const.assign(constant, constant + n);
  p->touch(same_structure ? unsigned(hs_problem::vLm) : unsigned(hs_problem::vLm | hs_problem::kVis));
  return HS_OK;
}

int hs_set_imu(hs_problem* p, const double* T_bs, const double* i_g, const double* i_a, const double* S_g, const double* X_a, int bias_order,
               double bias_t0, double bias_dt, int n_bias, const double* bias_g, const double* bias_a, int bias_constant) {
  if (!p) return HS_ERR_INVALID;
  if (!T_bs || !i_g || !i_a || !S_g || !X_a || !bias_g || !bias_a) HS_FAIL(HS_ERR_INVALID, "null IMU table");
  if (bias_order < 2 || bias_order > hsd::kMaxOrder || n_bias < bias_order || !(bias_dt > 0)) HS_FAIL(HS_ERR_INVALID, "bad bias spline");
  const bool same_structure = p->has_imu && p->kb == bias_order && p->bias_t0 == bias_t0 && p->bias_dt == bias_dt && p->n_bias == n_bias &&
                              p->bias_const == (bias_constant != 0);
  p->has_imu = true;
  std::memcpy(p->imu_T_bs, T_bs, 56), std::memcpy(p->imu_i_g, i_g, 48), std::memcpy(p->imu_i_a, i_a, 48);
  std::memcpy(p->imu_S_g, S_g, 72), std::memcpy(p->imu_X_a, X_a, 72);
  p->kb = bias_order, p->bias_t0 = bias_t0, p->bias_dt = bias_dt, p->n_bias = n_bias;
  p->bias_g.assign(bias_g, bias_g + size_t(4) * n_bias), p->bias_a.assign(bias_a, bias_a + size_t(4) * n_bias);
  p->bias_const = bias_constant != 0;
  // (new bias knots: the inertial records index them, the border of the reduced system changes size)
  p->touch(same_structure ? unsigned(hs_problem::vImu | hs_problem::vBias) : unsigned(hs_problem::vImu | hs_problem::vBias | hs_problem::kIne | hs_problem::kTail));
  return HS_OK;
}

int hs_set_inertial_jacobian(hs_problem* p, int mode) {
  if (!p) return HS_ERR_INVALID;
  if (mode != HS_INERTIAL_AS_REFERENCE && mode != HS_INERTIAL_EXACT) HS_FAIL(HS_ERR_INVALID, "unknown inertial Jacobian mode");
  if (mode != p->inertial_mode) p->inertial_mode = mode, p->touch(hs_problem::kTail);
  return HS_OK;
}

int hs_set_gravity(hs_problem* p, const double* g, int constant) {
  if (!p || !g) return HS_ERR_INVALID;
  const bool same_structure = p->gravity_const == (constant != 0);
  std::memcpy(p->gravity, g, 24);
  p->gravity_const = constant != 0;
  p->touch(same_structure ? unsigned(hs_problem::vGravity) : unsigned(hs_problem::vGravity | hs_problem::kTail));
  return HS_OK;
}

int hs_set_pixel_residuals(hs_problem* p, int n, const double* st, const double* px, const int32_t* lm, const int32_t* cam) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !px || !lm || !cam))) HS_FAIL(HS_ERR_INVALID, "bad pixel residual table");
  p->px_stamp.assign(st, st + n), p->px_meas.assign(px, px + size_t(2) * n), p->px_lm.assign(lm, lm + n), p->px_cam.assign(cam, cam + n);
  p->touch(hs_problem::kVis | hs_problem::kTail);
  return HS_OK;
}
int hs_set_bearing_residuals(hs_problem* p, int n, const double* st, const double* b, const int32_t* lm, const int32_t* cam) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !b || !lm || !cam))) HS_FAIL(HS_ERR_INVALID, "bad bearing residual table");
  p->br_stamp.assign(st, st + n), p->br_meas.assign(b, b + size_t(3) * n), p->br_lm.assign(lm, lm + n), p->br_cam.assign(cam, cam + n);
  p->touch(hs_problem::kVis | hs_problem::kTail);
  return HS_OK;
}
int hs_set_prior_residuals(hs_problem* p, int n, const double* st, const double* poses, const int32_t* sensor) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !poses || !sensor))) HS_FAIL(HS_ERR_INVALID, "bad prior residual table");
  p->pr_stamp.assign(st, st + n), p->pr_meas.assign(poses, poses + size_t(7) * n), p->pr_sensor.assign(sensor, sensor + n);
  p->touch(hs_problem::kPri | hs_problem::kTail);
  return HS_OK;
}
int hs_set_inertial_residuals(hs_problem* p, int n, const double* st, const double* m) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !m))) HS_FAIL(HS_ERR_INVALID, "bad inertial residual table");
  p->in_stamp.assign(st, st + n), p->in_meas.assign(m, m + size_t(6) * n);
  p->touch(hs_problem::kIne | hs_problem::kTail);
  return HS_OK;
}

// ---- delta interface ------------------------------------------------------------------------------------------------------------------
// The reference keeps ceres::Problem incrementally: AddResidualBlock when an observation arrives (optimizer.cpp:189-274 <- abstract.cpp:246-259),
// AddParameterBlock / RemoveParameterBlock when a landmark appears / leaves the window (optimizer.cpp:347-382), and optimize() only solves.
// Same here: rows are appended to / retired from the resident tables when the caller learns of them, hs_stage() sorts and uploads between
// solves, and an hs_solve() that finds nothing changed sends nothing (the control points, if they were re-sent with new values, only).

/// The host copies of the variables follow the device before a delta call edits the tables next to them (the solve moved the point on the
/// device; the next structural upload re-sends landmarks in a new order).
static int pull_state(hs_problem* p) {
  if (!p->device_ahead || p->dirty) return HS_OK;
  std::vector<double> a(p->cp.size()), b(p->lm.size()), c(p->bias_g.size()), d(p->bias_a.size());
  double g[3];
  int rc = HS_OK;
  if (!a.empty()) rc = hs_get_control_points(p, a.data());
  if (!rc && p->n_lm) rc = hs_get_landmarks(p, b.data());
  if (!rc && p->has_imu) rc = hs_get_bias(p, c.data(), d.data());
  if (!rc && p->has_imu) rc = hs_get_gravity(p, g);
  if (!rc) p->device_ahead = false;
  return rc;
}

int hs_append_landmarks(hs_problem* p, int n, const double* xyz, const uint8_t* constant, int32_t* first_index) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && !xyz)) HS_FAIL(HS_ERR_INVALID, "bad landmark rows");
  const int rc = pull_state(p);
  if (rc) return rc;
  if (first_index) *first_index = p->n_lm;
  if (n == 0) return HS_OK;
  p->lm.insert(p->lm.end(), xyz, xyz + size_t(3) * n);
  for (int i = 0; i < n; ++i) p->lm_const.push_back(constant ? constant[i] : 0);
  p->n_lm += n;
  p->touch(hs_problem::kVis | hs_problem::kTail | hs_problem::vLm);
  return HS_OK;
}

int hs_append_pixel_residuals(hs_problem* p, int n, const double* st, const double* px, const int32_t* lm, const int32_t* cam) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !px || !lm || !cam))) HS_FAIL(HS_ERR_INVALID, "bad pixel residual rows");
  if (n == 0) return HS_OK;
  const int rc = pull_state(p);
  if (rc) return rc;
  p->px_stamp.insert(p->px_stamp.end(), st, st + n), p->px_meas.insert(p->px_meas.end(), px, px + size_t(2) * n);
  p->px_lm.insert(p->px_lm.end(), lm, lm + n), p->px_cam.insert(p->px_cam.end(), cam, cam + n);
  p->touch(hs_problem::kVis | hs_problem::kTail);
  return HS_OK;
}
int hs_append_bearing_residuals(hs_problem* p, int n, const double* st, const double* b, const int32_t* lm, const int32_t* cam) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !b || !lm || !cam))) HS_FAIL(HS_ERR_INVALID, "bad bearing residual rows");
  if (n == 0) return HS_OK;
  const int rc = pull_state(p);
  if (rc) return rc;
  p->br_stamp.insert(p->br_stamp.end(), st, st + n), p->br_meas.insert(p->br_meas.end(), b, b + size_t(3) * n);
  p->br_lm.insert(p->br_lm.end(), lm, lm + n), p->br_cam.insert(p->br_cam.end(), cam, cam + n);
  p->touch(hs_problem::kVis | hs_problem::kTail);
  return HS_OK;
}
int hs_append_prior_residuals(hs_problem* p, int n, const double* st, const double* poses, const int32_t* sensor) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !poses || !sensor))) HS_FAIL(HS_ERR_INVALID, "bad prior residual rows");
  if (n == 0) return HS_OK;
  const int rc = pull_state(p);
  if (rc) return rc;
  p->pr_stamp.insert(p->pr_stamp.end(), st, st + n), p->pr_meas.insert(p->pr_meas.end(), poses, poses + size_t(7) * n);
  p->pr_sensor.insert(p->pr_sensor.end(), sensor, sensor + n);
  p->touch(hs_problem::kPri | hs_problem::kTail);
  return HS_OK;
}
int hs_append_inertial_residuals(hs_problem* p, int n, const double* st, const double* m) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !m))) HS_FAIL(HS_ERR_INVALID, "bad inertial residual rows");
  if (n == 0) return HS_OK;
  const int rc = pull_state(p);
  if (rc) return rc;
  p->in_stamp.insert(p->in_stamp.end(), st, st + n), p->in_meas.insert(p->in_meas.end(), m, m + size_t(6) * n);
  p->touch(hs_problem::kIne | hs_problem::kTail);
  return HS_OK;
}

int hs_retire_landmarks(hs_problem* p, int n, const int32_t* ids, int32_t* remap) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && !ids)) HS_FAIL(HS_ERR_INVALID, "bad landmark list");
  const int n_old = p->n_lm;
  std::vector<int32_t> local;
  if (!remap) local.resize(n_old), remap = local.data();
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= n_old) HS_FAIL(HS_ERR_INVALID, "hs_retire_landmarks: landmark outside the landmark table");
  if (n == 0) {
    for (int t = 0; t < n_old; ++t) remap[t] = t;
    return HS_OK;
  }
  const int rc = pull_state(p);
  if (rc) return rc;
  for (int t = 0; t < n_old; ++t) remap[t] = 0;
  for (int i = 0; i < n; ++i) remap[ids[i]] = -1;
  int w = 0;
  for (int t = 0; t < n_old; ++t) {
    if (remap[t] < 0) continue;
    for (int c = 0; c < 3; ++c) p->lm[size_t(3) * w + c] = p->lm[size_t(3) * t + c];
    p->lm_const[w] = p->lm_const[t];
    remap[t] = w++;
  }
  p->n_lm = w, p->lm.resize(size_t(3) * w), p->lm_const.resize(w);
  // RemoveParameterBlock takes the residual blocks of the landmark along (optimizer.cpp:365-371, enable_fast_removal)
  drop_rows(int(p->px_stamp.size()), [&](int i) { return remap[p->px_lm[i]] < 0; }, &p->px_stamp, &p->px_meas, 2, &p->px_lm, &p->px_cam);
  drop_rows(int(p->br_stamp.size()), [&](int i) { return remap[p->br_lm[i]] < 0; }, &p->br_stamp, &p->br_meas, 3, &p->br_lm, &p->br_cam);
  for (int32_t& l : p->px_lm) l = remap[l];
  for (int32_t& l : p->br_lm) l = remap[l];
  p->touch(hs_problem::kVis | hs_problem::kTail | hs_problem::vLm);
  return HS_OK;
}

int hs_retire_residuals_before(hs_problem* p, int type, double stamp) {
  if (!p) return HS_ERR_INVALID;
  if (type < HS_PIXEL || type > HS_INERTIAL) HS_FAIL(HS_ERR_INVALID, "unknown factor type");
  const int rc = pull_state(p);  // (the variables stay where the last solve left them)
  if (rc) return rc;
  int dropped = 0;
  switch (type) {
    case HS_PIXEL: dropped = drop_rows(int(p->px_stamp.size()), [&](int i) { return p->px_stamp[i] < stamp; }, &p->px_stamp, &p->px_meas, 2, &p->px_lm, &p->px_cam); break;
    case HS_BEARING: dropped = drop_rows(int(p->br_stamp.size()), [&](int i) { return p->br_stamp[i] < stamp; }, &p->br_stamp, &p->br_meas, 3, &p->br_lm, &p->br_cam); break;
    case HS_PRIOR: dropped = drop_rows(int(p->pr_stamp.size()), [&](int i) { return p->pr_stamp[i] < stamp; }, &p->pr_stamp, &p->pr_meas, 7, &p->pr_sensor); break;
    default: dropped = drop_rows(int(p->in_stamp.size()), [&](int i) { return p->in_stamp[i] < stamp; }, &p->in_stamp, &p->in_meas, 6); break;
  }
  if (dropped) {
    p->touch((type <= HS_BEARING ? hs_problem::kVis : type == HS_PRIOR ? hs_problem::kPri : hs_problem::kIne) | hs_problem::kTail);
  }
  return HS_OK;
}

int hs_stage(hs_problem* p) {
  if (!p) return HS_ERR_INVALID;
  return prepare(p);  // sorts what changed and enqueues its upload on the handle's stream; nothing is awaited
}

int hs_num_residuals(hs_problem* p, int type) {
  if (!p) return -1;
  switch (type) {
    case HS_PIXEL: return int(p->px_stamp.size());
    case HS_BEARING: return int(p->br_stamp.size());
    case HS_PRIOR: return int(p->pr_stamp.size());
    case HS_INERTIAL: return int(p->in_stamp.size());
  }
  return -1;
}
int hs_dim_pose(hs_problem* p) { return p ? 6 * p->n_cp + (p->has_imu ? 6 * p->n_bias + 2 : 0) : -1; }

int hs_residual_layout(hs_problem* p, int type, int idx, int32_t* num_blocks, int32_t* indices, int32_t* sizes, int32_t* offsets, int32_t* block_ids,
                       int32_t* num_parameters, int32_t* num_residuals) {
  if (!p) return HS_ERR_INVALID;
  if (type < 0 || type > 3 || idx < 0 || idx >= hs_num_residuals(p, type)) HS_FAIL(HS_ERR_INVALID, "residual index out of range");
  if (p->n_cp == 0) HS_FAIL(HS_ERR_STATE, "hs_set_spline has not been called");
  const BlockLayout L = make_block_layout(type, p->k, p->kb);
  *num_blocks = L.num_blocks, *num_parameters = L.num_parameters, *num_residuals = L.num_residuals;
  for (int i = 0; i < 4; ++i) indices[i] = L.indices[i];
  for (int i = 0; i < L.num_blocks; ++i) sizes[i] = L.sizes[i], offsets[i] = L.offsets[i];
  double st = 0;
  switch (type) {
    case HS_PIXEL: st = p->px_stamp[idx]; break;
    case HS_BEARING: st = p->br_stamp[idx]; break;
    case HS_PRIOR: st = p->pr_stamp[idx]; break;
    case HS_INERTIAL: st = p->in_stamp[idx]; break;
  }
  const int first = h_segment_first(st, p->t0, p->dt, p->k);
  int b = 0;
  for (int j = 0; j < p->k; ++j) block_ids[b++] = first + j;
  if (type == HS_PIXEL || type == HS_BEARING) {
    const int cam = type == HS_PIXEL ? p->px_cam[idx] : p->br_cam[idx];
    block_ids[b++] = cam, block_ids[b++] = cam, block_ids[b++] = cam;
    block_ids[b++] = type == HS_PIXEL ? p->px_lm[idx] : p->br_lm[idx];
  } else if (type == HS_PRIOR) {
    block_ids[b++] = p->pr_sensor[idx];
  } else {
    for (int j = 0; j < 5; ++j) block_ids[b++] = 0;
    const int fb = h_segment_first(st, p->bias_t0, p->bias_dt, p->kb);
    for (int j = 0; j < p->kb; ++j) block_ids[b++] = fb + j;
    for (int j = 0; j < p->kb; ++j) block_ids[b++] = fb + j;
    block_ids[b++] = 0;
  }
  return HS_OK;
}

/// Optional sensor-block outputs of hs_linearize (kernels_sensor.hpp): a pass of its own, the solver's kernels do not carry these columns.
static int linearize_sensor_blocks(hs_problem* p, int type, int robustify, const hs_linearization* out) {
  if (!out->J_extrinsics && !out->J_intrinsics && !out->J_distortion && !out->J_gyro_intrinsics && !out->J_acc_intrinsics && !out->J_gyro_sensitivity &&
      !out->J_acc_offsets)
    return HS_OK;
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  const int k = p->k;
  const bool visual = type == HS_PIXEL || type == HS_BEARING;
  const int n = visual ? T.n_vis : (type == HS_PRIOR ? T.n_pri : T.n_ine);
  const int REC = visual ? kSensorRecVisual : (type == HS_PRIOR ? kSensorRecPrior : kSensorRecInertial);
  if (n == 0) return HS_OK;
  HIP_TRY(p->d_dbg.reserve(size_t(n) * REC));
  const int nb = (n + kBlock - 1) / kBlock;
  if (visual) {
    HS_ORDER_SWITCH(k, k_sensor_visual<K><<<nb, kBlock, 0, s>>>(T, p->d_dbg.p, p->d_v_dbgpos.p, robustify));
  } else if (type == HS_PRIOR) {
    HS_ORDER_SWITCH(k, k_sensor_prior<K><<<nb, kBlock, 0, s>>>(T, p->d_dbg.p));
  } else {
    HS_ORDER_SWITCH(k, k_sensor_inertial<K, 4><<<nb, kBlock, 0, s>>>(T, p->d_dbg.p, robustify));
  }
  HIP_TRY(hipGetLastError());
  std::vector<double> rec(size_t(n) * REC);
  HIP_TRY(hipMemcpyAsync(rec.data(), p->d_dbg.p, rec.size() * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (visual) {
    const int n_px = int(p->px_stamp.size()), n_br = int(p->br_stamp.size());
    const int base = type == HS_PIXEL ? 0 : n_px, cnt = type == HS_PIXEL ? n_px : n_br, nres = type == HS_PIXEL ? 2 : 1;
    for (int i = 0; i < cnt; ++i) {
      const double* r = &rec[size_t(base + i) * REC];
      if (out->J_extrinsics) std::memcpy(out->J_extrinsics + size_t(i) * nres * 6, r, sizeof(double) * nres * 6);
      if (type == HS_PIXEL) {
        if (out->J_intrinsics) std::memcpy(out->J_intrinsics + size_t(i) * 8, r + 12, 64);
        if (out->J_distortion) std::memcpy(out->J_distortion + size_t(i) * 8, r + 20, 64);
      }
    }
  } else if (type == HS_PRIOR) {
    for (int d = 0; d < n; ++d)
      if (out->J_extrinsics) std::memcpy(out->J_extrinsics + size_t(p->pr_order[d]) * 36, &rec[size_t(d) * REC], 36 * 8);
  } else {
    for (int d = 0; d < n; ++d) {
      const size_t i = size_t(p->in_order[d]);
      const double* r = &rec[size_t(d) * REC];
      if (out->J_extrinsics) std::memcpy(out->J_extrinsics + i * 36, r, 36 * 8);
      if (out->J_gyro_intrinsics) std::memcpy(out->J_gyro_intrinsics + i * 36, r + 36, 36 * 8);
      if (out->J_acc_intrinsics) std::memcpy(out->J_acc_intrinsics + i * 36, r + 72, 36 * 8);
      if (out->J_gyro_sensitivity) std::memcpy(out->J_gyro_sensitivity + i * 54, r + 108, 54 * 8);
      if (out->J_acc_offsets) std::memcpy(out->J_acc_offsets + i * 54, r + 162, 54 * 8);
    }
  }
  return HS_OK;
}

static int linearize_impl(hs_problem* p, int type, int robustify, const hs_linearization* out);

/// HS_GUARD=1 (host_tables.hpp, GuardRegistry): the patterns behind every device table of the process are checked before hs_solve, hs_cost,
/// hs_reduced_system and hs_linearize return.
static int guard_check(hs_problem* p) {
  if (!GuardRegistry::on()) return HS_OK;
  size_t at = 0;
  const size_t bytes = GuardRegistry::get().check(&at);
  if (bytes) HS_FAIL(HS_ERR_DEVICE, "HS_GUARD: a device table of " + std::to_string(bytes) + " bytes was written past its end (first overwritten guard byte at +" + std::to_string(at) + ")");
  return HS_OK;
}

/// hs_linearize with CostConfiguration::weights (exteroceptive.cpp:129-147): output = W * distance, J_w = W * J_m * J_e, then Ceres' loss
/// corrector on the weighted residual. The device produces the unweighted, uncorrected rows (linearize_impl, robustify = 0); W (n_res x
/// n_res) and the corrector are applied per residual block while the rows are handed out.
int hs_linearize(hs_problem* p, int type, int robustify, const hs_linearization* out) {
  if (!p || !out) return HS_ERR_INVALID;
  if (type < HS_PIXEL || type > HS_INERTIAL) HS_FAIL(HS_ERR_INVALID, "unknown factor type");
  const std::vector<double>& W = p->weights[type];
  if (W.empty()) {
    const int rc0 = linearize_impl(p, type, robustify, out);
    return rc0 ? rc0 : guard_check(p);
  }
  const int n = hs_num_residuals(p, type), nr = type == HS_PIXEL ? 2 : (type == HS_BEARING ? 1 : 6);
  hs_linearization o = *out;
  std::vector<double> r_tmp, c_tmp;
  if (!o.r) r_tmp.resize(size_t(n) * nr), o.r = r_tmp.data();
  if (!o.cost) c_tmp.resize(n), o.cost = c_tmp.data();
  int rc = linearize_impl(p, type, 0, &o);
  if (!rc) rc = guard_check(p);
  if (rc) return rc;
  struct Block {
    double* J;
    int cols;
  };
  const Block blocks[] = {{o.J_state, 6 * p->k},       {o.J_landmark, 3},        {o.J_bias_g, 3 * p->kb},      {o.J_bias_a, 3 * p->kb},
                          {o.J_gravity, 2},            {o.J_extrinsics, 6},      {o.J_intrinsics, 4},          {o.J_distortion, 4},
                          {o.J_gyro_intrinsics, 6},    {o.J_acc_intrinsics, 6},  {o.J_gyro_sensitivity, 9},    {o.J_acc_offsets, 9}};
  const bool visual = type == HS_PIXEL || type == HS_BEARING;
  std::vector<double> tmp;
  for (int i = 0; i < n; ++i) {
    double rw[6], s2 = 0.0;
    for (int a = 0; a < nr; ++a) {
      rw[a] = 0.0;
      for (int c = 0; c < nr; ++c) rw[a] += W[a * nr + c] * o.r[size_t(i) * nr + c];
      s2 += rw[a] * rw[a];
    }
    double rho = s2, sr = 1.0;  // loss of the factor type on the weighted residual (optimizer.cpp:204,226,250,267)
    if (visual) {
      const double a = type == HS_PIXEL ? kHuberPixel : kHuberBearing;
      if (s2 > a * a) rho = 2.0 * a * std::sqrt(s2) - a * a, sr = std::sqrt(a / std::sqrt(s2));
    } else if (type == HS_INERTIAL) {
      rho = kScaleInertial * s2, sr = std::sqrt(kScaleInertial);
    }
    if (!robustify) sr = 1.0;
    o.cost[i] = 0.5 * rho;
    for (int a = 0; a < nr; ++a) o.r[size_t(i) * nr + a] = sr * rw[a];
    for (const Block& b : blocks) {
      if (!b.J) continue;
      if ((b.J == o.J_landmark || b.J == o.J_intrinsics || b.J == o.J_distortion) && !visual) continue;
      if ((b.J == o.J_intrinsics || b.J == o.J_distortion) && type != HS_PIXEL) continue;
      if ((b.J == o.J_bias_g || b.J == o.J_bias_a || b.J == o.J_gravity || b.J == o.J_gyro_intrinsics || b.J == o.J_acc_intrinsics ||
           b.J == o.J_gyro_sensitivity || b.J == o.J_acc_offsets) && type != HS_INERTIAL)
        continue;
      double* J = b.J + size_t(i) * nr * b.cols;
      tmp.assign(J, J + size_t(nr) * b.cols);
      for (int a = 0; a < nr; ++a)
        for (int c = 0; c < b.cols; ++c) {
          double v = 0.0;
          for (int m = 0; m < nr; ++m) v += W[a * nr + m] * tmp[size_t(m) * b.cols + c];
          J[size_t(a) * b.cols + c] = sr * v;
        }
    }
  }
  return HS_OK;
}

static int linearize_impl(hs_problem* p, int type, int robustify, const hs_linearization* out) {
  if (!p || !out) return HS_ERR_INVALID;
  int rc = prepare(p);
  if (rc) return rc;
  rc = reset_state(p, 0, 1e4);
  if (rc) return rc;
  if (type < HS_PIXEL || type > HS_INERTIAL) HS_FAIL(HS_ERR_INVALID, "unknown factor type");
  rc = linearize_sensor_blocks(p, type, robustify, out);
  if (rc) return rc;
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  const int k = p->k;
  if (type == HS_PIXEL || type == HS_BEARING) {
    const int n_px = int(p->px_stamp.size()), n_br = int(p->br_stamp.size());
    const int n = T.n_vis, REC = 8 + 12 * k;
    if (n == 0) return HS_OK;
    HIP_TRY(p->d_dbg.reserve(size_t(n) * REC));
    HIP_TRY(p->d_dbg_cost.reserve(n));
    HS_ORDER_SWITCH(k, k_linearize_visual<K><<<(n + lin_block<K>() - 1) / lin_block<K>(), lin_block<K>(), lin_lds_bytes<K>(p), s>>>(T, p->d_dbg.p, p->d_v_dbgpos.p, robustify, nullptr, p->d_dbg_cost.p));
    HIP_TRY(hipGetLastError());
    std::vector<double> rec(size_t(n) * REC), cost(n);
    HIP_TRY(hipMemcpyAsync(rec.data(), p->d_dbg.p, rec.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(cost.data(), p->d_dbg_cost.p, cost.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int base = type == HS_PIXEL ? 0 : n_px, cnt = type == HS_PIXEL ? n_px : n_br, nres = type == HS_PIXEL ? 2 : 1;
    for (int i = 0; i < cnt; ++i) {
      const double* r = &rec[size_t(base + i) * REC];
      for (int rr = 0; rr < nres; ++rr) {
        if (out->r) out->r[size_t(i) * nres + rr] = r[rr];
        if (out->J_landmark)
          for (int c = 0; c < 3; ++c) out->J_landmark[(size_t(i) * nres + rr) * 3 + c] = r[2 + 3 * rr + c];
        if (out->J_state)
          for (int c = 0; c < 6 * k; ++c) out->J_state[(size_t(i) * nres + rr) * 6 * k + c] = r[8 + rr * 6 * k + c];
      }
      if (out->cost) out->cost[i] = cost[base + i];
      if (out->first_cp) out->first_cp[i] = h_segment_first(type == HS_PIXEL ? p->px_stamp[i] : p->br_stamp[i], p->t0, p->dt, k);
    }
    return HS_OK;
  }
  if (type == HS_PRIOR) {
    const int n = T.n_pri, REC = 6 + 36 * k;
    if (n == 0) return HS_OK;
    HIP_TRY(p->d_dbg.reserve(size_t(n) * REC));
    HIP_TRY(p->d_dbg_cost.reserve(n));
    HS_ORDER_SWITCH(k, k_linearize_prior<K><<<p->nb_pri, kBlock, cp_lds_bytes(p), s>>>(T, p->d_dbg.p, nullptr, p->d_dbg_cost.p));
    HIP_TRY(hipGetLastError());
    std::vector<double> rec(size_t(n) * REC), cost(n);
    HIP_TRY(hipMemcpyAsync(rec.data(), p->d_dbg.p, rec.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(cost.data(), p->d_dbg_cost.p, cost.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int d = 0; d < n; ++d) {
      const int i = p->pr_order[d];
      const double* r = &rec[size_t(d) * REC];
      if (out->r) std::memcpy(out->r + size_t(i) * 6, r, 48);
      if (out->J_state) std::memcpy(out->J_state + size_t(i) * 36 * k, r + 6, sizeof(double) * 36 * k);
      if (out->cost) out->cost[i] = cost[d];
      if (out->first_cp) out->first_cp[i] = p->pr_first[d];
    }
    return HS_OK;
  }
  {
    const int n = T.n_ine, kb = p->kb, REC = 18 + 36 * k + 2 * kb;
    if (n == 0) return HS_OK;
    HIP_TRY(p->d_dbg.reserve(size_t(n) * REC));
    HIP_TRY(p->d_dbg_cost.reserve(n));
    HS_ORDER_SWITCH(k, k_linearize_inertial<K, 4><<<p->nb_ine, kInertialBlock * K, cp_lds_bytes(p), s>>>(T, p->d_dbg.p, robustify, nullptr, p->d_dbg_cost.p));
    HIP_TRY(hipGetLastError());
    std::vector<double> rec(size_t(n) * REC), cost(n);
    HIP_TRY(hipMemcpyAsync(rec.data(), p->d_dbg.p, rec.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(cost.data(), p->d_dbg_cost.p, cost.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int d = 0; d < n; ++d) {
      const int i = p->in_order[d];
      const double* r = &rec[size_t(d) * REC];
      if (out->r) std::memcpy(out->r + size_t(i) * 6, r, 48);
      if (out->J_state) std::memcpy(out->J_state + size_t(i) * 36 * k, r + 6, sizeof(double) * 36 * k);
      const double* wg = r + 6 + 36 * k;
      const double* wa = wg + kb;
      const double* jg = wa + kb;
      if (out->J_bias_g || out->J_bias_a)
        for (int row = 0; row < 6; ++row)
          for (int j = 0; j < kb; ++j)
            for (int c = 0; c < 3; ++c) {
              if (out->J_bias_g) out->J_bias_g[(size_t(i) * 6 + row) * 3 * kb + 3 * j + c] = (row == c) ? wg[j] : 0.0;
              if (out->J_bias_a) out->J_bias_a[(size_t(i) * 6 + row) * 3 * kb + 3 * j + c] = (row == 3 + c) ? wa[j] : 0.0;
            }
      if (out->J_gravity) std::memcpy(out->J_gravity + size_t(i) * 12, jg, 96);
      if (out->cost) out->cost[i] = cost[d];
      if (out->first_cp) out->first_cp[i] = p->in_first[d];
      if (out->first_bias) out->first_bias[i] = p->in_first_bias[d];
    }
    return HS_OK;
  }
}

int hs_cost(hs_problem* p, double* cost) {
  if (!p || !cost) return HS_ERR_INVALID;
  if (p->has_weights()) HS_FAIL(HS_ERR_INVALID, kWeightsMessage);
  int rc = prepare(p);
  if (rc) return rc;
  rc = reset_state(p, 0, 1e4);
  if (rc) return rc;
  HS_ORDER_SWITCH(p->k, rc = launch_linearize<K>(p, false, true));
  if (rc) return rc;
  k_pack_exchange<<<1, kBlock, 0, p->stream>>>(p->T, 0);
  rc = exchange(p, p->T.xbuf + p->T.xo_cost, 1);
  if (rc) return rc;
  k_cost_reduce<<<1, kBlock, 0, p->stream>>>(p->T);
  HIP_TRY(hipGetLastError());
  DevState st;
  HIP_TRY(hipMemcpyAsync(&st, p->d_state.p, sizeof(st), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  *cost = st.cost;
  return guard_check(p);
}

int hs_reduced_system(hs_problem* p, double radius, double* S, double* g) {
  if (!p || !S || !g) return HS_ERR_INVALID;
  if (p->has_weights()) HS_FAIL(HS_ERR_INVALID, kWeightsMessage);
  int rc = prepare(p);
  if (rc) return rc;
  rc = reset_state(p, 1, radius);
  if (rc) return rc;
  HS_ORDER_SWITCH(p->k, rc = launch_linearize<K>(p));
  if (rc) return rc;
  HS_ORDER_SWITCH(p->k, rc = launch_build<K>(p));
  if (rc) return rc;
  const int np = p->T.np, ncb = 6 * p->T.bw;
  std::vector<double> Sb(size_t(np) * ncb), gs(np);
  HIP_TRY(hipMemcpyAsync(Sb.data(), p->d_Sb.p, Sb.size() * 8, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipMemcpyAsync(gs.data(), p->d_g_s.p, gs.size() * 8, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  const int dim = hs_dim_pose(p);
  std::memset(S, 0, sizeof(double) * size_t(dim) * dim);
  for (int rho = 0; rho < np; ++rho) {
    const int c0 = 6 * (rho / 6);
    for (int c = 0; c < ncb && c0 + c < np; ++c) {
      const double v = Sb[size_t(rho) * ncb + c];
      if (c0 + c >= rho) S[size_t(rho) * dim + c0 + c] = v, S[size_t(c0 + c) * dim + rho] = v;
    }
    g[rho] = gs[rho];
  }
  if (p->T.nb) {
    const int nb = p->T.nb;
    std::vector<double> Spb(size_t(np) * nb), Sbb(size_t(nb) * nb), gb(nb);
    HIP_TRY(hipMemcpyAsync(Spb.data(), p->d_Spb.p, Spb.size() * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(Sbb.data(), p->d_Sbb.p, Sbb.size() * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(gb.data(), p->d_gb_s.p, gb.size() * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int rho = 0; rho < np; ++rho)
      for (int b = 0; b < nb; ++b) S[size_t(rho) * dim + np + b] = S[size_t(np + b) * dim + rho] = Spb[size_t(rho) * nb + b];
    for (int a = 0; a < nb; ++a) {
      for (int b = 0; b < nb; ++b) S[size_t(np + a) * dim + np + b] = Sbb[size_t(a) * nb + b];
      g[np + a] = gb[a];
    }
  }
  return guard_check(p);
}

int hs_set_weights(hs_problem* p, int type, const double* weights) {
  if (!p) return HS_ERR_INVALID;
  if (type < HS_PIXEL || type > HS_INERTIAL) HS_FAIL(HS_ERR_INVALID, "unknown factor type");
  const int nr = type == HS_PIXEL ? 2 : (type == HS_BEARING ? 1 : 6);
  if (weights)
    p->weights[type].assign(weights, weights + nr * nr);
  else
    p->weights[type].clear();
  return HS_OK;
}

int hs_set_guard(int enabled) {
  GuardRegistry::flag() = enabled != 0 ? 1 : 0;
  return HS_OK;
}

int hs_set_stage_timing(hs_problem* p, int enabled) {
  if (!p) return HS_ERR_INVALID;
  p->stage_timing = enabled != 0;
  return HS_OK;
}

int hs_solve(hs_problem* p, int max_iterations, hs_summary* summary, hs_iteration* iterations) {
  if (!p || !summary) return HS_ERR_INVALID;
  if (p->has_weights()) HS_FAIL(HS_ERR_INVALID, kWeightsMessage);
  if (max_iterations < 0 || max_iterations > kMaxIterations) HS_FAIL(HS_ERR_INVALID, "max_iterations out of range");
  p->results_cached = false;
  int rc = prepare(p);
  if (rc) return rc;
  const bool spec = speculative_solve(p);
  // (fused visual-only windows defer also when they are small enough for the decision kernel to commit inline: deferred landmarks are what lets the
  //  decision ride in the next build — one launch less per iteration, which at the replay's window sizes is 6 us of 130)
  const bool deferred = (fused_visual_only(p) || (spec && !commit_inline(p))) && !(p->T.debug_flags & 67108864);  // A/B switch 67108864: k_commit in every iteration
  const bool fold = fold_decision_into_build(p, deferred);  // the decision of iteration i rides in k_build_visual of iteration i + 1
  rc = reset_state(p, max_iterations, 1e4, spec ? (deferred ? 2 : 1) : (deferred ? 4 : 0));
  if (rc) return rc;
  hipStream_t s = p->stream;
  const auto host_t3 = std::chrono::steady_clock::now();
  // stage timing (optional, hs_set_stage_timing): 4 stages per iteration bracketed by HIP events on the launch stream; every event is a
  // barrier packet (~5.7 us of idle device each, rocprofv3 kernel trace), so by default only the two ends of the solve are stamped
  const bool stages = p->stage_timing;
  const size_t n_ev = size_t(6) * max_iterations + 1;  // (+ two per iteration around the candidate's linearisation of a speculative solve)
  const size_t ev_cand = size_t(4) * max_iterations + 1;
  while (p->events.size() < n_ev) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    p->events.push_back(e);
  }
  std::vector<hipEvent_t>& ev = p->events;
  HIP_TRY(hipEventRecord(ev[0], s));
  for (int it = 0; it < max_iterations; ++it) {
    if (it == 0 || !spec) {  // (speculative solves: the linearisation of the current point came with the previous iteration's candidate)
      HS_ORDER_SWITCH(p->k, rc = launch_linearize<K>(p, true));
      if (rc) return rc;
    }
    if (stages && !p->fused) HIP_TRY(hipEventRecord(ev[4 * it + 1], s));
    hipEvent_t after_build = stages && p->fused ? ev[4 * it + 1] : nullptr;  // fused build: the linearise stage ends behind k_build_visual
    HS_ORDER_SWITCH(p->k, rc = launch_build<K>(p, after_build, it > 0, it > 0 && fold));
    if (rc) return rc;
    if (stages) HIP_TRY(hipEventRecord(ev[4 * it + 2], s));
    rc = launch_factor(p);
    if (rc) return rc;
    if (stages) HIP_TRY(hipEventRecord(ev[4 * it + 3], s));
    const bool lin_cand = spec && it + 1 < max_iterations;
    hipEvent_t* lin_ev = stages && lin_cand ? &ev[ev_cand + 2 * it] : nullptr;
    HS_ORDER_SWITCH(p->k, rc = launch_update<K>(p, lin_cand, deferred, lin_ev, fold && it + 1 < max_iterations));
    if (rc) return rc;
    if (deferred && it + 1 == max_iterations) launch_commit(p);  // the last accepted candidate (also when a convergence test ended the solve early)
    if (stages || it + 1 == max_iterations) HIP_TRY(hipEventRecord(ev[4 * it + 4], s));
  }
  if (max_iterations == 0) {
    HS_ORDER_SWITCH(p->k, rc = launch_linearize<K>(p, false, true));
    if (rc) return rc;
    k_pack_exchange<<<1, kBlock, 0, s>>>(p->T, 0);
    rc = exchange(p, p->T.xbuf + p->T.xo_cost, 1);
    if (rc) return rc;
    k_cost_reduce<<<1, kBlock, 0, s>>>(p->T);
  }
  DevState& st = *p->h_state;
  const auto host_t4 = std::chrono::steady_clock::now();
  if (p->want_results) {  // [cp | lm (device order) | bias_g | bias_a | gravity] -> pinned host memory, behind the last kernel
    const size_t n_cp8 = p->cp.size(), n_lm3 = p->lm.size(), n_b = p->has_imu ? p->bias_g.size() : 0, total = n_cp8 + n_lm3 + 2 * n_b + 3;
    if (total > p->h_result_cap) {
      if (p->h_result) (void)hipHostFree(p->h_result);
      p->h_result = nullptr, p->h_result_cap = 0;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->h_result), 2 * total * sizeof(double), hipHostMallocDefault));
      p->h_result_cap = 2 * total;
    }
    double* h = p->h_result;
    HIP_TRY(hipMemcpyAsync(h, p->d_cp.p, n_cp8 * 8, hipMemcpyDeviceToHost, s));
    if (n_lm3) HIP_TRY(hipMemcpyAsync(h + n_cp8, p->d_lm.p, n_lm3 * 8, hipMemcpyDeviceToHost, s));
    if (n_b) {
      HIP_TRY(hipMemcpyAsync(h + n_cp8 + n_lm3, p->d_bias_g.p, n_b * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(h + n_cp8 + n_lm3 + n_b, p->d_bias_a.p, n_b * 8, hipMemcpyDeviceToHost, s));
    }
    if (p->has_imu) HIP_TRY(hipMemcpyAsync(h + n_cp8 + n_lm3 + 2 * n_b, p->d_gravity.p, 24, hipMemcpyDeviceToHost, s));  // also with an empty bias table
  }
  HIP_TRY(hipMemcpyAsync(&st, p->d_state.p, sizeof(st), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (max_iterations > 0) p->device_ahead = true;
  p->results_cached = p->want_results;  // only once every copy above has completed: a failed copy or synchronisation leaves the getters on the device path
  if (p->host_timing) {
    const auto host_t5 = std::chrono::steady_clock::now();
    const double tl = std::chrono::duration<double, std::milli>(host_t4 - host_t3).count(), tw = std::chrono::duration<double, std::milli>(host_t5 - host_t4).count();
    p->host_ms[2] += tl, p->host_ms[3] += tw;
    p->host_log.insert(p->host_log.end(), {p->host_prepare[0], p->host_prepare[1], tl, tw});
    p->host_prepare[0] = p->host_prepare[1] = 0;
    ++p->host_calls;
  }
  std::memset(summary, 0, sizeof(*summary));
  summary->initial_cost = st.records[0].cost;
  summary->final_cost = st.cost;
  summary->num_iterations = st.num_iterations;
  summary->num_successful_steps = st.num_successful;
  summary->termination = st.termination;
  summary->num_residual_blocks = p->T.n_vis + p->T.n_pri + p->T.n_ine;
  if (!stages) summary->linearize_ms = summary->schur_ms = summary->solve_ms = summary->update_ms = -1.0;  // not measured (hs_set_stage_timing)
  for (int it = 0; stages && it < max_iterations; ++it) {
    float t[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) (void)hipEventElapsedTime(&t[k], ev[4 * it + k], ev[4 * it + k + 1]);
    summary->linearize_ms += t[0], summary->schur_ms += t[1], summary->solve_ms += t[2], summary->update_ms += t[3];
    if (spec && it + 1 < max_iterations) {  // the linearisation of this iteration's candidate = the next iteration's linearisation when accepted
      float tc = 0;
      (void)hipEventElapsedTime(&tc, ev[ev_cand + 2 * it], ev[ev_cand + 2 * it + 1]);
      summary->linearize_ms += tc, summary->update_ms -= tc;
    }
  }
  if (max_iterations > 0) {
    float t = 0;
    (void)hipEventElapsedTime(&t, ev[0], ev[4 * max_iterations]);
    summary->total_ms = t;
  }
  if (iterations) {
    std::memset(iterations, 0, sizeof(hs_iteration) * (size_t(max_iterations) + 1));
    const int n = std::min(st.num_iterations, max_iterations);
    for (int i = 0; i <= n; ++i) iterations[i] = st.records[i];
  }
  if (st.chol_failed && st.termination == HS_FAILURE)
    p->err = st.chol_failed == 2 ? "two-ended solve: the partner workgroup did not arrive within 2 s" : "reduced system not positive definite";
  return guard_check(p);
}

#if HS_PROFILE_HOOKS
/// Profiling builds only (hipcc -DHS_PROFILE_HOOKS=1, tools/build_profiling_lib.sh): the phase timestamps the kernels wrote under
/// HS_DEBUG_FLAGS 16 / 32. Not part of the C ABI: the product library neither contains the hooks nor exports this function.
extern "C" int hs_debug_read(hs_problem* p, double* dst, int n) {
  if (!p || !dst) return HS_ERR_INVALID;
  HIP_TRY(hipMemcpyAsync(dst, p->d_xpart.p, size_t(n) * 8, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return HS_OK;
}
#endif

int hs_snapshot(hs_problem* p) {
  if (!p) return HS_ERR_INVALID;
  int rc = prepare(p);
  if (rc) return rc;
  HIP_TRY(p->d_cp_snap.reserve(p->cp.size()));
  HIP_TRY(p->d_lm_snap.reserve(p->lm.size() + 1));
  HIP_TRY(hipMemcpyAsync(p->d_cp_snap.p, p->d_cp.p, p->cp.size() * 8, hipMemcpyDeviceToDevice, p->stream));
  if (p->n_lm) HIP_TRY(hipMemcpyAsync(p->d_lm_snap.p, p->d_lm.p, p->lm.size() * 8, hipMemcpyDeviceToDevice, p->stream));
  if (p->has_imu) {
    HIP_TRY(p->d_bias_g_snap.reserve(p->bias_g.size() + 1));
    HIP_TRY(p->d_bias_a_snap.reserve(p->bias_a.size() + 1));
    HIP_TRY(p->d_gravity_snap.reserve(3));
    HIP_TRY(hipMemcpyAsync(p->d_bias_g_snap.p, p->d_bias_g.p, p->bias_g.size() * 8, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_bias_a_snap.p, p->d_bias_a.p, p->bias_a.size() * 8, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_gravity_snap.p, p->d_gravity.p, 24, hipMemcpyDeviceToDevice, p->stream));
  }
  p->has_snapshot = true;
  return HS_OK;
}

int hs_restore(hs_problem* p) {
  if (!p) return HS_ERR_INVALID;
  if (!p->has_snapshot || p->dirty) HS_FAIL(HS_ERR_STATE, "hs_restore without a valid hs_snapshot");
  p->results_cached = false;
  HIP_TRY(hipMemcpyAsync(p->d_cp.p, p->d_cp_snap.p, p->cp.size() * 8, hipMemcpyDeviceToDevice, p->stream));
  if (p->n_lm) HIP_TRY(hipMemcpyAsync(p->d_lm.p, p->d_lm_snap.p, p->lm.size() * 8, hipMemcpyDeviceToDevice, p->stream));
  if (p->has_imu) {
    HIP_TRY(hipMemcpyAsync(p->d_bias_g.p, p->d_bias_g_snap.p, p->bias_g.size() * 8, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_bias_a.p, p->d_bias_a_snap.p, p->bias_a.size() * 8, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_gravity.p, p->d_gravity_snap.p, 24, hipMemcpyDeviceToDevice, p->stream));
  }
  return HS_OK;
}

int hs_rccl_unique_id(char id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!id) return HS_ERR_INVALID;
  RcclApi* api = rccl_api();
  if (!api) return HS_ERR_DEVICE;
  ncclUniqueId u;
  if (api->GetUniqueId(&u) != ncclSuccess) return HS_ERR_DEVICE;
  std::memcpy(id, &u, 128);
  return HS_OK;
}

int hs_rccl_init(hs_problem* p, const char id[128], int rank, int world) {
  if (!p || !id || world < 1 || rank < 0 || rank >= world) return HS_ERR_INVALID;
  RcclApi* api = rccl_api();
  if (!api) HS_FAIL(HS_ERR_DEVICE, "librccl.so could not be loaded");
  HIP_TRY(hipSetDevice(p->device));
  if (p->rccl_comm) {  // a second initialisation replaces the communicator instead of leaking it
    (void)api->CommDestroy(static_cast<ncclComm_t>(p->rccl_comm));
    p->rccl_comm = nullptr;
  }
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  ncclComm_t comm = nullptr;
  const ncclResult_t r = api->CommInitRank(&comm, world, u, rank);
  if (r != ncclSuccess) HS_FAIL(HS_ERR_DEVICE, std::string("ncclCommInitRank failed: ") + (api->GetErrorString ? api->GetErrorString(r) : "?"));
  p->rccl_comm = comm;
  return HS_OK;
}

int hs_rccl_shutdown(hs_problem* p) {
  if (!p) return HS_ERR_INVALID;
  if (p->rccl_comm && rccl_api()) {
    (void)hipStreamSynchronize(p->stream);
    (void)rccl_api()->CommDestroy(static_cast<ncclComm_t>(p->rccl_comm));
  }
  p->rccl_comm = nullptr;
  return HS_OK;
}

int hs_exchange_info(hs_problem* p, int32_t* rccl_ranks, int64_t* doubles_per_linearisation, int64_t* doubles_per_decision) {
  if (!p) return HS_ERR_INVALID;
  if (rccl_ranks) {
    int n = 0;
    if (p->rccl_comm && rccl_api() && rccl_api()->CommCount) {
      const ncclResult_t r = rccl_api()->CommCount(static_cast<ncclComm_t>(p->rccl_comm), &n);
      if (r != ncclSuccess) HS_FAIL(HS_ERR_DEVICE, "ncclCommCount failed");
    }
    *rccl_ranks = n;
  }
  if (doubles_per_linearisation) *doubles_per_linearisation = p->dirty ? 0 : p->T.x_count1;
  if (doubles_per_decision) *doubles_per_decision = 5;
  return HS_OK;
}

int hs_set_allreduce(hs_problem* p, hs_allreduce_fn fn, void* user) {
  if (!p) return HS_ERR_INVALID;
  p->allreduce = fn, p->allreduce_user = user;
  return HS_OK;
}

int hs_set_shard(hs_problem* p, int rank, int world, int min_band_blocks) {
  if (!p) return HS_ERR_INVALID;
  if (world < 1 || rank < 0 || rank >= world || min_band_blocks < 0) HS_FAIL(HS_ERR_INVALID, "bad shard description");
  p->rank = rank, p->world = world, p->min_bw = min_band_blocks;
  p->touch(hs_problem::kAll);
  return HS_OK;
}

int hs_band_blocks(hs_problem* p) {
  if (!p) return -1;
  if (prepare(p) != HS_OK) return -1;
  return p->T.bw;
}

int hs_get_control_points(hs_problem* p, double* cp) {
  if (!p || !cp) return HS_ERR_INVALID;
  if (p->dirty) {
    std::memcpy(cp, p->cp.data(), p->cp.size() * 8);
    return HS_OK;
  }
  if (p->results_cached) {
    std::memcpy(cp, p->h_result, p->cp.size() * 8);
  } else {
    p->want_results = true;
    HIP_TRY(hipMemcpyAsync(cp, p->d_cp.p, size_t(8) * p->n_cp * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  std::memcpy(p->cp.data(), cp, p->cp.size() * 8);
  return HS_OK;
}
int hs_get_landmarks(hs_problem* p, double* xyz) {
  if (!p || !xyz) return HS_ERR_INVALID;
  if (p->dirty || p->n_lm == 0) {
    std::memcpy(xyz, p->lm.data(), p->lm.size() * 8);
    return HS_OK;
  }
  std::vector<double> fetched;
  const double* dev = p->h_result + p->cp.size();
  if (!p->results_cached) {
    p->want_results = true;
    fetched.resize(size_t(3) * p->n_lm);
    HIP_TRY(hipMemcpyAsync(fetched.data(), p->d_lm.p, fetched.size() * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    dev = fetched.data();
  }
  for (int d = 0; d < p->n_lm; ++d) {
    const int t = p->vs.table_of_dev[d];
    for (int c = 0; c < 3; ++c) xyz[3 * t + c] = p->lm[3 * t + c] = dev[3 * d + c];
  }
  return HS_OK;
}
int hs_get_bias(hs_problem* p, double* bg, double* ba) {
  if (!p || !bg || !ba) return HS_ERR_INVALID;
  if (!p->dirty && p->has_imu && !p->bias_g.empty()) {
    if (p->results_cached) {
      const double* h = p->h_result + p->cp.size() + p->lm.size();
      std::memcpy(p->bias_g.data(), h, p->bias_g.size() * 8), std::memcpy(p->bias_a.data(), h + p->bias_g.size(), p->bias_a.size() * 8);
    } else {
      p->want_results = true;
      HIP_TRY(hipMemcpyAsync(p->bias_g.data(), p->d_bias_g.p, p->bias_g.size() * 8, hipMemcpyDeviceToHost, p->stream));
      HIP_TRY(hipMemcpyAsync(p->bias_a.data(), p->d_bias_a.p, p->bias_a.size() * 8, hipMemcpyDeviceToHost, p->stream));
      HIP_TRY(hipStreamSynchronize(p->stream));
    }
  }
  std::memcpy(bg, p->bias_g.data(), p->bias_g.size() * 8), std::memcpy(ba, p->bias_a.data(), p->bias_a.size() * 8);
  return HS_OK;
}
int hs_get_gravity(hs_problem* p, double* g) {
  if (!p || !g) return HS_ERR_INVALID;
  if (!p->dirty && p->has_imu) {
    if (p->results_cached) {
      std::memcpy(p->gravity, p->h_result + p->cp.size() + p->lm.size() + 2 * p->bias_g.size(), 24);
    } else {
      p->want_results = true;
      HIP_TRY(hipMemcpyAsync(p->gravity, p->d_gravity.p, 24, hipMemcpyDeviceToHost, p->stream));
      HIP_TRY(hipStreamSynchronize(p->stream));
    }
  }
  std::memcpy(g, p->gravity, 24);
  return HS_OK;
}

// Left inverse of EigenQuaternionManifold's PlusJacobian (orthonormal columns e_i (x) q): J_ambient = J_local * P^T.
static void quat_plus_jacobian_T(const double* q, double* PT /* 3 x 4 */) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  // column i of P = (e_i, 0) (x) q  (Hamilton, xyzw)
  const double P[4][3] = {{w, z, -y}, {-z, w, x}, {y, -x, w}, {-x, -y, -z}};
  for (int i = 0; i < 3; ++i)
    for (int r = 0; r < 4; ++r) PT[4 * i + r] = P[r][i];
}

int hs_cost_function_evaluate(hs_problem* p, int type, int idx, const double* const* parameters, double* residuals, double** jacobians) {
  if (!p || !parameters || !residuals) return HS_ERR_INVALID;
  if (type < 0 || type > 3 || idx < 0 || idx >= hs_num_residuals(p, type)) HS_FAIL(HS_ERR_INVALID, "residual index out of range");
  const int k = p->k, kb = p->kb;
  const BlockLayout L = make_block_layout(type, k, kb);
  // one-residual window at the given parameter values, on a scratch handle that lives as long as p (same device, same stream)
  if (!p->scratch) {
    const int rc0 = hs_create(p->device, p->stream, &p->scratch);
    if (rc0) HS_FAIL(rc0, "hs_cost_function_evaluate: scratch handle");
  }
  hs_problem* q = p->scratch;
  q->inertial_mode = p->inertial_mode;
  for (int t = 0; t < 4; ++t) q->weights[t] = p->weights[t];
  // tables of the previous call (possibly another factor type)
  q->px_stamp.clear(), q->px_meas.clear(), q->px_lm.clear(), q->px_cam.clear(), q->br_stamp.clear(), q->br_meas.clear(), q->br_lm.clear(), q->br_cam.clear();
  q->pr_stamp.clear(), q->pr_meas.clear(), q->pr_sensor.clear(), q->in_stamp.clear(), q->in_meas.clear();
  q->has_imu = false, q->n_bias = 0, q->bias_g.clear(), q->bias_a.clear(), q->n_lm = 0, q->lm.clear(), q->lm_const.clear();
  q->touch(hs_problem::kAll);
  int rc = HS_OK;
  std::vector<double> cps(size_t(8) * k);
  for (int j = 0; j < k; ++j) std::memcpy(&cps[8 * j], parameters[j], 64);
  const double t0 = cps[7], dt = cps[15] - cps[7];
  rc = hs_set_spline(q, k, t0, dt, k, cps.data(), nullptr, 0, 0);
  int32_t zero = 0;
  double stamp = 0;
  int nres = L.num_residuals;
  if (!rc && (type == HS_PIXEL || type == HS_BEARING)) {
    rc = hs_set_cameras(q, 1, parameters[k], parameters[k + 1], parameters[k + 2]);
    if (!rc) rc = hs_set_landmarks(q, 1, parameters[k + 3], nullptr);
    stamp = type == HS_PIXEL ? p->px_stamp[idx] : p->br_stamp[idx];
    if (!rc) rc = type == HS_PIXEL ? hs_set_pixel_residuals(q, 1, &stamp, &p->px_meas[2 * idx], &zero, &zero)
                                   : hs_set_bearing_residuals(q, 1, &stamp, &p->br_meas[3 * idx], &zero, &zero);
  } else if (!rc && type == HS_PRIOR) {
    rc = hs_set_sensors(q, 1, parameters[k]);
    stamp = p->pr_stamp[idx];
    if (!rc) rc = hs_set_prior_residuals(q, 1, &stamp, &p->pr_meas[7 * idx], &zero);
  } else if (!rc) {
    std::vector<double> bg(size_t(4) * kb), ba(size_t(4) * kb);
    for (int j = 0; j < kb; ++j) std::memcpy(&bg[4 * j], parameters[k + 5 + j], 32), std::memcpy(&ba[4 * j], parameters[k + 5 + kb + j], 32);
    rc = hs_set_imu(q, parameters[k], parameters[k + 1], parameters[k + 2], parameters[k + 3], parameters[k + 4], kb, bg[3], bg[7] - bg[3], kb,
                    bg.data(), ba.data(), 0);
    if (!rc) rc = hs_set_gravity(q, parameters[k + 5 + 2 * kb], 0);
    stamp = p->in_stamp[idx];
    if (!rc) rc = hs_set_inertial_residuals(q, 1, &stamp, &p->in_meas[6 * idx]);
  }
  if (rc) HS_FAIL(rc, std::string("hs_cost_function_evaluate: ") + hs_last_error(q));
  std::vector<double> r(6), Js(size_t(6) * 6 * k), Jl(18), Jbg(size_t(18) * kb), Jba(size_t(18) * kb), Jg(12);
  double Jext[36], Jintr[8], Jdist[8], Jig[36], Jia[36], JSg[54], JXa[54];
  hs_linearization lin;
  std::memset(&lin, 0, sizeof(lin));
  lin.r = r.data(), lin.J_state = Js.data(), lin.J_landmark = Jl.data(), lin.J_bias_g = Jbg.data(), lin.J_bias_a = Jba.data(), lin.J_gravity = Jg.data();
  if (jacobians) {  // sensor blocks (static_sensor_idx ..): only when asked for — a pass of their own
    const int s0 = L.indices[1];
    if (jacobians[s0]) lin.J_extrinsics = Jext;
    if (type == HS_PIXEL) {
      if (jacobians[s0 + 1]) lin.J_intrinsics = Jintr;
      if (jacobians[s0 + 2]) lin.J_distortion = Jdist;
    } else if (type == HS_INERTIAL) {
      if (jacobians[s0 + 1]) lin.J_gyro_intrinsics = Jig;
      if (jacobians[s0 + 2]) lin.J_acc_intrinsics = Jia;
      if (jacobians[s0 + 3]) lin.J_gyro_sensitivity = JSg;
      if (jacobians[s0 + 4]) lin.J_acc_offsets = JXa;
    }
  }
  rc = hs_linearize(q, type, /*robustify=*/0, &lin);
  if (rc) HS_FAIL(rc, std::string("hs_cost_function_evaluate: ") + hs_last_error(q));
  for (int i = 0; i < nres; ++i) residuals[i] = r[i];
  if (!jacobians) return HS_OK;
  for (int j = 0; j < k; ++j) {  // Stamped<SE3>: [q(4) p(3) t(1)], local [rot(3) trans(3)]
    if (!jacobians[j]) continue;
    double PT[12];
    quat_plus_jacobian_T(parameters[j], PT);
    for (int row = 0; row < nres; ++row) {
      const double* jl = &Js[(size_t(row) * k + j) * 6];
      double* out = jacobians[j] + size_t(row) * 8;
      for (int c = 0; c < 4; ++c) out[c] = jl[0] * PT[c] + jl[1] * PT[4 + c] + jl[2] * PT[8 + c];
      out[4] = jl[3], out[5] = jl[4], out[6] = jl[5], out[7] = 0.0;
    }
  }
  if (double* out = jacobians[k]) {  // sensor extrinsics SE3 [q(4) p(3)]: ambient = local * P^T as for the control points
    double PT[12];
    quat_plus_jacobian_T(parameters[k], PT);
    for (int row = 0; row < nres; ++row) {
      const double* jl = &Jext[6 * row];
      for (int c = 0; c < 4; ++c) out[7 * row + c] = jl[0] * PT[c] + jl[1] * PT[4 + c] + jl[2] * PT[8 + c];
      out[7 * row + 4] = jl[3], out[7 * row + 5] = jl[4], out[7 * row + 6] = jl[5];
    }
  }
  if (type == HS_PIXEL || type == HS_BEARING) {
    // intrinsics / distortion: Euclidean blocks; the bearing evaluator leaves them zero (bearing.cpp:58-77)
    if (jacobians[k + 1])
      for (int e = 0; e < nres * 4; ++e) jacobians[k + 1][e] = type == HS_PIXEL ? Jintr[e] : 0.0;
    if (jacobians[k + 2])
      for (int e = 0; e < nres * 4; ++e) jacobians[k + 2][e] = type == HS_PIXEL ? Jdist[e] : 0.0;
    if (jacobians[k + 3])
      for (int e = 0; e < nres * 3; ++e) jacobians[k + 3][e] = Jl[e];
  } else if (type == HS_INERTIAL) {
    if (jacobians[k + 1]) std::memcpy(jacobians[k + 1], Jig, sizeof(Jig));
    if (jacobians[k + 2]) std::memcpy(jacobians[k + 2], Jia, sizeof(Jia));
    if (jacobians[k + 3]) std::memcpy(jacobians[k + 3], JSg, sizeof(JSg));
    if (jacobians[k + 4]) std::memcpy(jacobians[k + 4], JXa, sizeof(JXa));
    for (int j = 0; j < kb; ++j)
      for (int part = 0; part < 2; ++part) {
        double* out = jacobians[k + 5 + part * kb + j];
        if (!out) continue;
        const std::vector<double>& J = part ? Jba : Jbg;
        for (int row = 0; row < 6; ++row) {
          for (int c = 0; c < 3; ++c) out[row * 4 + c] = J[(size_t(row) * kb + j) * 3 + c];
          out[row * 4 + 3] = 0.0;
        }
      }
    if (double* out = jacobians[k + 5 + 2 * kb]) {  // gravity: left inverse of the SphereManifold PlusJacobian = P^T / |x|^2
      const double* x = parameters[k + 5 + 2 * kb];
      // Householder basis as in Ceres (same code path as the device: recompute on the host)
      double v[3] = {x[0], x[1], 1.0}, beta = 0.0;
      const double sigma = x[0] * x[0] + x[1] * x[1];
      if (sigma <= 2.220446049250313e-16) {
        if (x[2] < 0) beta = 2.0;
      } else {
        const double mu = std::sqrt(x[2] * x[2] + sigma);
        const double vp = (x[2] <= 0.0) ? (x[2] - mu) : (-sigma / (x[2] + mu));
        beta = 2.0 * vp * vp / (sigma + vp * vp);
        v[0] /= vp, v[1] /= vp;
      }
      const double nx2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], nx = std::sqrt(nx2);
      double P[6];
      for (int i = 0; i < 2; ++i)
        for (int rr = 0; rr < 3; ++rr) P[rr * 2 + i] = nx * ((rr == i ? 1.0 : 0.0) - beta * v[rr] * v[i]);
      for (int row = 0; row < 6; ++row)
        for (int c = 0; c < 3; ++c) out[row * 3 + c] = (Jg[row * 2] * P[c * 2] + Jg[row * 2 + 1] * P[c * 2 + 1]) / nx2;
    }
  }
  return HS_OK;
}

/// Evaluation-only device tables (spline + cameras) for the entry points that do not touch residuals: the residual tables of the
/// last solve may refer to control points that the sliding window has already dropped, so prepare() is not run here.
static int eval_tables(hs_problem* p, Tables* T, DBuf<double>* d_cp, DBuf<double>* d_cam) {
  if (p->n_cp < p->k || p->cp.empty()) HS_FAIL(HS_ERR_STATE, "hs_set_spline has not been called");
  hipStream_t s = p->stream;
  HIP_TRY(d_cp->upload(p->cp, s));
  if (!p->cam.empty()) HIP_TRY(d_cam->upload(p->cam, s));
  std::memset(T, 0, sizeof(*T));
  T->sp = Spline{p->k, p->n_cp, p->t0, p->dt, 1.0 / p->dt, p->rot_const, p->trans_const};
  T->basis = make_basis_coef(p->k);
  T->cp = d_cp->p, T->cam = d_cam->p;
  return HS_OK;
}

int hs_process_tracks(hs_problem* p, double stamp, int n, const double* pixels0, const double* pixels1, double* bearings0, double* bearings1,
                      double* positions_w) {
  if (!p || n < 0 || (n && (!pixels0 || !pixels1))) return HS_ERR_INVALID;
  if (p->cam.size() < 32) HS_FAIL(HS_ERR_STATE, "hs_process_tracks needs a stereo pair (two cameras)");
  if (n == 0) return HS_OK;
  Tables T;
  DBuf<double> d_cp, d_cam;
  int rc = eval_tables(p, &T, &d_cp, &d_cam);
  if (rc) return rc;
  const int k = p->k, n_seg = p->n_cp - k + 1;
  const int f = h_segment_first(stamp, p->t0, p->dt, k);
  if (positions_w && (f < 0 || f >= n_seg)) HS_FAIL(HS_ERR_INVALID, "stamp outside the valid range of the spline");
  hipStream_t s = p->stream;
  DBuf<double> d_p0, d_p1, d_b0, d_b1, d_pw;
  HIP_TRY(d_p0.upload(std::vector<double>(pixels0, pixels0 + 2 * size_t(n)), s));
  HIP_TRY(d_p1.upload(std::vector<double>(pixels1, pixels1 + 2 * size_t(n)), s));
  if (bearings0) HIP_TRY(d_b0.reserve(size_t(3) * n));
  if (bearings1) HIP_TRY(d_b1.reserve(size_t(3) * n));
  if (positions_w) HIP_TRY(d_pw.reserve(size_t(3) * n));
  const int nb = (n + kBlock - 1) / kBlock;
  HS_ORDER_SWITCH(k, k_process_tracks<K><<<nb, kBlock, cp_lds_bytes(p), s>>>(T, stamp, n, d_p0.p, d_p1.p, bearings0 ? d_b0.p : nullptr, bearings1 ? d_b1.p : nullptr,
                                                           positions_w ? d_pw.p : nullptr));
  HIP_TRY(hipGetLastError());
  if (bearings0) HIP_TRY(hipMemcpyAsync(bearings0, d_b0.p, size_t(3) * n * 8, hipMemcpyDeviceToHost, s));
  if (bearings1) HIP_TRY(hipMemcpyAsync(bearings1, d_b1.p, size_t(3) * n * 8, hipMemcpyDeviceToHost, s));
  if (positions_w) HIP_TRY(hipMemcpyAsync(positions_w, d_pw.p, size_t(3) * n * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return HS_OK;
}

int hs_manifold_tangent_size(int kind, int ambient) {
  switch (kind) {
    case HS_MANIFOLD_CONSTANT: return (ambient >= 1 && ambient <= 9) ? 0 : -1;
    case HS_MANIFOLD_EUCLIDEAN: return (ambient >= 1 && ambient <= 9) ? ambient : -1;
    case HS_MANIFOLD_CONTROL_POINT: return ambient == 8 ? 6 : -1;
    case HS_MANIFOLD_SE3: return ambient == 7 ? 6 : -1;
    case HS_MANIFOLD_SPHERE3: return ambient == 3 ? 2 : -1;
    case HS_MANIFOLD_BIAS_POINT: return ambient == 4 ? 3 : -1;
    default: return -1;
  }
}

static int manifold_launch(hs_problem* p, int kind, int ambient, int n, const double* x, const double* delta, double* out, double* jac) {
  if (!p || n < 0 || (n && !x)) return HS_ERR_INVALID;
  const int tangent = hs_manifold_tangent_size(kind, ambient);
  if (tangent < 0) HS_FAIL(HS_ERR_INVALID, "unknown manifold kind / ambient size");
  if (out && tangent > 0 && !delta) HS_FAIL(HS_ERR_INVALID, "delta is null");
  if (n == 0) return HS_OK;
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  DBuf<double> d_x, d_d, d_o, d_j;
  HIP_TRY(d_x.upload(std::vector<double>(x, x + size_t(n) * ambient), s));
  if (out && tangent > 0) HIP_TRY(d_d.upload(std::vector<double>(delta, delta + size_t(n) * tangent), s));
  if (out) HIP_TRY(d_o.reserve(size_t(n) * ambient));
  if (jac && tangent > 0) HIP_TRY(d_j.reserve(size_t(n) * ambient * tangent));
  k_manifold_plus<<<(n + kBlock - 1) / kBlock, kBlock, 0, s>>>(kind, ambient, tangent, n, d_x.p, d_d.p, out ? d_o.p : nullptr,
                                                             (jac && tangent > 0) ? d_j.p : nullptr);
  HIP_TRY(hipGetLastError());
  if (out) HIP_TRY(hipMemcpyAsync(out, d_o.p, size_t(n) * ambient * 8, hipMemcpyDeviceToHost, s));
  if (jac && tangent > 0) HIP_TRY(hipMemcpyAsync(jac, d_j.p, size_t(n) * ambient * tangent * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return HS_OK;
}

int hs_manifold_plus(hs_problem* p, int kind, int ambient, int n, const double* x, const double* delta, double* x_plus_delta) {
  if (n > 0 && !x_plus_delta) return HS_ERR_INVALID;
  return manifold_launch(p, kind, ambient, n, x, delta, x_plus_delta, nullptr);
}

int hs_manifold_plus_jacobian(hs_problem* p, int kind, int ambient, int n, const double* x, double* jacobian) {
  if (n > 0 && !jacobian) return HS_ERR_INVALID;
  return manifold_launch(p, kind, ambient, n, x, nullptr, nullptr, jacobian);
}

static int manifold_minus_launch(hs_problem* p, int kind, int ambient, int n, const double* y, const double* x, double* out, double* jac) {
  if (!p || n < 0 || (n && !x)) return HS_ERR_INVALID;
  const int tangent = hs_manifold_tangent_size(kind, ambient);
  if (tangent < 0) HS_FAIL(HS_ERR_INVALID, "unknown manifold kind / ambient size");
  if (out && !y) HS_FAIL(HS_ERR_INVALID, "y is null");
  if (n == 0 || tangent == 0) return HS_OK;
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  DBuf<double> d_x, d_y, d_o, d_j;
  HIP_TRY(d_x.upload(std::vector<double>(x, x + size_t(n) * ambient), s));
  if (out) HIP_TRY(d_y.upload(std::vector<double>(y, y + size_t(n) * ambient), s));
  if (out) HIP_TRY(d_o.reserve(size_t(n) * tangent));
  if (jac) HIP_TRY(d_j.reserve(size_t(n) * ambient * tangent));
  k_manifold_minus<<<(n + kBlock - 1) / kBlock, kBlock, 0, s>>>(kind, ambient, tangent, n, out ? d_y.p : nullptr, d_x.p, out ? d_o.p : nullptr, jac ? d_j.p : nullptr);
  HIP_TRY(hipGetLastError());
  if (out) HIP_TRY(hipMemcpyAsync(out, d_o.p, size_t(n) * tangent * 8, hipMemcpyDeviceToHost, s));
  if (jac) HIP_TRY(hipMemcpyAsync(jac, d_j.p, size_t(n) * ambient * tangent * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return HS_OK;
}

int hs_manifold_minus(hs_problem* p, int kind, int ambient, int n, const double* y, const double* x, double* y_minus_x) {
  if (n > 0 && hs_manifold_tangent_size(kind, ambient) > 0 && !y_minus_x) return HS_ERR_INVALID;
  return manifold_minus_launch(p, kind, ambient, n, y, x, y_minus_x, nullptr);
}

int hs_manifold_minus_jacobian(hs_problem* p, int kind, int ambient, int n, const double* x, double* jacobian) {
  if (n > 0 && hs_manifold_tangent_size(kind, ambient) > 0 && !jacobian) return HS_ERR_INVALID;
  return manifold_minus_launch(p, kind, ambient, n, nullptr, x, nullptr, jacobian);
}

int hs_sample_trajectory(hs_problem* p, int n, const double* stamps, double* pose, double* velocity, double* acceleration) {
  if (!p || n < 0 || (n && (!stamps || !pose))) return HS_ERR_INVALID;
  if (n == 0) return HS_OK;
  Tables T;
  DBuf<double> d_cp, d_cam;
  int rc = eval_tables(p, &T, &d_cp, &d_cam);
  if (rc) return rc;
  const int k = p->k, n_seg = p->n_cp - k + 1;
  for (int i = 0; i < n; ++i) {
    const int f = h_segment_first(stamps[i], p->t0, p->dt, k);
    if (f < 0 || f >= n_seg) HS_FAIL(HS_ERR_INVALID, "stamp outside the valid range of the spline");
  }
  hipStream_t s = p->stream;
  DBuf<double> d_st, d_pose, d_vel, d_acc;
  std::vector<double> st(stamps, stamps + n);
  HIP_TRY(d_st.upload(st, s));
  HIP_TRY(d_pose.reserve(size_t(7) * n));
  if (velocity) HIP_TRY(d_vel.reserve(size_t(6) * n));
  if (acceleration) HIP_TRY(d_acc.reserve(size_t(6) * n));
  const int nb = (n + kBlock - 1) / kBlock;
  HS_ORDER_SWITCH(k, k_sample_trajectory<K><<<nb, kBlock, cp_lds_bytes(p), s>>>(T, n, d_st.p, d_pose.p, velocity ? d_vel.p : nullptr, acceleration ? d_acc.p : nullptr));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(pose, d_pose.p, size_t(7) * n * 8, hipMemcpyDeviceToHost, s));
  if (velocity) HIP_TRY(hipMemcpyAsync(velocity, d_vel.p, size_t(6) * n * 8, hipMemcpyDeviceToHost, s));
  if (acceleration) HIP_TRY(hipMemcpyAsync(acceleration, d_acc.p, size_t(6) * n * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return HS_OK;
}

}  // extern "C"
