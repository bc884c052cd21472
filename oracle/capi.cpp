// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_math.hpp header). PARITY UNPINNED.
// C entry points of the CPU restatement, mirroring include/hyperslam_hip.h one-to-one with the prefix `hso_`
// so the parity tests drive both sides with the same tables.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../include/hyperslam_hip.h"
#include "hs_problem.hpp"

using namespace hso;

struct hso_problem {
  Problem P;
  std::string err;
  hs_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int rank = 0, world = 1;
};

static void attach(hso_problem* p, LM* lm) {
  lm->rank = p->rank, lm->world = p->world;
  if (p->allreduce) lm->allreduce = [p](double* buf, int64_t n) { p->allreduce(p->allreduce_user, buf, n, nullptr); };
}

#define CHECK_ARG(cond, msg)  \
  do {                        \
    if (!(cond)) {            \
      p->err = msg;           \
      return HS_ERR_INVALID;  \
    }                         \
  } while (0)

/// Same input validation (and message) as the product library: every residual stamp must have all its control points.
static int validate(hso_problem* p) {
  const Problem& P = p->P;
  const int n_seg = int(P.cp.size() / 8) - P.k + 1;
  double u;
  auto bad = [&](const std::vector<double>& st) {
    for (double t : st) {
      const int f = segment_of(t, P.t0, P.dt, P.k, &u);
      if (f < 0 || f >= n_seg) return true;
    }
    return false;
  };
  if (bad(P.px_stamp) || bad(P.br_stamp)) return p->err = "visual residual stamp outside the valid range of the spline", HS_ERR_INVALID;
  if (bad(P.pr_stamp)) return p->err = "prior residual stamp outside the valid range of the spline", HS_ERR_INVALID;
  if (bad(P.in_stamp)) return p->err = "inertial residual stamp outside the valid range of the spline", HS_ERR_INVALID;
  for (double t : P.in_stamp) {
    const int f = segment_of(t, P.bias_t0, P.bias_dt, P.kb, &u);
    if (f < 0 || f + P.kb > int(P.bias_g.size() / 4)) return p->err = "inertial residual stamp outside the valid range of the bias splines", HS_ERR_INVALID;
  }
  return HS_OK;
}
#define VALIDATE()                        \
  do {                                    \
    if (int rc_ = validate(p)) return rc_; \
  } while (0)

template <class Keep>
static void keep_rows(Keep&& keep, std::vector<double>* stamp, std::vector<double>* meas, int width, std::vector<int32_t>* a = nullptr, std::vector<int32_t>* b = nullptr) {
  size_t w = 0;
  for (size_t i = 0; i < stamp->size(); ++i) {
    if (!keep(i)) continue;
    (*stamp)[w] = (*stamp)[i];
    for (int c = 0; c < width; ++c) (*meas)[width * w + c] = (*meas)[width * i + c];
    if (a) (*a)[w] = (*a)[i];
    if (b) (*b)[w] = (*b)[i];
    ++w;
  }
  stamp->resize(w), meas->resize(width * w);
  if (a) a->resize(w);
  if (b) b->resize(w);
}

extern "C" {

int hso_create(int, void*, hso_problem** out) {
  *out = new hso_problem();
  if (const char* e = std::getenv("HS_REFERENCE_LITERAL")) (*out)->P.inertial_mode = std::atoi(e) ? HS_INERTIAL_AS_REFERENCE : HS_INERTIAL_EXACT;
  return HS_OK;
}
int hso_destroy(hso_problem* p) {
  delete p;
  return HS_OK;
}
const char* hso_last_error(const hso_problem* p) { return p ? p->err.c_str() : "null handle"; }

int hso_set_spline(hso_problem* p, int order, double t0, double dt, int n_cp, const double* cp, const uint8_t* cp_constant, int rot_c, int trans_c) {
  CHECK_ARG(order >= 2 && order <= kMaxOrder, "order out of range");
  CHECK_ARG(n_cp >= order && dt > 0, "need n_cp >= order and dt > 0");
  const double knot_tol = std::max(1e-6 * dt, 8.0 * 2.220446049250313e-16 * std::max(std::fabs(t0), std::fabs(t0 + n_cp * dt)));
  for (int j = 0; j < n_cp; ++j)  // uniform basis: same rule, tolerance, code and message as hs_set_spline
    if (!(std::fabs(cp[8 * j + 7] - (t0 + j * dt)) <= knot_tol)) {
      p->err = "control-point stamps are not t0 + j dt: the spline basis is uniform, a table with a hole or non-uniform knots is refused";
      return HS_ERR_KNOTS;
    }
  Problem& P = p->P;
  P.k = order, P.t0 = t0, P.dt = dt, P.n_cp = n_cp;
  P.cp.assign(cp, cp + size_t(8) * n_cp);
  P.cp_const.assign(n_cp, 0);
  if (cp_constant) P.cp_const.assign(cp_constant, cp_constant + n_cp);
  P.rot_const = rot_c != 0, P.trans_const = trans_c != 0;
  return HS_OK;
}
int hso_set_cameras(hso_problem* p, int n, const double* T, const double* in, const double* di) {
  Problem& P = p->P;
  P.n_cam = n;
  P.cam_T_bs.assign(T, T + 7 * n), P.cam_intr.assign(in, in + 4 * n), P.cam_dist.assign(di, di + 4 * n);
  return HS_OK;
}
int hso_set_sensors(hso_problem* p, int n, const double* T) {
  p->P.n_sensor = n;
  p->P.sensor_T_bs.assign(T, T + 7 * n);
  return HS_OK;
}
int hso_set_landmarks(hso_problem* p, int n, const double* xyz, const uint8_t* c) {
  Problem& P = p->P;
  P.n_lm = n;
  P.lm.assign(xyz, xyz + 3 * n);
  P.lm_const.assign(n, 0);
  if (c) P.lm_const.assign(c, c + n);
  return HS_OK;
}
int hso_set_imu(hso_problem* p, const double* T, const double* ig, const double* ia, const double* Sg, const double* Xa, int kb, double bt0,
                double bdt, int nb, const double* bg, const double* ba, int bias_constant) {
  CHECK_ARG(kb >= 2 && kb <= kMaxOrder && nb >= kb && bdt > 0, "bad bias spline");
  Problem& P = p->P;
  P.has_imu = true;
  std::memcpy(P.imu_T_bs, T, 7 * 8), std::memcpy(P.imu_i_g, ig, 6 * 8), std::memcpy(P.imu_i_a, ia, 6 * 8);
  std::memcpy(P.imu_S_g, Sg, 9 * 8), std::memcpy(P.imu_X_a, Xa, 9 * 8);
  P.kb = kb, P.bias_t0 = bt0, P.bias_dt = bdt, P.n_bias = nb;
  P.bias_g.assign(bg, bg + 4 * nb), P.bias_a.assign(ba, ba + 4 * nb);
  P.bias_const = bias_constant != 0;
  return HS_OK;
}
int hso_set_weights(hso_problem* p, int type, const double* w) {
  CHECK_ARG(type >= 0 && type <= 3, "unknown factor type");
  const int nr = type == HS_PIXEL ? 2 : (type == HS_BEARING ? 1 : 6);
  if (w)
    p->P.weights[type].assign(w, w + nr * nr);
  else
    p->P.weights[type].clear();
  return HS_OK;
}
int hso_set_stage_timing(hso_problem*, int) { return HS_OK; }  // (the oracle's stage times are host clocks: always on)
int hso_set_inertial_jacobian(hso_problem* p, int mode) {
  CHECK_ARG(mode == HS_INERTIAL_AS_REFERENCE || mode == HS_INERTIAL_EXACT, "unknown inertial Jacobian mode");
  p->P.inertial_mode = mode;
  return HS_OK;
}
int hso_set_gravity(hso_problem* p, const double* g, int constant) {
  std::memcpy(p->P.gravity, g, 24);
  p->P.gravity_const = constant != 0;
  return HS_OK;
}
int hso_set_pixel_residuals(hso_problem* p, int n, const double* st, const double* px, const int32_t* lm, const int32_t* cam) {
  Problem& P = p->P;
  P.px_stamp.assign(st, st + n), P.px_meas.assign(px, px + 2 * n), P.px_lm.assign(lm, lm + n), P.px_cam.assign(cam, cam + n);
  return HS_OK;
}
int hso_set_bearing_residuals(hso_problem* p, int n, const double* st, const double* b, const int32_t* lm, const int32_t* cam) {
  Problem& P = p->P;
  P.br_stamp.assign(st, st + n), P.br_meas.assign(b, b + 3 * n), P.br_lm.assign(lm, lm + n), P.br_cam.assign(cam, cam + n);
  return HS_OK;
}
int hso_set_prior_residuals(hso_problem* p, int n, const double* st, const double* poses, const int32_t* sensor) {
  Problem& P = p->P;
  P.pr_stamp.assign(st, st + n), P.pr_meas.assign(poses, poses + 7 * n), P.pr_sensor.assign(sensor, sensor + n);
  return HS_OK;
}
int hso_set_inertial_residuals(hso_problem* p, int n, const double* st, const double* m) {
  Problem& P = p->P;
  P.in_stamp.assign(st, st + n), P.in_meas.assign(m, m + 6 * n);
  return HS_OK;
}

// ---- delta interface (include/hyperslam_hip.h): the oracle's variables live in the same vectors its solve updates, so a delta call has
//      nothing to pull; hso_stage has nothing to do (no device tables) ----
int hso_append_landmarks(hso_problem* p, int n, const double* xyz, const uint8_t* constant, int32_t* first_index) {
  Problem& P = p->P;
  CHECK_ARG(n >= 0 && (n == 0 || xyz), "bad landmark rows");
  if (first_index) *first_index = P.n_lm;
  P.lm.insert(P.lm.end(), xyz, xyz + 3 * size_t(n));
  for (int i = 0; i < n; ++i) P.lm_const.push_back(constant ? constant[i] : 0);
  P.n_lm += n;
  return HS_OK;
}
int hso_append_pixel_residuals(hso_problem* p, int n, const double* st, const double* px, const int32_t* lm, const int32_t* cam) {
  Problem& P = p->P;
  P.px_stamp.insert(P.px_stamp.end(), st, st + n), P.px_meas.insert(P.px_meas.end(), px, px + 2 * size_t(n));
  P.px_lm.insert(P.px_lm.end(), lm, lm + n), P.px_cam.insert(P.px_cam.end(), cam, cam + n);
  return HS_OK;
}
int hso_append_bearing_residuals(hso_problem* p, int n, const double* st, const double* b, const int32_t* lm, const int32_t* cam) {
  Problem& P = p->P;
  P.br_stamp.insert(P.br_stamp.end(), st, st + n), P.br_meas.insert(P.br_meas.end(), b, b + 3 * size_t(n));
  P.br_lm.insert(P.br_lm.end(), lm, lm + n), P.br_cam.insert(P.br_cam.end(), cam, cam + n);
  return HS_OK;
}
int hso_append_prior_residuals(hso_problem* p, int n, const double* st, const double* poses, const int32_t* sensor) {
  Problem& P = p->P;
  P.pr_stamp.insert(P.pr_stamp.end(), st, st + n), P.pr_meas.insert(P.pr_meas.end(), poses, poses + 7 * size_t(n));
  P.pr_sensor.insert(P.pr_sensor.end(), sensor, sensor + n);
  return HS_OK;
}
int hso_append_inertial_residuals(hso_problem* p, int n, const double* st, const double* m) {
  Problem& P = p->P;
  P.in_stamp.insert(P.in_stamp.end(), st, st + n), P.in_meas.insert(P.in_meas.end(), m, m + 6 * size_t(n));
  return HS_OK;
}
int hso_retire_landmarks(hso_problem* p, int n, const int32_t* ids, int32_t* remap) {
  Problem& P = p->P;
  const int n_old = P.n_lm;
  std::vector<int32_t> local;
  if (!remap) local.resize(n_old), remap = local.data();
  for (int i = 0; i < n; ++i) CHECK_ARG(ids[i] >= 0 && ids[i] < n_old, "hs_retire_landmarks: landmark outside the landmark table");
  for (int t = 0; t < n_old; ++t) remap[t] = 0;
  for (int i = 0; i < n; ++i) remap[ids[i]] = -1;
  int w = 0;
  for (int t = 0; t < n_old; ++t) {
    if (remap[t] < 0) continue;
    for (int c = 0; c < 3; ++c) P.lm[3 * size_t(w) + c] = P.lm[3 * size_t(t) + c];
    P.lm_const[w] = P.lm_const[t];
    remap[t] = w++;
  }
  P.n_lm = w, P.lm.resize(3 * size_t(w)), P.lm_const.resize(w);
  keep_rows([&](size_t i) { return remap[P.px_lm[i]] >= 0; }, &P.px_stamp, &P.px_meas, 2, &P.px_lm, &P.px_cam);
  keep_rows([&](size_t i) { return remap[P.br_lm[i]] >= 0; }, &P.br_stamp, &P.br_meas, 3, &P.br_lm, &P.br_cam);
  for (int32_t& l : P.px_lm) l = remap[l];
  for (int32_t& l : P.br_lm) l = remap[l];
  return HS_OK;
}
int hso_retire_residuals_before(hso_problem* p, int type, double stamp) {
  Problem& P = p->P;
  CHECK_ARG(type >= 0 && type <= 3, "unknown factor type");
  switch (type) {
    case HS_PIXEL: keep_rows([&](size_t i) { return !(P.px_stamp[i] < stamp); }, &P.px_stamp, &P.px_meas, 2, &P.px_lm, &P.px_cam); break;
    case HS_BEARING: keep_rows([&](size_t i) { return !(P.br_stamp[i] < stamp); }, &P.br_stamp, &P.br_meas, 3, &P.br_lm, &P.br_cam); break;
    case HS_PRIOR: keep_rows([&](size_t i) { return !(P.pr_stamp[i] < stamp); }, &P.pr_stamp, &P.pr_meas, 7, &P.pr_sensor); break;
    default: keep_rows([&](size_t i) { return !(P.in_stamp[i] < stamp); }, &P.in_stamp, &P.in_meas, 6); break;
  }
  return HS_OK;
}
int hso_stage(hso_problem*) { return HS_OK; }

int hso_num_residuals(hso_problem* p, int type) { return p->P.n_res(FactorType(type)); }
int hso_dim_pose(hso_problem* p) { return p->P.dim_pose(); }

int hso_residual_layout(hso_problem* p, int type, int idx, int32_t* num_blocks, int32_t* indices, int32_t* sizes, int32_t* offsets,
                        int32_t* block_ids, int32_t* num_parameters, int32_t* num_residuals) {
  const Problem& P = p->P;
  CHECK_ARG(type >= 0 && type < 4 && idx >= 0 && idx < P.n_res(FactorType(type)), "residual index out of range");
  const Layout L = make_layout(FactorType(type), P.k, P.kb);
  *num_blocks = int(L.sizes.size());
  indices[0] = L.static_state_idx, indices[1] = L.static_sensor_idx, indices[2] = L.dynamic_sensor_idx, indices[3] = L.static_observation_idx;
  for (size_t i = 0; i < L.sizes.size(); ++i) sizes[i] = L.sizes[i], offsets[i] = L.offsets[i];
  *num_parameters = L.num_parameters, *num_residuals = L.num_residuals;
  double st = 0, u;
  switch (type) {
    case kPixel: st = P.px_stamp[idx]; break;
    case kBearing: st = P.br_stamp[idx]; break;
    case kPrior: st = P.pr_stamp[idx]; break;
    case kInertial: st = P.in_stamp[idx]; break;
  }
  const int first = segment_of(st, P.t0, P.dt, P.k, &u);
  int b = 0;
  for (int j = 0; j < P.k; ++j) block_ids[b++] = first + j;
  if (type == kPixel || type == kBearing) {
    const int cam = type == kPixel ? P.px_cam[idx] : P.br_cam[idx];
    block_ids[b++] = cam, block_ids[b++] = cam, block_ids[b++] = cam;
    block_ids[b++] = type == kPixel ? P.px_lm[idx] : P.br_lm[idx];
  } else if (type == kPrior) {
    block_ids[b++] = P.pr_sensor[idx];
  } else {
    for (int j = 0; j < 5; ++j) block_ids[b++] = 0;
    const int fb = segment_of(st, P.bias_t0, P.bias_dt, P.kb, &u);
    for (int j = 0; j < P.kb; ++j) block_ids[b++] = fb + j;
    for (int j = 0; j < P.kb; ++j) block_ids[b++] = fb + j;
    block_ids[b++] = 0;
  }
  return HS_OK;
}

int hso_linearize(hso_problem* p, int type, int robustify, const hs_linearization* out) {
  VALIDATE();
  const Problem& P = p->P;
  const FactorType t = FactorType(type);
  const int n = P.n_res(t), k = P.k, kb = P.kb;
  Evaluator ev(P);
  Linearized lin;
  const bool want_sensor = out->J_extrinsics || out->J_intrinsics || out->J_distortion || out->J_gyro_intrinsics || out->J_acc_intrinsics ||
                           out->J_gyro_sensitivity || out->J_acc_offsets;
  for (int i = 0; i < n; ++i) {
    ev.evaluate(t, i, robustify != 0, &lin, nullptr, nullptr, true, want_sensor);
    const int nr = lin.n_res;
    if (out->J_extrinsics) std::memcpy(out->J_extrinsics + size_t(i) * nr * 6, lin.J_ext, sizeof(double) * nr * 6);
    if (t == kPixel) {
      if (out->J_intrinsics) std::memcpy(out->J_intrinsics + size_t(i) * 8, lin.J_intr, sizeof(double) * 8);
      if (out->J_distortion) std::memcpy(out->J_distortion + size_t(i) * 8, lin.J_dist, sizeof(double) * 8);
    }
    if (t == kInertial) {
      if (out->J_gyro_intrinsics) std::memcpy(out->J_gyro_intrinsics + size_t(i) * 36, lin.J_ig, sizeof(double) * 36);
      if (out->J_acc_intrinsics) std::memcpy(out->J_acc_intrinsics + size_t(i) * 36, lin.J_ia, sizeof(double) * 36);
      if (out->J_gyro_sensitivity) std::memcpy(out->J_gyro_sensitivity + size_t(i) * 54, lin.J_Sg, sizeof(double) * 54);
      if (out->J_acc_offsets) std::memcpy(out->J_acc_offsets + size_t(i) * 54, lin.J_Xa, sizeof(double) * 54);
    }
    if (out->r)
      for (int r = 0; r < nr; ++r) out->r[size_t(i) * nr + r] = lin.r[r];
    if (out->J_state) std::memcpy(out->J_state + size_t(i) * nr * 6 * k, lin.J_state.data(), sizeof(double) * nr * 6 * k);
    if (out->J_landmark && lin.lm >= 0) std::memcpy(out->J_landmark + size_t(i) * nr * 3, lin.J_lm, sizeof(double) * nr * 3);
    if (t == kInertial) {
      if (out->J_bias_g) std::memcpy(out->J_bias_g + size_t(i) * 6 * 3 * kb, lin.J_bias_g.data(), sizeof(double) * 6 * 3 * kb);
      if (out->J_bias_a) std::memcpy(out->J_bias_a + size_t(i) * 6 * 3 * kb, lin.J_bias_a.data(), sizeof(double) * 6 * 3 * kb);
      if (out->J_gravity) std::memcpy(out->J_gravity + size_t(i) * 12, lin.J_grav, sizeof(double) * 12);
      if (out->first_bias) out->first_bias[i] = lin.first_bias;
    }
    if (out->first_cp) out->first_cp[i] = lin.first_cp;
    if (out->cost) out->cost[i] = lin.cost;
  }
  return HS_OK;
}

int hso_cost_function_evaluate(hso_problem* p, int type, int idx, const double* const* parameters, double* residuals, double** jacobians) {
  const Problem& P = p->P;
  const FactorType t = FactorType(type);
  CHECK_ARG(idx >= 0 && idx < P.n_res(t), "residual index out of range");
  const Basis basis = make_basis(P.k), bias_basis = make_basis(P.kb);
  const Layout L = make_layout(t, P.k, P.kb);
  double stamp = 0;
  const double* meas = nullptr;
  switch (t) {
    case kPixel: stamp = P.px_stamp[idx], meas = &P.px_meas[2 * idx]; break;
    case kBearing: stamp = P.br_stamp[idx], meas = &P.br_meas[3 * idx]; break;
    case kPrior: stamp = P.pr_stamp[idx], meas = &P.pr_meas[7 * idx]; break;
    case kInertial: stamp = P.in_stamp[idx], meas = &P.in_meas[6 * idx]; break;
  }
  const CostContext ctx = {t, &basis, &bias_basis, stamp, meas, P.inertial_mode == 0, P.weights[t].empty() ? nullptr : P.weights[t].data()};
  cost_evaluate(ctx, L, parameters, residuals, jacobians);
  return HS_OK;
}

int hso_cost(hso_problem* p, double* cost) {
  VALIDATE();
  *cost = Solver(p->P).total_cost();
  return HS_OK;
}

int hso_reduced_system(hso_problem* p, double radius, double* S, double* g) {
  VALIDATE();
  LM lm(p->P);
  attach(p, &lm);
  lm.radius = radius;
  NormalEquations ne;
  lm.solver.build(&ne);
  lm.globalize(&ne);
  lm.compute_scaling(ne);
  std::vector<double> sp, sl;
  ReducedSystem rs;
  if (!lm.solve_step(ne, &sp, &sl, &rs)) {
    p->err = "reduced system not positive definite";
    // still export the system
  }
  std::memcpy(S, rs.S.data(), sizeof(double) * rs.S.size());
  std::memcpy(g, rs.g.data(), sizeof(double) * rs.g.size());
  return HS_OK;
}

int hso_set_allreduce(hso_problem* p, hs_allreduce_fn fn, void* user) {
  p->allreduce = fn, p->allreduce_user = user;
  return HS_OK;
}
int hso_set_shard(hso_problem* p, int rank, int world, int) {
  p->rank = rank, p->world = world;
  return HS_OK;
}
int hso_band_blocks(hso_problem*) { return 0; }

int hso_solve(hso_problem* p, int max_iterations, hs_summary* summary, hs_iteration* iterations) {
  VALIDATE();
  const auto t0 = std::chrono::steady_clock::now();
  LM lm(p->P);
  attach(p, &lm);
  const Summary s = lm.run(max_iterations);
  const auto t1 = std::chrono::steady_clock::now();
  std::memset(summary, 0, sizeof(*summary));
  summary->initial_cost = s.initial_cost, summary->final_cost = s.final_cost;
  summary->num_iterations = s.num_iterations, summary->num_successful_steps = s.num_successful_steps;
  summary->termination = s.termination;
  summary->num_residual_blocks = p->P.n_res(kPixel) + p->P.n_res(kBearing) + p->P.n_res(kPrior) + p->P.n_res(kInertial);
  summary->total_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  summary->linearize_ms = s.linearize_ms, summary->schur_ms = s.schur_ms, summary->solve_ms = s.solve_ms, summary->update_ms = s.update_ms;
  if (iterations) {
    std::memset(iterations, 0, sizeof(hs_iteration) * (max_iterations + 1));
    for (size_t i = 0; i < s.iterations.size() && int(i) <= max_iterations; ++i) {
      const IterationRecord& r = s.iterations[i];
      iterations[i] = {r.iteration, r.step_is_valid, r.step_is_successful, 0, r.cost, r.cost_change, r.gradient_max_norm, r.step_norm, r.relative_decrease, r.radius};
    }
  }
  return HS_OK;
}

int hso_get_control_points(hso_problem* p, double* cp) {
  std::memcpy(cp, p->P.cp.data(), sizeof(double) * p->P.cp.size());
  return HS_OK;
}
int hso_get_landmarks(hso_problem* p, double* xyz) {
  std::memcpy(xyz, p->P.lm.data(), sizeof(double) * p->P.lm.size());
  return HS_OK;
}
int hso_get_bias(hso_problem* p, double* bg, double* ba) {
  std::memcpy(bg, p->P.bias_g.data(), sizeof(double) * p->P.bias_g.size());
  std::memcpy(ba, p->P.bias_a.data(), sizeof(double) * p->P.bias_a.size());
  return HS_OK;
}
int hso_get_gravity(hso_problem* p, double* g) {
  std::memcpy(g, p->P.gravity, 24);
  return HS_OK;
}

/// AbstractOptimizer::process(VisualTracks) front half (abstract.cpp:197-223,250-255) with the EXTERNAL pieces restated:
/// Camera::convertPixelsToBearings = radtan undistortion by 20 fixed-point iterations + normalisation, Camera::Triangulate =
/// midpoint of the two rays in the frame of camera 0, lifted to the world through the spline value at `stamp`.
static V3 pixel_to_bearing(const double* intr, const double* dist, double u, double v) {
  const double xd = (u - intr[0]) / intr[2], yd = (v - intr[1]) / intr[3];
  const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
  double x = xd, y = yd;
  for (int it = 0; it < 20; ++it) {
    const double r2 = x * x + y * y, rad = 1 + k1 * r2 + k2 * r2 * r2;
    const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x), dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
    x = (xd - dx) / rad, y = (yd - dy) / rad;
  }
  const double n = std::sqrt(x * x + y * y + 1);
  return v3(x / n, y / n, 1 / n);
}

int hso_process_tracks(hso_problem* p, double stamp, int n, const double* pixels0, const double* pixels1, double* bearings0, double* bearings1,
                       double* positions_w) {
  const Problem& P = p->P;
  CHECK_ARG(P.cam_T_bs.size() >= 14, "hs_process_tracks needs a stereo pair (two cameras)");
  Quat q_wb{0, 0, 0, 1};
  V3 p_wb = v3(0, 0, 0);
  if (positions_w) {
    double u;
    const int first = segment_of(stamp, P.t0, P.dt, P.k, &u);
    CHECK_ARG(first >= 0 && first + P.k <= P.n_cp, "stamp outside the valid range of the spline");
    const Basis basis = make_basis(P.k);
    const double* cps[kMaxOrder];
    for (int j = 0; j < P.k; ++j) cps[j] = &P.cp[8 * (first + j)];
    SplineValue sv;
    spline_evaluate(basis, cps, u, 1.0 / P.dt, 0, false, &sv);
    q_wb = sv.q, p_wb = v3(sv.p[0], sv.p[1], sv.p[2]);
  }
  const double* T0 = &P.cam_T_bs[0], *T1 = &P.cam_T_bs[7];
  const M3 R_wb = qmat(q_wb), R_b0 = qmat(Quat{T0[0], T0[1], T0[2], T0[3]}), R_b1 = qmat(Quat{T1[0], T1[1], T1[2], T1[3]});
  const V3 t_b0 = v3(T0[4], T0[5], T0[6]), t_b1 = v3(T1[4], T1[5], T1[6]);
  const M3 R_01 = T(R_b0) * R_b1;
  const V3 o = T(R_b0) * (t_b1 - t_b0);
  for (int i = 0; i < n; ++i) {
    const V3 b0 = pixel_to_bearing(&P.cam_intr[0], &P.cam_dist[0], pixels0[2 * i], pixels0[2 * i + 1]);
    const V3 b1 = pixel_to_bearing(&P.cam_intr[4], &P.cam_dist[4], pixels1[2 * i], pixels1[2 * i + 1]);
    if (bearings0)
      for (int c = 0; c < 3; ++c) bearings0[3 * i + c] = b0[c];
    if (bearings1)
      for (int c = 0; c < 3; ++c) bearings1[3 * i + c] = b1[c];
    if (!positions_w) continue;
    const V3 d1 = R_01 * b1;
    const double a = dot(b0, b0), b = dot(b0, d1), c = dot(d1, d1), e = dot(b0, o), f = dot(d1, o);
    const double den = a * c - b * b;
    const double s0 = den > 1e-12 ? (c * e - b * f) / den : 1.0, s1 = den > 1e-12 ? (b * e - a * f) / den : 1.0;
    const V3 p0 = 0.5 * (s0 * b0 + o + s1 * d1);
    const V3 pw = R_wb * (R_b0 * p0 + t_b0) + p_wb;
    for (int cc = 0; cc < 3; ++cc) positions_w[3 * i + cc] = pw[cc];
  }
  return HS_OK;
}

int hso_sample_trajectory(hso_problem* p, int n, const double* stamps, double* pose, double* velocity, double* acceleration) {
  const Problem& P = p->P;
  const Basis basis = make_basis(P.k);
  for (int i = 0; i < n; ++i) {
    double u;
    const int first = segment_of(stamps[i], P.t0, P.dt, P.k, &u);
    CHECK_ARG(first >= 0 && first + P.k <= P.n_cp, "stamp outside the spline's valid range");
    const double* cps[kMaxOrder];
    for (int j = 0; j < P.k; ++j) cps[j] = &P.cp[8 * (first + j)];
    SplineValue s;
    spline_evaluate(basis, cps, u, 1.0 / P.dt, 2, false, &s);
    double* o = pose + 7 * i;
    o[0] = s.q.x, o[1] = s.q.y, o[2] = s.q.z, o[3] = s.q.w, o[4] = s.p[0], o[5] = s.p[1], o[6] = s.p[2];
    if (velocity)
      for (int c = 0; c < 3; ++c) velocity[6 * i + c] = s.w[c], velocity[6 * i + 3 + c] = s.v[c];
    if (acceleration)
      for (int c = 0; c < 3; ++c) acceleration[6 * i + c] = s.al[c], acceleration[6 * i + 3 + c] = s.a[c];
  }
  return HS_OK;
}

// Manifold::Plus / PlusJacobian of the variable classes (wrapper.hpp:32-38): hs_factors.hpp manifold_plus / manifold_plus_jacobian.
int hso_manifold_tangent_size(int kind, int ambient) {
  const bool free_size = kind == HS_MANIFOLD_CONSTANT || kind == HS_MANIFOLD_EUCLIDEAN;
  const int need[6] = {0, 0, 8, 7, 3, 4};
  if (kind < 0 || kind > 5 || (free_size ? (ambient < 1 || ambient > 9) : ambient != need[kind])) return -1;
  return manifold_local_size(ManifoldKind(kind), ambient);
}
int hso_manifold_plus(hso_problem* p, int kind, int ambient, int n, const double* x, const double* delta, double* x_plus_delta) {
  const int tangent = hso_manifold_tangent_size(kind, ambient);
  CHECK_ARG(tangent >= 0, "unknown manifold kind / ambient size");
  for (int i = 0; i < n; ++i)
    manifold_plus(ManifoldKind(kind), ambient, x + size_t(i) * ambient, delta ? delta + size_t(i) * tangent : nullptr, x_plus_delta + size_t(i) * ambient);
  return HS_OK;
}
int hso_manifold_plus_jacobian(hso_problem* p, int kind, int ambient, int n, const double* x, double* jacobian) {
  const int tangent = hso_manifold_tangent_size(kind, ambient);
  CHECK_ARG(tangent >= 0, "unknown manifold kind / ambient size");
  for (int i = 0; i < n; ++i) manifold_plus_jacobian(ManifoldKind(kind), ambient, x + size_t(i) * ambient, jacobian + size_t(i) * ambient * tangent);
  return HS_OK;
}
int hso_manifold_minus(hso_problem* p, int kind, int ambient, int n, const double* y, const double* x, double* y_minus_x) {
  const int tangent = hso_manifold_tangent_size(kind, ambient);
  CHECK_ARG(tangent >= 0, "unknown manifold kind / ambient size");
  for (int i = 0; i < n && tangent > 0; ++i)
    manifold_minus(ManifoldKind(kind), ambient, y + size_t(i) * ambient, x + size_t(i) * ambient, y_minus_x + size_t(i) * tangent);
  return HS_OK;
}
int hso_manifold_minus_jacobian(hso_problem* p, int kind, int ambient, int n, const double* x, double* jacobian) {
  const int tangent = hso_manifold_tangent_size(kind, ambient);
  CHECK_ARG(tangent >= 0, "unknown manifold kind / ambient size");
  for (int i = 0; i < n && tangent > 0; ++i) manifold_minus_jacobian(ManifoldKind(kind), ambient, x + size_t(i) * ambient, jacobian + size_t(i) * ambient * tangent);
  return HS_OK;
}

}  // extern "C"
