// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_math.hpp header). PARITY UNPINNED.
// DIAGNOSTIC build of the CPU restatement with every `double` of hs_math.hpp / hs_factors.hpp / hs_problem.hpp compiled as the x87
// 80-bit `long double` (64-bit mantissa, eps 1.08e-19): the same algorithm, the same operation order, 2048 times less rounding.
// It answers "which side carries the error" when the HIP library and the double oracle differ by more than the lock-step bar on an
// ill-conditioned window: tests/harness/replay_lockstep takes this library as its shadow
//     replay_lockstep oracle/liboracle_ld.so 3.6 0 4 hsl_     double oracle (master) vs long-double oracle (shadow), CPU only
// and tools/lockstep_three_way.py compares oracle(double), oracle(long double) and HIP from the same tables on the GPU box.
// Exported: the subset of include/hyperslam_hip.h the lock-step harness binds, under a prefix of its own, hsl_* — NOT the product's hs_*: a
// test library that answers to the product's names could stand in for it through LD_PRELOAD and turn a parity test green against itself.
// Inputs and outputs cross the boundary as double; nothing inside is rounded to double.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../include/hyperslam_hip.h"  // the ABI keeps its doubles: included before the type switch below

typedef long double hs_real;
#define double hs_real  // the restatement's arithmetic type, for the three headers only
#include "hs_problem.hpp"
#undef double

using namespace hso;

struct hs_problem {
  Problem P;
  std::string err;
};

template <class D, class S>
static void put(D* dst, const S* src, size_t n) {
  for (size_t i = 0; i < n; ++i) dst[i] = D(src[i]);
}

extern "C" {

int hsl_create(int, void*, hs_problem** out) {
  *out = new hs_problem();
  if (const char* e = std::getenv("HS_REFERENCE_LITERAL")) (*out)->P.inertial_mode = std::atoi(e) ? HS_INERTIAL_AS_REFERENCE : HS_INERTIAL_EXACT;
  return HS_OK;
}
int hsl_destroy(hs_problem* p) {
  delete p;
  return HS_OK;
}
const char* hsl_last_error(const hs_problem* p) { return p ? p->err.c_str() : "null handle"; }

int hsl_set_spline(hs_problem* p, int order, double t0, double dt, int n_cp, const double* cp, const uint8_t* cp_constant, int rot_c, int trans_c) {
  if (order < 2 || order > kMaxOrder || n_cp < order || !(dt > 0)) return p->err = "bad spline", HS_ERR_INVALID;
  Problem& P = p->P;
  P.k = order, P.t0 = t0, P.dt = dt, P.n_cp = n_cp;
  P.cp.assign(cp, cp + size_t(8) * n_cp);
  P.cp_const.assign(n_cp, 0);
  if (cp_constant) P.cp_const.assign(cp_constant, cp_constant + n_cp);
  P.rot_const = rot_c != 0, P.trans_const = trans_c != 0;
  return HS_OK;
}
int hsl_set_cameras(hs_problem* p, int n, const double* T, const double* in, const double* di) {
  Problem& P = p->P;
  P.n_cam = n;
  P.cam_T_bs.assign(T, T + 7 * n), P.cam_intr.assign(in, in + 4 * n), P.cam_dist.assign(di, di + 4 * n);
  return HS_OK;
}
int hsl_set_sensors(hs_problem* p, int n, const double* T) {
  p->P.n_sensor = n;
  p->P.sensor_T_bs.assign(T, T + 7 * n);
  return HS_OK;
}
int hsl_set_landmarks(hs_problem* p, int n, const double* xyz, const uint8_t* c) {
  Problem& P = p->P;
  P.n_lm = n;
  P.lm.assign(xyz, xyz + 3 * n);
  P.lm_const.assign(n, 0);
  if (c) P.lm_const.assign(c, c + n);
  return HS_OK;
}
int hsl_set_imu(hs_problem* p, const double* T, const double* ig, const double* ia, const double* Sg, const double* Xa, int kb, double bt0, double bdt,
               int nb, const double* bg, const double* ba, int bias_constant) {
  if (kb < 2 || kb > kMaxOrder || nb < kb || !(bdt > 0)) return p->err = "bad bias spline", HS_ERR_INVALID;
  Problem& P = p->P;
  P.has_imu = true;
  put(P.imu_T_bs, T, 7), put(P.imu_i_g, ig, 6), put(P.imu_i_a, ia, 6), put(P.imu_S_g, Sg, 9), put(P.imu_X_a, Xa, 9);
  P.kb = kb, P.bias_t0 = bt0, P.bias_dt = bdt, P.n_bias = nb;
  P.bias_g.assign(bg, bg + 4 * nb), P.bias_a.assign(ba, ba + 4 * nb);
  P.bias_const = bias_constant != 0;
  return HS_OK;
}
int hsl_set_gravity(hs_problem* p, const double* g, int constant) {
  put(p->P.gravity, g, 3);
  p->P.gravity_const = constant != 0;
  return HS_OK;
}
int hsl_set_pixel_residuals(hs_problem* p, int n, const double* st, const double* px, const int32_t* lm, const int32_t* cam) {
  Problem& P = p->P;
  P.px_stamp.assign(st, st + n), P.px_meas.assign(px, px + 2 * n), P.px_lm.assign(lm, lm + n), P.px_cam.assign(cam, cam + n);
  return HS_OK;
}
int hsl_set_bearing_residuals(hs_problem* p, int n, const double* st, const double* b, const int32_t* lm, const int32_t* cam) {
  Problem& P = p->P;
  P.br_stamp.assign(st, st + n), P.br_meas.assign(b, b + 3 * n), P.br_lm.assign(lm, lm + n), P.br_cam.assign(cam, cam + n);
  return HS_OK;
}
int hsl_set_prior_residuals(hs_problem* p, int n, const double* st, const double* poses, const int32_t* sensor) {
  Problem& P = p->P;
  P.pr_stamp.assign(st, st + n), P.pr_meas.assign(poses, poses + 7 * n), P.pr_sensor.assign(sensor, sensor + n);
  return HS_OK;
}
int hsl_set_inertial_residuals(hs_problem* p, int n, const double* st, const double* m) {
  Problem& P = p->P;
  P.in_stamp.assign(st, st + n), P.in_meas.assign(m, m + 6 * n);
  return HS_OK;
}
int hsl_dim_pose(hs_problem* p) { return p->P.dim_pose(); }
int hsl_band_blocks(hs_problem*) { return 0; }

int hsl_cost(hs_problem* p, double* cost) {
  *cost = double(Solver(p->P).total_cost());
  return HS_OK;
}

int hsl_reduced_system(hs_problem* p, double radius, double* S, double* g) {
  LM lm(p->P);
  lm.radius = radius;
  NormalEquations ne;
  lm.solver.build(&ne);
  lm.globalize(&ne);
  lm.compute_scaling(ne);
  std::vector<hs_real> sp, sl;
  ReducedSystem rs;
  if (!lm.solve_step(ne, &sp, &sl, &rs)) p->err = "reduced system not positive definite";
  put(S, rs.S.data(), rs.S.size()), put(g, rs.g.data(), rs.g.size());
  return HS_OK;
}

int hsl_solve(hs_problem* p, int max_iterations, hs_summary* summary, hs_iteration* iterations) {
  LM lm(p->P);
  const Summary s = lm.run(max_iterations);
  std::memset(summary, 0, sizeof(*summary));
  summary->initial_cost = double(s.initial_cost), summary->final_cost = double(s.final_cost);
  summary->num_iterations = s.num_iterations, summary->num_successful_steps = s.num_successful_steps;
  summary->termination = s.termination;
  summary->num_residual_blocks = p->P.n_res(kPixel) + p->P.n_res(kBearing) + p->P.n_res(kPrior) + p->P.n_res(kInertial);
  if (iterations) {
    std::memset(iterations, 0, sizeof(hs_iteration) * (max_iterations + 1));
    for (size_t i = 0; i < s.iterations.size() && int(i) <= max_iterations; ++i) {
      const IterationRecord& r = s.iterations[i];
      iterations[i] = {r.iteration,           r.step_is_valid,     r.step_is_successful,        0,
                       double(r.cost),        double(r.cost_change), double(r.gradient_max_norm), double(r.step_norm),
                       double(r.relative_decrease), double(r.radius)};
    }
  }
  return HS_OK;
}

int hsl_get_control_points(hs_problem* p, double* cp) { return put(cp, p->P.cp.data(), p->P.cp.size()), HS_OK; }
int hsl_get_landmarks(hs_problem* p, double* xyz) { return put(xyz, p->P.lm.data(), p->P.lm.size()), HS_OK; }
int hsl_get_bias(hs_problem* p, double* bg, double* ba) {
  return put(bg, p->P.bias_g.data(), p->P.bias_g.size()), put(ba, p->P.bias_a.data(), p->P.bias_a.size()), HS_OK;
}
int hsl_get_gravity(hs_problem* p, double* g) { return put(g, p->P.gravity, 3), HS_OK; }

}  // extern "C"
