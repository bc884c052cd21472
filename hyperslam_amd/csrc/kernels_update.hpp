// kernels_update.hpp — candidate point, trust-region state machine, acceptance (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include "kernels_common.hpp"

namespace hs {

// ---------------------------------------------------------------------------------------------------------------------
// Candidate point of the step, one launch. Workgroups [0, n_lm_part): landmark back-substitution (one wave per landmark):
//   y_l = L^-T (yh_l - Yh_l' (Sp o y_p)),  step_l = -y_l, with y_p = -step_p;   candidate = lm + S_l o step_l,
// with the landmark-side terms of the decision (|x|^2, |x - x+|^2, g.step, step'D^2 step) summed per workgroup in a fixed
// order. Workgroups [n_lm_part, n_lm_part + n_norm_part): candidate control points / bias points / gravity = Plus(x, delta) per
// Ceres manifold (quaternion left-multiplicative, R^3 additive, stamp constant, sphere) and their norms.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_backsub_retract(Tables T) {
  if (T.st->done) return;
  __shared__ double red[kBlock / 64][4];
  // Speculative solves with a deferred commit (DevState::spec == 2): the candidate accepted by the previous iteration is still only in
  // the candidate buffers — it is the current point here, and it is copied to x on the way (every element of x passes through this
  // kernel once per iteration), which replaces one k_commit launch per iteration. hs_solve launches k_commit once behind the last iteration.
  const bool pend = T.st->spec == 2 && T.st->accepted;  // (spec == 4, the fused path: k_update_visual)
  if (int(blockIdx.x) < T.n_lm_part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dl = blockIdx.x * (kBlock / 64) + wave;
    double xl = 0.0, sl = 0.0, gd = 0.0, dd = 0.0;
    if (dl < T.n_lm) {
      const int rows = 6 * T.lm_ncp[dl], r0 = 6 * T.lm_cfirst[dl];
      const double* Y = T.Y + T.lm_yoff[dl];
      // lane 0's operands of the 3x3 solve are requested before the dot products (one memory round trip less on the chain)
      double L[6] = {1, 0, 1, 0, 0, 1}, yh[3] = {0, 0, 0}, x[3] = {0, 0, 0}, sc[3] = {0, 0, 0}, sb[3] = {0, 0, 0}, d2[3] = {0, 0, 0};
      bool active = false;
      if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 6; ++a) L[a] = T.lm_L[6 * dl + a];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          yh[a] = T.lm_yhat[3 * dl + a], x[a] = (pend ? T.lm_cand : T.lm)[3 * dl + a], sc[a] = T.lm_scale[3 * dl + a];
          sb[a] = T.lm_sb[3 * dl + a], d2[a] = T.lm_D2[3 * dl + a];
        }
        active = (T.lm_ptr[dl + 1] > T.lm_ptr[dl]) && !T.lm_const[dl];
      }
      double t0 = 0, t1 = 0, t2 = 0;
      for (int rho0 = lane; rho0 < rows; rho0 += 128) {  // two 64-row passes per round of loads (a track of <= 21 control points: one round)
        double yv[2][3], yp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int rho = rho0 + 64 * u;
          const bool ok = rho < rows;
          yp[u] = ok ? -T.step_p[r0 + rho] * T.scale_p[r0 + rho] : 0.0;
#pragma unroll
          for (int c = 0; c < 3; ++c) yv[u][c] = ok ? Y[3 * rho + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) t0 = fma(yv[u][0], yp[u], t0), t1 = fma(yv[u][1], yp[u], t1), t2 = fma(yv[u][2], yp[u], t2);
      }
      t0 = wave_sum(t0), t1 = wave_sum(t1), t2 = wave_sum(t2);
      if (lane == 0) {
        // L' y = z
        const double z0 = yh[0] - t0, z1 = yh[1] - t1, z2 = yh[2] - t2;
        const double y2 = z2 / L[5], y1 = (z1 - L[4] * y2) / L[2], y0 = (z0 - L[1] * y1 - L[3] * y2) / L[0];
        const double s[3] = {active ? -y0 : 0.0, active ? -y1 : 0.0, active ? -y2 : 0.0};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const double y = x[a] + sc[a] * s[a];
          T.lm_cand[3 * dl + a] = y;
          if (pend) T.lm[3 * dl + a] = x[a];
          if (active) {
            xl = fma(x[a], x[a], xl), sl = fma(x[a] - y, x[a] - y, sl);
            gd = fma(sb[a], s[a], gd);
            dd = fma(d2[a] * s[a], s[a], dd);
          }
        }
      }
    }
    if (lane == 0) red[wave][0] = xl, red[wave][1] = sl, red[wave][2] = gd, red[wave][3] = dd;
    __syncthreads();
    if (threadIdx.x < 4) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) v += red[w][threadIdx.x];
      T.lm_part[4 * blockIdx.x + threadIdx.x] = v;
    }
    return;
  }
  const int blk = blockIdx.x - T.n_lm_part;
  const int j = blk * blockDim.x + threadIdx.x;
  double xs = 0.0, ss = 0.0;
  if (j < T.sp.n_cp) {
    double x[8];  // (a copy: with a pending commit the current point is read from the buffer the candidate is written to)
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = (pend ? T.cp_cand : T.cp)[8 * j + c];
    if (pend) {
#pragma unroll
      for (int c = 0; c < 8; ++c) T.cp[8 * j + c] = x[c];
    }
    double* y = T.cp_cand + 8 * j;
    const double* d = T.delta_p + 6 * j;
    bool any = false;
#pragma unroll
    for (int c = 0; c < 6; ++c) any |= (T.D2p[6 * j + c] != 0.0);
    const Quat q = quat_plus(Quat{x[0], x[1], x[2], x[3]}, V3{d[0], d[1], d[2]});
    y[0] = q.x, y[1] = q.y, y[2] = q.z, y[3] = q.w;
    y[4] = x[4] + d[3], y[5] = x[5] + d[4], y[6] = x[6] + d[5];
    y[7] = x[7];
    if (any) {
#pragma unroll
      for (int c = 0; c < 8; ++c) xs = fma(x[c], x[c], xs), ss = fma(x[c] - y[c], x[c] - y[c], ss);
    }
  }
  // border unknowns (replicated like the control points): bias control points [x y z t] and gravity
  if (T.nb > 0) {
    for (int b = j; b < 2 * T.n_bias; b += T.n_norm_part * blockDim.x) {
      const bool acc = b >= T.n_bias;
      const int bi = acc ? b - T.n_bias : b;
      const double* x = (acc ? T.bias_a : T.bias_g) + 4 * bi;
      double* y = (acc ? T.bias_a_cand : T.bias_g_cand) + 4 * bi;
      const double* d = T.delta_b + 3 * b;
      const bool any = T.D2b[3 * b] != 0.0 || T.D2b[3 * b + 1] != 0.0 || T.D2b[3 * b + 2] != 0.0;
      y[0] = x[0] + d[0], y[1] = x[1] + d[1], y[2] = x[2] + d[2], y[3] = x[3];
      if (any) {
#pragma unroll
        for (int c = 0; c < 4; ++c) xs = fma(x[c], x[c], xs), ss = fma(x[c] - y[c], x[c] - y[c], ss);
      }
    }
    if (j == 0) {
      const double* d = T.delta_b + 6 * T.n_bias;
      double y[3];
      sphere_plus(T.gravity, d, y);
      const bool any = T.D2b[6 * T.n_bias] != 0.0 || T.D2b[6 * T.n_bias + 1] != 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        T.gravity_cand[c] = y[c];
        if (any) xs = fma(T.gravity[c], T.gravity[c], xs), ss = fma(T.gravity[c] - y[c], T.gravity[c] - y[c], ss);
      }
    }
  }
  double* lds = &red[0][0];
  xs = block_sum(xs, lds), ss = block_sum(ss, lds);
  if (threadIdx.x == 0) T.norm_part[2 * blk] = xs, T.norm_part[2 * blk + 1] = ss;
}

// ---------------------------------------------------------------------------------------------------------------------
// Trust-region state machine (one workgroup). Restates TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy
// (Ceres) with the in-tree options of optimizer.cpp:38-54.
//   phase 0: after the first linearisation — record iteration 0.
//   phase 1: after the candidate cost — accept / reject, radius update, termination tests.
// ---------------------------------------------------------------------------------------------------------------------
HSD double ordered_sum(const double* p, int n, double* lds) {
  return block_sum(strided_sum(p, n), lds);
}

__global__ void __launch_bounds__(kBlock) k_cost_reduce(Tables T) {
  // (global) cost of the current linearisation point -> st->cost, gradient max norm -> st->gmax; iteration bookkeeping
  __shared__ double red[kBlock / 64];
  DevState* st = T.st;
  if (st->done) return;
  double gm = strided_max(T.gabs, T.np + T.nb);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_xor(gm, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gm;
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int i = 1; i < int(blockDim.x >> 6); ++i) gm = fmax(gm, red[i]);
  const double c = T.xbuf[T.xo_cost];
  for (int r = 0; r < T.world; ++r) gm = fmax(gm, T.xbuf[T.xo_gmax + r]);
  begin_iteration(T, c, gm, true);
}

/// (global) cost and gradient max norm of the current linearisation point -> state; iteration 0 record; termination tests
/// that precede a step (single lane).
HSD void begin_iteration(const Tables& T, double c, double gm, bool set_scaling_ready) {
  DevState* st = T.st;
  st->cost = c;  // (speculative solves: the sum of the shards' kept costs, pack_exchange_body — equal to what decide_step took over)
  st->gmax = gm;
  if (set_scaling_ready) st->scaling_ready = 1;  // Jacobi scaling is computed at iteration 0 only (else: set by decide_step)
  if (st->iteration == 0) {
    hs_iteration& r = st->records[0];
    r.iteration = 0, r.step_is_valid = 1, r.step_is_successful = 1, r.cost = c, r.cost_change = 0, r.gradient_max_norm = gm;
    r.step_norm = 0, r.relative_decrease = 0, r.radius = st->radius;
    st->iteration = 1;
  } else {
    // TrustRegionMinimizer reports, with iteration i, the gradient at the point AFTER its step (it re-evaluates the Jacobian inside
    // HandleSuccessfulStep). Here that gradient only exists once the next iteration has linearised: patch the previous record. The
    // record of the LAST iteration of a solve keeps the gradient from before its step (no linearisation at the final point: that
    // would be a sixth linearise + build per optimize() whose only consumer is this field).
    st->records[st->iteration - 1].gradient_max_norm = gm;
  }
  // FinalizeIterationAndCheckIfMinimizerCanContinue
  if (st->iteration - 1 >= st->max_iterations) {
    st->done = 1, st->termination = HS_NO_CONVERGENCE;
  } else if (gm <= 1e-10) {
    st->done = 1, st->termination = HS_CONVERGENCE;
  } else if (st->radius <= 1e-32) {
    st->done = 1, st->termination = HS_CONVERGENCE;
  }
}

/// What decide_step reads from the solver state, as one batch of independent loads (its lane requests them with the partial sums of the
/// decision, a memory round trip earlier than it needs them: read one after the other behind the sums they were 1.4 us of the chain).
struct DecideIn {
  double g_pose, g_far, d2_pose, d2_far, cost, gmax, radius, decrease_factor;
  int chol_failed, iteration, rec_pending, invalid_streak;
};
HSD DecideIn decide_inputs(const DevState* st) {
  DecideIn in;
  in.g_pose = st->g_dot_step_pose, in.g_far = st->g_dot_step_far, in.d2_pose = st->d2_step2_pose, in.d2_far = st->d2_step2_far;
  in.cost = st->cost, in.gmax = st->gmax, in.radius = st->radius, in.decrease_factor = st->decrease_factor;
  in.chol_failed = st->chol_failed, in.iteration = st->iteration, in.rec_pending = st->rec_pending, in.invalid_streak = st->invalid_streak;
  return in;
}

HSD void decide_step(const Tables& T, const double* D_known, const DecideIn& in);
HSD void decide_step(const Tables& T, const double* D_known = nullptr);
HSD void commit_body(const Tables& T, int idx, int stride);
HSD void commit_control_points(const Tables& T, int idx, int stride);

/// Second exchange buffer (5 doubles, additive across shards): candidate cost, |x|^2, |x - x+|^2 and the landmark-side
/// terms of the model cost change. The replicated control-point part of the norms is contributed by rank 0 only.
/// publish (fold mode, k_build_visual's decision workgroup): the flag word raised — release, agent scope — as soon as the decision is in the
/// solver state, BEFORE the accepted control points are copied (the waiting workgroups read an accepted point from cp_cand, nobody reads T.cp).
HSD void pack_decision_body(const Tables& T, const int decide_here, double* red /* 5 * kBlock / 64 + 1 doubles of LDS */, unsigned* publish) {
  DevState* st = T.st;
  const int done_at_entry = st->done;  // (requested with the partials below, looked at behind them: one memory round trip, not two)
  // phase stamps of the decision workgroup of a fold-mode build (profiling builds, HS_DEBUG_FLAGS 32; tools/fold_phase_timing.py)
  const bool dprof = prof_enabled(T.debug_flags, 32) && publish && threadIdx.x == 0;
  long long* dlog = reinterpret_cast<long long*>(T.xpart) + 48 * 1024 + 64 * 1023;
  if (dprof) dlog[0] = wall_clock64();
  DecideIn din = {};
  if (threadIdx.x == 0 && decide_here) din = decide_inputs(st);
  // The partial arrays are short (one entry per workgroup of the producing kernels): one combined pass with every load of a round
  // issued before the first use (six separate strided sums cost six memory round trips, 8 us). Fixed order: bit-reproducible.
  double cand = 0.0, xs = 0.0, ss = 0.0, gd = 0.0, dd = 0.0;
  const int n_max = max(T.n_cost_part, max(T.n_lm_part, T.n_norm_part));
  const bool with_replicated = T.rank == 0;  // control points / bias points / gravity are counted once
  for (int i0 = threadIdx.x; i0 < n_max; i0 += 4 * kBlock) {
    double c[4];
    double2 la[4], lb[4], nr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kBlock;
      c[u] = i < T.n_cost_part ? T.cand_part[i] : 0.0;
      const double2* lp = reinterpret_cast<const double2*>(T.lm_part) + 2 * size_t(i);
      la[u] = i < T.n_lm_part ? lp[0] : make_double2(0.0, 0.0);  // (|x|^2, |x - x+|^2)
      lb[u] = i < T.n_lm_part ? lp[1] : make_double2(0.0, 0.0);  // (g.step, step'D^2 step)
      nr[u] = (with_replicated && i < T.n_norm_part) ? reinterpret_cast<const double2*>(T.norm_part)[i] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      cand += c[u];
      xs += la[u].x, ss += la[u].y, gd += lb[u].x, dd += lb[u].y;
      xs += nr[u].x, ss += nr[u].y;
    }
  }
  if (done_at_entry) {  // an earlier iteration ended the solve
    if (publish && threadIdx.x == 0) __hip_atomic_store(publish, (T.fold_epoch << 2) | 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  double v5[5] = {cand, xs, ss, gd, dd};
  if (dprof) dlog[1] = wall_clock64();  // partials loaded
  block_sum_n<5>(v5, red);  // (one pair of barriers for the five sums; the decision takes them from registers, not back from memory)
  if (dprof) dlog[2] = wall_clock64();  // summed
  int* accepted = reinterpret_cast<int*>(red + 5 * (kBlock / 64));  // (handed over in LDS: the other waves may hold the state's cache line from their `done` test)
  if (threadIdx.x == 0) {
    double* D = T.xbuf + T.xo_dec;
#pragma unroll
    for (int e = 0; e < 5; ++e) D[e] = v5[e];
    st->local_cand = v5[0];  // this shard's part (the exchange sums D over the shards)
    if (decide_here) decide_step(T, v5, din);
    if (decide_here >= 2) *accepted = st->accepted;
    if (dprof) dlog[3] = wall_clock64();  // decided
    if (publish)  // (epoch << 2) | (done << 1) | accepted: what the waiting workgroups need at once (fold_wait, kernels_build.hpp)
      __hip_atomic_store(publish, (T.fold_epoch << 2) | (st->done ? 2u : 0u) | (st->accepted ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (dprof) dlog[4] = wall_clock64();  // published
  }
  if (decide_here >= 2) {  // 2, small problems: x <- candidate right here instead of a k_commit launch behind this one;
                           // 3, fused path with deferred landmarks: the control points only (k_build_visual / k_update_visual read T.cp)
    __syncthreads();
    if (*accepted && decide_here == 2) commit_body(T, threadIdx.x, blockDim.x);
    if (*accepted && decide_here == 3) commit_control_points(T, threadIdx.x, blockDim.x);
  }
}

__global__ void __launch_bounds__(kBlock) k_pack_decision(Tables T, int decide_here /* no exchange between packing and deciding */) {
  __shared__ double red[5 * (kBlock / 64) + 1];
  pack_decision_body(T, decide_here, red, nullptr);
}

/// Trust-region decision of one LM iteration (single lane): step quality, acceptance, radius update, termination tests.
HSD void decide_step(const Tables& T, const double* D_known, const DecideIn& in) {
  DevState* st = T.st;
  const double* D = D_known ? D_known : T.xbuf + T.xo_dec;  // (after an exchange: the summed terms from the buffer)
  const double cand = D[0], xs = D[1], ss = D[2];
  // model_cost_change = -g.step/2 + step'D^2 step/2 (exact for the solved system; TrustRegionMinimizer evaluates
  // -(J step).(r + J step/2), identical algebraically)
  const double g_step = (in.g_pose + in.g_far) + D[3], d_step = (in.d2_pose + in.d2_far) + D[4];
  const double mcc = -0.5 * g_step + 0.5 * d_step;
  st->model_cost_change = mcc;
  st->scaling_ready = 1;  // a step was computed: the Jacobi scaling of this solve is fixed from here on
  const int step_valid = (isfinite(mcc) && !in.chol_failed && mcc > 0.0) ? 1 : 0;  // TrustRegionMinimizer: step_is_valid = model_cost_change > 0
  st->step_valid = step_valid;
  const int it = in.iteration;
  hs_iteration& r = st->records[it];
  r.iteration = it, r.cost = in.cost, r.cost_change = 0, r.gradient_max_norm = in.gmax, r.step_norm = 0, r.relative_decrease = 0;
  r.step_is_valid = step_valid, r.step_is_successful = 0;
  st->num_iterations = it;
  st->accepted = 0;
  const int rec_pending = in.rec_pending;  // the candidate was linearised into the other record buffer (k_linearize_visual)
  st->rec_pending = 0;
  st->gmax_bits = 0ull, st->gmax_pose_bits = 0ull;  // the next linearisation re-accumulates them
  double radius = in.radius;
  if (!step_valid) {  // HandleInvalidStep
    if (in.invalid_streak + 1 >= 5) {
      st->done = 1, st->termination = HS_FAILURE;
    } else {
      radius *= 0.5;
      st->radius = radius;
      // The decision CONSUMES the factorisation's verdict (1 = a non-positive pivot): cleared here, behind the kernels that raise it and in
      // front of the next iteration's — not by the next factorisation's own bookkeeping next to its far end's stores (ordering by
      // timing). A solve that ends on it keeps the value for hs_solve's message; 2 (a bounded wait gave up) ends the solve where it is raised.
      if (in.chol_failed == 1) st->chol_failed = 0;
    }
    st->invalid_streak = in.invalid_streak + 1;
    r.radius = radius;
    st->iteration = it + 1;
    return;
  }
  st->invalid_streak = 0;
  st->cand_cost = cand;
  const double step_norm = sqrt(ss);
  r.step_norm = step_norm;
  // ParameterToleranceReached
  if (step_norm <= 1e-8 * (sqrt(xs) + 1e-8)) {
    st->done = 1, st->termination = HS_CONVERGENCE;
    r.radius = radius;
    return;
  }
  // FunctionToleranceReached
  const double cost_change = in.cost - cand;
  r.cost_change = cost_change;
  if (fabs(cost_change) <= 1e-6 * in.cost) {
    st->done = 1, st->termination = HS_CONVERGENCE;
    r.radius = radius;
    return;
  }
  const double relative_decrease = (in.cost - cand) / mcc;
  r.relative_decrease = relative_decrease;
  if (relative_decrease > 1e-3) {  // HandleSuccessfulStep
    r.step_is_successful = 1;
    st->accepted = 1;
    st->num_successful++;
    st->cost = cand;
    st->local_cost = st->local_cand;
    if (rec_pending) st->rec_sel ^= 1;  // its records are the linearisation of the new current point
    r.cost = cand;
    const double q = 2.0 * relative_decrease - 1.0;
    radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - q * q * q));
    st->decrease_factor = 2.0;
  } else {
    r.cost = cand;  // TrustRegionMinimizer reports the candidate's cost for an unsuccessful step (the point itself is unchanged)
    radius = radius / in.decrease_factor;
    st->decrease_factor = in.decrease_factor * 2.0;
  }
  st->radius = radius;
  r.radius = radius;
  st->iteration = it + 1;
}
HSD void decide_step(const Tables& T, const double* D_known) { decide_step(T, D_known, decide_inputs(T.st)); }

/// (commit_cps: the fused path with deferred landmarks — the accepted control points go to T.cp right here)
__global__ void __launch_bounds__(kBlock) k_decide(Tables T, int commit_cps) {
  if (T.st->done) return;
  __shared__ int accepted;
  if (threadIdx.x == 0) {
    decide_step(T);
    accepted = T.st->accepted;
  }
  __syncthreads();
  if (commit_cps && accepted) commit_control_points(T, threadIdx.x, blockDim.x);
}

/// x <- candidate when the step was accepted.
HSD void commit_body(const Tables& T, const int idx, const int stride) {
  // eight loads in flight per lane (a plain copy loop pays one memory round trip per element and lane)
  auto copy = [&](double* __restrict__ dst, const double* __restrict__ src, int n) {
    for (int e0 = idx; e0 < n; e0 += 8 * stride) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * stride;
        v[u] = e < n ? src[e] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * stride;
        if (e < n) dst[e] = v[u];
      }
    }
  };
  copy(T.cp, T.cp_cand, 8 * T.sp.n_cp);
  copy(T.lm, T.lm_cand, 3 * T.n_lm);
  if (T.nb > 0) {
    for (int e = idx; e < 4 * T.n_bias; e += stride) T.bias_g[e] = T.bias_g_cand[e], T.bias_a[e] = T.bias_a_cand[e];
    if (idx < 3) T.gravity[idx] = T.gravity_cand[idx];
  }
}

HSD void commit_control_points(const Tables& T, const int idx, const int stride) {
  for (int e0 = idx; e0 < 8 * T.sp.n_cp; e0 += 4 * stride) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = e0 + u * stride < 8 * T.sp.n_cp ? T.cp_cand[e0 + u * stride] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (e0 + u * stride < 8 * T.sp.n_cp) T.cp[e0 + u * stride] = v[u];
  }
}

constexpr int kCommitInline = 4096;  // elements copied by the decision kernel itself (16 per lane); larger problems launch k_commit (16 000 elements in the
                                     // decision kernel's single workgroup took as long as the launch it saves: measured, configs[1])

__global__ void __launch_bounds__(kBlock) k_commit(Tables T) {
  // note: reads `accepted` even when `done` was just set by a convergence test (those leave accepted = 0)
  if (T.st->accepted) commit_body(T, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

}  // namespace hs
