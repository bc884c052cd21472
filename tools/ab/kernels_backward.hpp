// A/B ALTERNATIVE, PROFILING BUILDS ONLY (tools/build_profiling_lib.sh, -DHS_PROFILE_HOOKS=1; included by hyperslam_amd/csrc/kernels.hpp behind
// that switch): k_band_backward_w / k_premultiply (single-wave register sweep, HS_DEBUG_FLAGS 65536), the measured alternative of k_band_backward_sb. Not part of the product library.
// kernels_backward.hpp — register-resident backward sweep of the banded solve (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include "../../hyperslam_amd/csrc/kernels_factor.hpp"

namespace hs {

// ---------------------------------------------------------------------------------------------------------------------
// Backward sweep U x = y in ONE wave per end, no LDS and no barrier on the chain.
//
// The sweep is a dependency chain over the block rows (x_j needs every x_m, m > j, that row j couples to): the previous kernels
// (k_band_backward / k_band_backward2) spent 0.5 - 0.67 us per block row on an LDS round trip + barrier between four waves.
// Here a pending row keeps its running right-hand side in a REGISTER for its whole lifetime:
//   * row rho lives in slot z = rho mod (6 bw) = lane + 64 s of the wave (s < NS): rows 6 bw apart never overlap in time, because a
//     row is pending only while the sweep is inside its band (bw block rows);
//   * the sweep runs on the block-row-scaled factor V = diag(U_jj^-1) U (unit diagonal blocks) and yt = diag(U_jj^-1) y, written
//     by the factorisation (k_premultiply for the kernels that only produce U): x_j = yt_j - sum_{m > j} V_jm x_m, so the finished
//     running sums of block row j ARE x_j — no triangular solve on the chain;
//   * step j: the six entries of x_j are broadcast with v_readlane (wave-uniform lane / slot), every pending row subtracts
//     V[rho][cols of j] . x_j (6 FMAs per slot), the slots of block row j are re-initialised with yt of block row j - bw;
//   * the operands of step j - D (six factor entries per slot, the yt of the rows that start their life) are requested D steps
//     ahead into rotating register sets: no memory latency on the chain.
// Chain per block row: 12 readlanes + ~4 dependent FMAs. Two-ended systems run one wave per end (grid = 2):
// block 0 solves the top system, publishes the middle solution (agent-scope release + flag), block 1 solves the reversed bottom
// system whose first `given` block rows (in sweep order) are that middle solution. The block that finishes last turns the
// solution into the step outputs.
// ---------------------------------------------------------------------------------------------------------------------
HSD double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

constexpr int kBackDepth = 4;  // prefetch distance in block rows

/// V = diag(U_jj^-1) U and yt = diag(U_jj^-1) y for factorisation kernels that only write U, U_jj^-1 and y (one workgroup per
/// block row; blocks [0, n0) serve job 0, the rest job 1). Runs after the border correction of y, if any.
__global__ void __launch_bounds__(128) k_premultiply(Tables T, BackJob j0, BackJob j1, int n0) {
  if (T.st->done) return;
  const bool first = int(blockIdx.x) < n0;
  const BackJob J = first ? j0 : j1;
  const int r = first ? blockIdx.x : blockIdx.x - n0, ncb = 6 * T.bw;
  double W[21];
#pragma unroll
  for (int e = 0; e < 21; ++e) W[e] = J.Ubk[size_t(r) * 24 + e];
  double* V = const_cast<double*>(J.Vb);
  for (int c = threadIdx.x; c <= ncb; c += blockDim.x) {
    double u[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = c < ncb ? J.Ub[size_t(6 * r + k) * ncb + c] : J.ybuf[6 * r + k];
    int p = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double v = 0.0;
#pragma unroll
      for (int k = a; k < 6; ++k) v = fma(W[p++], u[k], v);
      if (c < ncb)
        V[size_t(6 * r + a) * ncb + c] = v;
      else
        const_cast<double*>(J.yt)[6 * r + a] = v;
    }
  }
}

constexpr int kBackBlocks = 10;  // block rows per slot: 60 of the 64 lanes carry a row, a block row never straddles two slots

template <int NS>  // slots per lane: bw <= 10 NS
__global__ void __launch_bounds__(64) k_band_backward_w(Tables T, BackJob j0, BackJob j1, int m_mid) {
  HS_DYNAMIC_LDS(xs);  // solution of the own block rows (own order), flushed to HBM in bulk
  DevState* st = T.st;
  if (st->done) return;
  const BackJob J = blockIdx.x == 0 ? j0 : j1;
  const int lane = threadIdx.x;
  const int bw = T.bw, ncb = 6 * bw, np = T.np;
  const int n_rows = J.n_rows, jtop = J.n_rows + J.given - 1;
  constexpr int D = kBackDepth;
  const bool prof = prof_enabled(T.debug_flags, 16) && lane == 0;  // HS_DEBUG_FLAGS: phase timestamps (100 MHz clock) -> hs_debug_read
  long long* tlog = reinterpret_cast<long long*>(T.xpart) + 64 * 1024 + 2048 * blockIdx.x;
  if (prof) tlog[0] = wall_clock64();
  // Slot s of lane l carries the row (ring block 10 s + l / 6, component l % 6): block row beta lives in ring block beta mod bw.
  const int cz = lane % 6, lb = lane / 6;
  bool sok[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) sok[s] = lane < 60 && kBackBlocks * s + lb < bw;
  // Request stream (runs D block rows ahead of the sweep). Per slot: dq = how far the live block row lies below the requested
  // step (0: the slot belongs to that block row itself), bq = that live block row, oq = offset (in doubles) of its six factor
  // entries for the requested step inside Vb. Rows that do not exist read the zero pad behind Vb / yt (np * ncb, np): selecting
  // the ADDRESS keeps the loads free of consumers until the step that uses them.
  const int zero_v = np * ncb, zero_y = np;
  int dq[NS], bq[NS], oq[NS];
  double ub[D][NS][6], yb[D][NS];
  auto request = [&](int jr, double (*u)[6], double* y) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bool own_step = dq[s] == 0;  // the slot's row is block row jr itself: re-initialised with yt of block row jr - bw
      const bool upd = sok[s] && !own_step && jr >= 0 && bq[s] >= 0 && bq[s] < n_rows;
      const double2* src = reinterpret_cast<const double2*>(J.Vb + (upd ? oq[s] : zero_v));
      const double2 a0 = src[0], a1 = src[1], a2 = src[2];  // 16-byte aligned: rows are 48 bw bytes apart, 6 d doubles = 48 d bytes
      u[s][0] = a0.x, u[s][1] = a0.y, u[s][2] = a1.x, u[s][3] = a1.y, u[s][4] = a2.x, u[s][5] = a2.y;
      const int bn = jr - bw;
      const bool ini = sok[s] && own_step && bn >= 0 && bn < n_rows;
      y[s] = J.yt[ini ? 6 * bn + cz : zero_y];
      // next request: one block row further down
      bq[s] = own_step ? bq[s] - bw : bq[s];
      oq[s] = own_step ? oq[s] - (6 * bw * ncb - 6 * (bw - 1)) : oq[s] - 6;
      dq[s] = own_step ? bw - 1 : dq[s] - 1;
    }
  };
  double acc[NS];
  const int jm_top = ((jtop % bw) + bw) % bw;  // ring block of the first step
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int rb = kBackBlocks * s + lb;
    const int d0 = sok[s] ? (jm_top - rb + bw) % bw : 0;
    dq[s] = d0, bq[s] = jtop - d0, oq[s] = (6 * bq[s] + cz) * ncb + 6 * d0;
    const bool live = sok[s] && bq[s] >= 0 && bq[s] < n_rows;
    acc[s] = J.yt[live ? 6 * bq[s] + cz : zero_y];
  }
#pragma unroll
  for (int b = 0; b < D; ++b) request(jtop - b, ub[b], yb[b]);
  if (prof) tlog[1] = wall_clock64();
  // given block rows (solution of the other sweep): lane t holds entry t of the given part (up to 6 (bw - 1) <= 64 NS entries)
  double xg[NS > 4 ? NS : 4];
#pragma unroll
  for (int s = 0; s < (NS > 4 ? NS : 4); ++s) xg[s] = 0.0;
  if (J.given) {  // wait for the middle solution (bounded: a missing partner becomes a reported failure instead of a hang)
    const long long t0 = wall_clock64();
    bool ok = true;
    while (__hip_atomic_load(T.join_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < T.join_epoch) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > 200000000ll) {  // 2 s at the 100 MHz constant clock
        ok = false;
        break;
      }
    }
    if (!ok) {
      if (lane == 0) give_up(st);
      return;
    }
#pragma unroll
    for (int s = 0; s < (NS > 4 ? NS : 4); ++s) {
      const int e = lane + 64 * s;  // entry e of the given part = own-order row 6 n_rows + e
      const int rho = 6 * n_rows + e;
      if (e < 6 * J.given) xg[s] = __builtin_nontemporal_load(T.xsol + (J.reversed ? np - 1 - rho : rho));
    }
  }
  if (prof) tlog[2] = wall_clock64();
  int jm = jm_top;  // j mod bw
  auto step = [&](int j, double (*u)[6], double* y) {
    double x[6];
    if (j >= n_rows) {  // given by the other sweep (uniform branch, no memory operation inside)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int e = 6 * (j - n_rows) + c;
        double v = readlane_f64(xg[0], e & 63);
#pragma unroll
        for (int s = 1; s < (NS > 4 ? NS : 4); ++s) {
          const double t = readlane_f64(xg[s], e & 63);
          v = (e >> 6) == s ? t : v;
        }
        x[c] = v;
      }
    } else {
      // unit diagonal blocks: the finished running sums of block row j are x_j. Its six rows sit in one slot (wave-uniform index)
      const int sj = jm / kBackBlocks, l0 = 6 * (jm - kBackBlocks * sj);
      double av = acc[0];
#pragma unroll
      for (int s = 1; s < NS; ++s) av = sj == s ? acc[s] : av;
#pragma unroll
      for (int c = 0; c < 6; ++c) x[c] = readlane_f64(av, l0 + c);
      // the owners keep their entry of the solution in LDS: a global store per block row would sit in the same in-order vmcnt
      // queue as the operand prefetches and make every counted wait as slow as a store acknowledgement (~2 us)
      if (lane >= l0 && lane < l0 + 6 && j >= 0) xs[6 * j + lane - l0] = av;
    }
    const int sj = jm / kBackBlocks, l0 = 6 * (jm - kBackBlocks * sj);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double t0 = fma(u[s][0], x[0], fma(u[s][1], x[1], u[s][2] * x[2]));
      const double t1 = fma(u[s][3], x[3], fma(u[s][4], x[4], u[s][5] * x[5]));
      const bool mine = sj == s && lane >= l0 && lane < l0 + 6;
      acc[s] = mine ? y[s] : acc[s] - (t0 + t1);  // block row j is done: its slots start the life of block row j - bw
    }
    jm = jm == 0 ? bw - 1 : jm - 1;
    request(j - D, u, y);
  };
  const int j_pub = (blockIdx.x == 0 && m_mid >= 0 && gridDim.x == 2) ? m_mid : -1;  // block 0 publishes the middle solution after block row m_mid
  auto flush = [&](int lo, int hi) {  // own-order rows [lo, hi) of the solution: LDS -> HBM (natural order)
    wait_lds();
    for (int rho = lo + lane; rho < hi; rho += 64) T.xsol[J.reversed ? np - 1 - rho : rho] = xs[rho];
  };
  auto publish = [&]() {
    // rows [6 m_mid, 6 n_rows) of the solution: store them, release them at agent scope, raise the flag
    flush(6 * m_mid, 6 * n_rows);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    wait_vmem();
    if (lane == 0) __hip_atomic_store(T.join_flag, T.join_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // Rounds of D block rows; the last round may run past block row 0 (steps with j < 0 find nothing alive and store nothing).
  for (int j = jtop; j >= 0; j -= D) {
#pragma unroll
    for (int b = 0; b < D; ++b) {
      step(j - b, ub[b], yb[b]);
      if (prof && j - b >= 0 && prof_enabled(T.debug_flags, 32)) tlog[16 + j - b] = wall_clock64();
      if (j - b == j_pub) publish();
    }
  }
  flush(0, j_pub >= 0 ? 6 * j_pub : 6 * n_rows);
  if (prof) tlog[3] = wall_clock64();
  // ---- the block that finishes last turns the solution into the step outputs (join_flag[1] advances by gridDim.x per launch) ----
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  wait_vmem();
  int last = 1;
  if (gridDim.x == 2) {
    unsigned prev = 0;
    if (lane == 0) prev = atomicAdd(T.join_flag + 1, 1u);
    last = (__builtin_amdgcn_readfirstlane(int(prev)) & 1) == 1;
  }
  if (prof) tlog[4] = wall_clock64();
  if (!last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  double gd = 0.0, dd = 0.0;
  for (int rho = lane; rho < np; rho += 64) {
    const double stp = -__builtin_nontemporal_load(T.xsol + rho);
    T.step_p[rho] = stp;
    T.delta_p[rho] = T.scale_p[rho] * stp;
    gd = fma(T.g_full[rho], stp, gd);
    dd = fma(T.D2p[rho] * stp, stp, dd);
  }
  for (int b = lane; b < T.nb; b += 64) {
    const double stp = -T.xb[b];
    T.delta_b[b] = T.scale_b[b] * stp;
    gd = fma(T.gb_s[b], stp, gd);
    dd = fma(T.D2b[b] * stp, stp, dd);
  }
  gd = wave_sum(gd), dd = wave_sum(dd);
  if (lane == 0) {
    st->g_dot_step_pose = gd;
    st->d2_step2_pose = dd;
  }
  if (prof) tlog[5] = wall_clock64();
}

}  // namespace hs
