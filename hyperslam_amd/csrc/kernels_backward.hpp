// kernels_backward.hpp — register-resident backward sweep of the banded solve (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include "kernels_factor.hpp"

namespace hs {

// ---------------------------------------------------------------------------------------------------------------------
// Backward sweep U x = y in ONE wave per end, no LDS and no barrier on the chain.
//
// The sweep is a dependency chain over the block rows (x_j needs every x_m, m > j, that row j couples to): the previous kernels
// (k_band_backward / k_band_backward2) spent 0.5 - 0.67 us per block row on an LDS round trip + barrier between four waves.
// Here a pending row keeps its running right-hand side in a REGISTER for its whole lifetime:
//   * row rho lives in slot z = rho mod (6 bw) = lane + 64 s of the wave (s < NS): rows 6 bw apart never overlap in time, because a
//     row is pending only while the sweep is inside its band (bw block rows);
//   * step j: the six finished entries of block row j are broadcast with v_readlane (wave-uniform lane / slot), x_j = U_jj^-1 a_j
//     is formed redundantly in every lane (21 FMAs, U_jj^-1 from Ubk), every pending row subtracts U[rho][cols of j] . x_j (6 FMAs
//     per slot), the slots of block row j are re-initialised with y of block row j - bw;
//   * the operands of step j - D (six factor entries per slot, the 21 entries of U_jj^-1, the y of the rows that start their
//     life) are requested D steps ahead into rotating register sets: no memory latency on the chain.
// Chain per block row: 12 readlanes + ~6 dependent FMAs + ~6 dependent FMAs. Two-ended systems run one wave per end (grid = 2):
// block 0 solves the top system, publishes the middle solution (agent-scope release + flag), block 1 solves the reversed bottom
// system whose first `given` block rows (in sweep order) are that middle solution. The block that finishes last turns the
// solution into the step outputs.
// ---------------------------------------------------------------------------------------------------------------------
HSD double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

constexpr int kBackDepth = 4;  // prefetch distance in block rows

template <int NS>  // slots per lane: 6 bw <= 64 NS
__global__ void __launch_bounds__(64) k_band_backward_w(Tables T, BackJob j0, BackJob j1, int m_mid) {
  DevState* st = T.st;
  if (st->done) return;
  const BackJob J = blockIdx.x == 0 ? j0 : j1;
  const int lane = threadIdx.x;
  const int bw = T.bw, ncb = 6 * bw, R = 6 * bw, np = T.np;
  const int n_rows = J.n_rows, jtop = J.n_rows + J.given - 1;
  constexpr int D = kBackDepth;
  // slot constants
  int bz[NS], cz[NS];
  bool sok[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int z = lane + 64 * s;
    sok[s] = z < R;
    bz[s] = sok[s] ? z / 6 : 0, cz[s] = sok[s] ? z % 6 : 0;
  }
  // Operand sets of the next D steps. For step j and slot s: d = (j - bz) mod bw is how far the slot's live block row lies below
  // j (d = 0: the slot belongs to block row j itself and is read out / re-initialised at this step).
  double ub[D][NS][6], wb[D][21], yb[D][NS];
  int dq[NS];  // d of the step being requested (runs D steps ahead of the sweep)
  auto request = [&](int jr, double (*u)[6], double* w, double* y) {
    // jr: block row of the step the operands are for (may be negative past the end: nothing to load)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int d = dq[s];
      const int beta = jr - d;  // live block row of the slot at step jr
      const bool upd = sok[s] && jr >= 0 && d != 0 && beta >= 0 && beta < n_rows;
      const double2* src = reinterpret_cast<const double2*>(J.Ub + (upd ? size_t(6 * beta + cz[s]) * ncb + 6 * d : 0));
      const double2 a0 = src[0], a1 = src[1], a2 = src[2];  // rows are 48 bw bytes apart, 6 d doubles = 48 d bytes: 16-byte aligned
      u[s][0] = upd ? a0.x : 0.0, u[s][1] = upd ? a0.y : 0.0, u[s][2] = upd ? a1.x : 0.0;
      u[s][3] = upd ? a1.y : 0.0, u[s][4] = upd ? a2.x : 0.0, u[s][5] = upd ? a2.y : 0.0;
      // the slot of block row jr is re-initialised at step jr with y of block row jr - bw
      const int bn = jr - bw;
      const bool ini = sok[s] && jr >= 0 && d == 0 && bn >= 0 && bn < n_rows;
      const double yv = J.ybuf[ini ? 6 * bn + cz[s] : 0];
      y[s] = ini ? yv : 0.0;
      dq[s] = d == 0 ? bw - 1 : d - 1;  // next request: one block row further down
    }
    const bool own = jr >= 0 && jr < n_rows;
    const double* W = J.Ubk + size_t(own ? jr : 0) * 24;  // wave-uniform address: one cache line serves the whole wave
#pragma unroll
    for (int e = 0; e < 21; ++e) w[e] = W[e];
  };
  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int d0 = sok[s] ? ((jtop - bz[s]) % bw + bw) % bw : 0;
    dq[s] = d0;
    const int beta = jtop - d0;  // initial live block row of the slot
    const bool live = sok[s] && beta >= 0 && beta < n_rows;
    const double yv = J.ybuf[live ? 6 * beta + cz[s] : 0];
    acc[s] = live ? yv : 0.0;
  }
#pragma unroll
  for (int b = 0; b < D; ++b) request(jtop - b, ub[b], wb[b], yb[b]);
  // given block rows (solution of the other sweep): lane t holds entry t of the given part (up to 6 (bw - 1) <= 64 NS entries)
  double xg[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) xg[s] = 0.0;
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
      if (lane == 0) st->chol_failed = 2;
      return;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int e = lane + 64 * s;  // entry e of the given part = own-order row 6 n_rows + e
      const int rho = 6 * n_rows + e;
      if (e < 6 * J.given) xg[s] = __builtin_nontemporal_load(T.xsol + (J.reversed ? np - 1 - rho : rho));
    }
  }
  int jm = ((jtop % bw) + bw) % bw;  // j mod bw
  auto step = [&](int j, double (*u)[6], double* w, double* y) {
    double x[6];
    if (j >= n_rows) {  // given by the other sweep
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int e = 6 * (j - n_rows) + c;
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s)
          if ((e >> 6) == s) v = readlane_f64(xg[s], e & 63);
        x[c] = v;
      }
    } else {
      double a[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int z = 6 * jm + c;
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s)
          if ((z >> 6) == s) v = readlane_f64(acc[s], z & 63);
        a[c] = v;
      }
      int p = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r) {  // x = U_jj^-1 a (packed upper)
        double v = 0.0;
#pragma unroll
        for (int c = r; c < 6; ++c) v = fma(w[p++], a[c], v);
        x[r] = v;
      }
      if (lane < 6) {
        const int rho = 6 * j + lane;
        double xv = x[0];
#pragma unroll
        for (int c = 1; c < 6; ++c) xv = lane == c ? x[c] : xv;
        T.xsol[J.reversed ? np - 1 - rho : rho] = xv;
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double t0 = fma(u[s][0], x[0], fma(u[s][1], x[1], u[s][2] * x[2]));
      const double t1 = fma(u[s][3], x[3], fma(u[s][4], x[4], u[s][5] * x[5]));
      acc[s] -= t0 + t1;
      const int z = lane + 64 * s;
      if (z >= 6 * jm && z < 6 * jm + 6) acc[s] = y[s];  // block row j is done: its slots start the life of block row j - bw
    }
    jm = jm == 0 ? bw - 1 : jm - 1;
    request(j - D, u, w, y);
  };
  const int j_pub = (blockIdx.x == 0 && m_mid >= 0 && gridDim.x == 2) ? m_mid : -1;  // block 0 publishes the middle solution after block row m_mid
  int j = jtop;
  auto publish = [&]() {
    // rows [6 m_mid, 6 n_rows) of the solution were stored by lanes 0..5 of this wave: release them at agent scope, raise the flag
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(T.join_flag, T.join_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  while (j >= 0) {
#pragma unroll
    for (int b = 0; b < D; ++b) {
      if (j >= 0) {
        step(j, ub[b], wb[b], yb[b]);
        if (j == j_pub) publish();
        --j;
      }
    }
  }
  // ---- the block that finishes last turns the solution into the step outputs (join_flag[1] advances by gridDim.x per launch) ----
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int last = 1;
  if (gridDim.x == 2) {
    unsigned prev = 0;
    if (lane == 0) prev = atomicAdd(T.join_flag + 1, 1u);
    last = (__builtin_amdgcn_readfirstlane(int(prev)) & 1) == 1;
  }
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
}

}  // namespace hs
