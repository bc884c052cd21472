// kernels_factor.hpp — block-banded Cholesky (look-ahead / legacy / wide) and the backward sweeps (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include "kernels_common.hpp"

namespace hs {

// ---------------------------------------------------------------------------------------------------------------------
// Block-banded Cholesky S = U'U, fused forward solve, then backward solve.  Single workgroup: the factorisation is a
// dependency chain over the n_cp block rows, so the design minimises the latency of one step instead of spreading
// work over CUs.  Row rho of the band stores S[rho][6*(rho/6) + c].
//   * the trailing window (bw block rows x bw band blocks of 6x6) lives in REGISTERS: thread t owns tile
//     (slot = t / bw, band block = t % bw) for the whole lifetime of a block row (slot = row % bw), so the rank-6 updates
//     never read-modify-write LDS; LDS only carries the current pivot row (rowbuf) and its solved form X (xbuf).
//   * step i :  owners of row i publish their tiles -> rowbuf, then immediately start loading row i + bw into the freed
//               registers (global latency hidden behind the rest of the step)          --- LDS barrier ---
//               P1: every thread factors the 6x6 diagonal block redundantly in registers (no serial section) and
//                   thread c solves column c of X = U_ii^-T [S_i,i+1.. | g_i] -> xbuf     --- LDS barrier ---
//               P2: each live tile (j, kk):  S_(i+j),kk -= X_j' X_(j+kk);  rhs: g_(i+j) -= X_j' y_i;  U row i streamed to HBM
//   * barriers drain LDS only (lds_barrier), so global prefetches stay in flight across them.
//   * backward: column oriented, U entries and U_jj^-1 prefetched three steps ahead, one barrier per block row.
// Outputs: Ub (factor), step_p = -S^-1 g (scaled step), delta_p = scale_p o step_p, and the two pose-side reductions
// of the model cost change.  f64 MFMA is not used here: the update has K = 6 and is bound by the pivot-row latency, the
// 16x16x4 f64 MFMA runs at the VALU FMA rate on gfx950 (78.6 TF both) and would only add operand shuffling.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kCholThreads = 256;

constexpr int kCholIo = 128;  // two extra waves that own all global traffic of the factorisation (loader, storer)

template <int TPT>  // tiles per thread: bw * bw <= TPT * kCholThreads
__global__ void __launch_bounds__(kCholThreads + kCholIo) k_band_factor(Tables T) {
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const int tid = threadIdx.x;
  constexpr int nthr = kCholThreads;
  const int bw = T.bw, ncb = 6 * bw, ld = ncb + 2;
  const int n_blk = T.np / 6;
  double* rowbuf = smem;            // 6 x ld : pivot row as published by its owners [band | rhs | pad]
  double* xbuf = smem + 6 * ld;     // 6 x ld : [U_ii | X | y_i]
  double* stage = smem + 12 * ld;   // 2 x 6 x ld : block rows i + bw (+1) staged by the IO wave ahead of their use
  double* xs = smem + 24 * ld;      // np : y (forward solve)
  __shared__ int fail;
  if (tid == 0) fail = 0;
  const bool io = tid >= nthr;  // the IO wave streams S rows in (global -> registers -> LDS stage) and factor rows out
  constexpr int kIoEnt = 12;    // entries per IO lane per block row: 6 * (6 * 21 + 1) = 762 <= 12 * 64
  const int n_ent = 6 * (ncb + 1);
  if (io) {  // ============ IO waves: a loader (wave 4) and a storer (wave 5); neither ever blocks the compute waves' math ============
    // Two separate waves because vmcnt is one in-order counter per wave: a wave that both loads and stores would wait for its
    // own (slow, just-issued) stores whenever it needs a prefetched load.
    const int l = (tid - nthr) & 63;
    const bool loader = tid < nthr + 64;
    // loop-invariant addressing of this lane's entries of a block row (no integer divisions inside the step loop)
    const double* e_base[kIoEnt];
    int e_stride[kIoEnt], e_lds[kIoEnt], e_dst[kIoEnt];
#pragma unroll
    for (int m = 0; m < kIoEnt; ++m) {
      const int e = l + m * 64;
      const bool ok = e < n_ent;
      const int a = ok ? e / (ncb + 1) : 0, c = ok ? e % (ncb + 1) : 0;
      e_lds[m] = ok ? a * ld + c : -1;
      e_base[m] = c < ncb ? T.Sb + a * ncb + c : T.g_s + a;
      e_stride[m] = c < ncb ? 6 * ncb : 6;
      e_dst[m] = c < ncb ? a * ncb + c : -1 - a;  // offset in the Ub block row, or -(1 + a): y entry
    }
    if (loader) {
      double v[kIoEnt];
      auto fetch = [&](int r) {
        const int rr = r < n_blk ? r : 0;
#pragma unroll
        for (int m = 0; m < kIoEnt; ++m) v[m] = (e_lds[m] >= 0 && r < n_blk) ? e_base[m][size_t(rr) * e_stride[m]] : 0.0;
      };
      auto put = [&](int r) {
        double* dst = stage + (r & 1) * 6 * ld;
#pragma unroll
        for (int m = 0; m < kIoEnt; ++m)
          if (e_lds[m] >= 0) dst[e_lds[m]] = v[m];
      };
      fetch(bw), put(bw), fetch(bw + 1);
      lds_barrier();  // initial window loaded / staged
      for (int i = 0; i < n_blk; ++i) {
        lds_barrier();  // B1
        put(i + bw + 1);
        fetch(i + bw + 2);
        lds_barrier();  // B2
      }
      lds_barrier();
    } else {
      lds_barrier();
      for (int i = 0; i < n_blk; ++i) {
        lds_barrier();  // B1
        lds_barrier();  // B2: xbuf = [U_ii | X | y_i] is complete
        double* Urow = T.Ub + size_t(6) * i * ncb;
#pragma unroll
        for (int m = 0; m < kIoEnt; ++m) {
          if (e_lds[m] < 0) continue;
          const double x = xbuf[e_lds[m]];
          if (e_dst[m] >= 0)
            Urow[e_dst[m]] = x;
          else
            xs[6 * i + (-1 - e_dst[m])] = x;
        }
      }
      lds_barrier();
      for (int rho = l; rho < T.np; rho += 64) T.ybuf[rho] = xs[rho];  // y = U^-T g
      if (l == 0 && fail) st->chol_failed = 1;  // never cleared here: k_factor_decoupled_rows may have raised it earlier in this iteration
    }
    return;
  }

  // ---- static tile ownership ------------------------------------------------------------------------------------
  int t_slot[TPT], t_kk[TPT];
  bool t_ok[TPT];
  double acc[TPT][36], rhs[TPT][6];
#pragma unroll
  for (int m = 0; m < TPT; ++m) {
    const int tl = tid + m * nthr;
    t_ok[m] = tl < bw * bw;
    t_slot[m] = t_ok[m] ? tl / bw : 0;
    t_kk[m] = t_ok[m] ? tl % bw : 0;
  }
  auto load_tile = [&](int m, int r) {  // block row r into tile m's registers (zeros past the end)
    const bool in = t_ok[m] && r < n_blk;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double* src = T.Sb + size_t(6 * r + a) * ncb + 6 * t_kk[m];
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[m][6 * a + c] = in ? src[c] : 0.0;
      rhs[m][a] = (in && t_kk[m] == 0) ? T.g_s[6 * r + a] : 0.0;
    }
  };
#pragma unroll
  for (int m = 0; m < TPT; ++m) load_tile(m, t_slot[m]);
  auto refill_tile = [&](int m, int r) {  // block row r from the LDS stage written by the IO wave (no global access here)
    const double* src = stage + (r & 1) * 6 * ld;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c = 0; c < 6; c += 2) {
        const double2 v = *reinterpret_cast<const double2*>(&src[a * ld + 6 * t_kk[m] + c]);
        acc[m][6 * a + c] = v.x, acc[m][6 * a + c + 1] = v.y;
      }
      rhs[m][a] = (t_kk[m] == 0 ? 1.0 : 0.0) * src[a * ld + ncb];
    }
  };
  lds_barrier();  // initial window loaded / staged
  const bool prof = prof_enabled(T.debug_flags, 16) && tid == 0;
  long long* tlog = reinterpret_cast<long long*>(T.xpart);

#define UIDX(a, c) ((a) * 6 - (a) * ((a)-1) / 2 + ((c) - (a)))
  for (int i = 0; i < n_blk; ++i) {
    const int si = i % bw;
    if (prof) tlog[8 * i + 0] = wall_clock64();
    // ---- publish the pivot row, then refill the freed registers with block row i + bw ----
#pragma unroll
    for (int m = 0; m < TPT; ++m)
      if (t_ok[m] && t_slot[m] == si) {
        const int c_rhs = t_kk[m] == 0 ? ncb : ncb + 1;  // lanes without a right-hand side write theirs to the pad column (no branch)
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = 0; c < 6; c += 2)
            *reinterpret_cast<double2*>(&rowbuf[a * ld + 6 * t_kk[m] + c]) = make_double2(acc[m][6 * a + c], acc[m][6 * a + c + 1]);
          rowbuf[a * ld + c_rhs] = rhs[m][a];
        }
        refill_tile(m, i + bw);
      }
    lds_barrier();
    if (prof) tlog[8 * i + 1] = wall_clock64();
    // ---- P1: redundant register factorisation of the diagonal block + one column of X per thread ----
    double U[21], inv[6];
    {
      int p = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = a; c < 6; ++c) U[p++] = rowbuf[a * ld + c];
    }
    bool bad = false;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double d = U[UIDX(a, a)];
#pragma unroll
      for (int k = 0; k < a; ++k) d = fma(-U[UIDX(k, a)], U[UIDX(k, a)], d);
      if (!(d > 0.0)) bad = true, d = 1.0;
      // hardware estimate + two Newton steps (full double precision; the library rsqrt's scaling / special cases are not needed
      // for a positive, well-scaled pivot) and a final correction of the square root
      double r = __builtin_amdgcn_rsq(d);
      r = r * fma(-0.5 * d * r, r, 1.5);
      r = r * fma(-0.5 * d * r, r, 1.5);
      double u = d * r;
      u = fma(0.5 * r, fma(-u, u, d), u);
      inv[a] = r;
      U[UIDX(a, a)] = u;
#pragma unroll
      for (int c = a + 1; c < 6; ++c) {
        double v = U[UIDX(a, c)];
#pragma unroll
        for (int k = 0; k < a; ++k) v = fma(-U[UIDX(k, a)], U[UIDX(k, c)], v);
        U[UIDX(a, c)] = v * r;
      }
    }
    if (bad && tid == 0) fail = 1;
    if (tid + 6 <= ncb) {  // columns 6 .. ncb (ncb = rhs)
      const int c = tid + 6;
      double x[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double v = rowbuf[a * ld + c];
#pragma unroll
        for (int k = 0; k < a; ++k) v = fma(-U[UIDX(k, a)], x[k], v);
        x[a] = v * inv[a];
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) xbuf[a * ld + c] = x[a];
    } else if (tid >= nthr - 36) {  // the last 36 threads publish U_ii (upper, zeros below)
      const int e = tid - (nthr - 36), a = e / 6, c = e % 6;
      double v = 0.0;
#pragma unroll
      for (int aa = 0; aa < 6; ++aa)
#pragma unroll
        for (int cc = aa; cc < 6; ++cc)
          if (aa == a && cc == c) v = U[UIDX(aa, cc)];
      xbuf[a * ld + c] = v;
    }
    lds_barrier();
    if (prof) tlog[8 * i + 2] = wall_clock64();
    // off the critical path (after the barrier): one otherwise idle lane inverts the factored diagonal block
    if (tid == nthr - 1) {  // W = U_ii^-1 (upper triangular) for the backward sweep: x_i = W y_i, no divisions there
      double W[21];
#pragma unroll
      for (int c = 5; c >= 0; --c) {
        W[UIDX(c, c)] = inv[c];
#pragma unroll
        for (int a = c - 1; a >= 0; --a) {
          double v = 0.0;
#pragma unroll
          for (int k = a + 1; k <= c; ++k) v = fma(U[UIDX(a, k)], W[UIDX(k, c)], v);
          W[UIDX(a, c)] = -v * inv[a];
        }
      }
#pragma unroll
      for (int e = 0; e < 21; ++e) T.Ubk[size_t(i) * 24 + e] = W[e];
    }
    // ---- P2: rank-6 update of the register tiles ----
#pragma unroll
    for (int m = 0; m < TPT; ++m) {
      int j = t_slot[m] - si;
      if (j < 0) j += bw;
      if (!t_ok[m] || j == 0 || j + t_kk[m] > bw - 1 || (T.debug_flags & 2)) continue;
      const int ca = 6 * j, cb = 6 * (j + t_kk[m]);
      // the right-hand side rides along in every lane (only the lanes with kk = 0 use theirs): as a branch around its loads and FMAs it
      // drained the LDS queue once per row of X (see k_band_factor_la)
      double yv[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) yv[a] = xbuf[a * ld + ncb];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double xa[6], xb[6];
#pragma unroll
        for (int c = 0; c < 6; c += 2) {
          const double2 va = *reinterpret_cast<const double2*>(&xbuf[a * ld + ca + c]);
          const double2 vb = *reinterpret_cast<const double2*>(&xbuf[a * ld + cb + c]);
          xa[c] = va.x, xa[c + 1] = va.y, xb[c] = vb.x, xb[c + 1] = vb.y;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
          for (int c = 0; c < 6; ++c) acc[m][6 * r + c] = fma(-xa[r], xb[c], acc[m][6 * r + c]);
          rhs[m][r] = fma(-xa[r], yv[a], rhs[m][r]);
        }
      }
    }
    if (prof) tlog[8 * i + 3] = wall_clock64();
    if (prof) tlog[8 * i + 4] = wall_clock64();
  }
#undef UIDX
  lds_barrier();
}

// ---------------------------------------------------------------------------------------------------------------------
// Look-ahead variant (the one launched): the panel work of step i + 1 (finish the pivot row, factor its diagonal block,
// solve X) is taken off the compute waves and runs in a dedicated PANEL wave concurrently with the rank-6 update of step i.
//   waves 0-2  compute : register tiles of rows i + 2 .. i + bw - 1 (+ prefetched rows); P2(i) with X_i, then the owners of
//                        row i + 2 publish it to rowbuf[(i + 2) & 1] and refill their registers with row i + 2 + bw.
//                        Only band blocks kk <= bw - 3 ever receive an update before their row becomes the panel row, so only
//                        those bw (bw - 2) tiles live in registers; the last two blocks go HBM -> rowbuf through the loader.
//   wave  3    panel   : row i + 1 (in LDS since step i - 1, updated through X_(i-1)) -= X_i,1' X_i ; U_(i+1) = chol ; X_(i+1).
//                        Waves are placed round-robin on the 4 SIMDs, so wave 3 has SIMD 3 to itself: sharing a SIMD with a
//                        compute wave stretched this latency chain 2-3x (tools/microbench/factor_probe.hip: 750 clk alone).
//   wave  4    loader  : streams block rows from HBM into the LDS stage two steps ahead (+ the tail blocks of row i + 2)
//   wave  5    storer  : streams X_i (the factor row) to HBM, keeps y in LDS, inverts U_ii for the backward sweep
// One LDS-only barrier per block row; critical path per step = max(panel chain, rank-6 update) instead of their sum.
// LDS (doubles): rowbuf 2 x 6 x ld | xbuf 2 x 6 x ld | stage 3 x 6 x ld | y np | diagonal scratch 36.
//
// Two-ended mode (grid = 2, visual-only systems): the chain over the block rows is halved by eliminating from both ends at once.
// Workgroup 1 factors the REVERSED system (written by k_finalize_reduced next to the natural one) for the last n - m - w block rows (w = bw - 1), dumps its trailing
// window (the Schur contribution of those rows to the middle block rows m .. m + w - 1) and raises a flag. Workgroup 0 factors
// rows 0 .. m - 1, waits for the flag, adds the other end's contribution to its own trailing window (entries of the middle
// rows that couple to the eliminated end become zero) and simply continues through the middle rows: it ends with the Cholesky
// factor of the system in which the far end has been eliminated. k_band_backward2 solves the top part normally and the bottom
// part in reversed coordinates once the middle solution is known.
// ---------------------------------------------------------------------------------------------------------------------
/// Waits until the partner workgroup of a two-ended kernel has published its half (join_flag >= join_epoch). Both workgroups of the
/// launch are resident together on every gfx950 configuration this library runs on, but a wait on another workgroup must not be able
/// to hang the device: after 2 s (100 MHz constant clock) the wait gives up and marks the factorisation as failed — the step is then
/// rejected as invalid (k_decide) and hs_solve reports the reason — and the caller carries on so that no barrier is left waiting.
HSD void wait_for_partner(const Tables& T) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(T.join_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < T.join_epoch) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > 200000000ll) {
      give_up(T.st);
      break;
    }
  }
}

/// Compute waves of the look-ahead kernel for a band of bw control points: ring slot s lives in compute wave s % NCW, so a wave holds
/// ceil(bw / NCW) slots of bw - 2 tiles, one tile per lane. 0 = the band does not fit (one-ended kernels below).
constexpr int la_compute_waves(int bw) { return ((bw + 2) / 3) * (bw - 2) <= 64 ? 3 : ((bw + 3) / 4) * (bw - 2) <= 64 ? 4 : 0; }
/// Waves of the workgroup: three compute waves on SIMDs 0 - 2, the panel wave alone on SIMD 3, loader and storer (waves 4, 5: SIMDs 0, 1);
/// a fourth compute wave (bands of 15 and 16 control points: order-6 splines) is wave 6 and shares SIMD 2 with compute wave 2.
constexpr int la_threads(int ncw) { return ncw == 3 ? 6 * 64 : 7 * 64; }

HSD void begin_iteration(const Tables& T, double cost, double gmax, bool set_scaling_ready);  // kernels_update.hpp

/// Iteration bookkeeping inside the factorisation (Tables::bookkeep; one wave, lane l): see the comment in the body.
HSD void factor_bookkeep(const Tables& T, int l) {
  DevState* st = T.st;
  // Iteration bookkeeping of a linearisation whose rows k_assemble wrote itself (no k_finalize_reduced launch): cost of the current point,
  // gradient max norm, record of the previous iteration, termination tests — what the packing workgroup of k_finalize_reduced does
  // (pack_exchange_body + begin_iteration). This wave has nothing to do until X_0 exists: it reduces the ~6 000 doubles alone, in a
  // fixed order, without a workgroup barrier. The factorisation does not wait for the verdict: if a termination test fires, this one
  // factorisation was for nothing and every later kernel of the solve exits on `done` as usual. (As a prologue of the whole workgroup —
  // reductions across six waves, early exit — the same work cost 4.5 us on the chain.)
  double s = 0.0, gm = 0.0;
  for (int i0 = l; i0 < T.n_cost_part; i0 += 8 * 64) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + 64 * u < T.n_cost_part ? T.cost_part[i0 + 64 * u] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  // landmark side of the gradient: one value per chunk of the fused build (launch_build only takes this path on the fused one)
  for (int i0 = l; i0 < T.n_chunk; i0 += 16 * 64) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = i0 + 64 * u < T.n_chunk ? T.ch_gmax[i0 + 64 * u] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) gm = fmax(gm, v[u]);
  }
  for (int i0 = l; i0 < T.np; i0 += 16 * 64) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = i0 + 64 * u < T.np ? T.gabs[i0 + 64 * u] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) gm = fmax(gm, v[u]);
  }
  s = wave_sum(s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_xor(gm, o));
  if (l == 0) {
    T.xbuf[T.xo_cost] = s;
    st->local_cost = s;
    begin_iteration(T, s, gm, false);
  }
}

template <int TPT, int NCW>  // one tile per compute lane; NCW compute waves
__global__ void __launch_bounds__(la_threads(NCW)) k_band_factor_la(Tables T) {
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const FactorJob J = T.fj[blockIdx.x];
  const int n_steps = J.n_steps;
  const int m_at = J.merge_at;                                  // job 0, two-ended: first middle block row (junction before it)
  const bool dump = J.win != nullptr && m_at < 0;                // job 1, two-ended: hand the trailing window to job 0 at the end
  const int w_mid = T.bw - 1;
  // Job 1 hands its window over as the CORRECTION the other end has to add, already in the other end's coordinates: entry
  // (vr, off) of the reversed window (scalar row vr of the middle block, band offset off) is entry (r, cl) of the natural one with
  // cl = dm - 1 - vr, r = dm - 1 - off - 6 floor(vr / 6); stored at win[r][cl - 6 floor(r / 6)] (both triangles of a diagonal block).
  auto hand_over = [&](int vr, int off, double value_minus_original) {
    const int dm = 6 * w_mid, wl = 6 * T.bw + 1;
    const int cl = dm - 1 - vr, r = dm - 1 - off - 6 * (vr / 6);
    if (r < 0 || cl < 0) return;
    const int rb = 6 * (r / 6);
    if (cl >= rb) J.win[size_t(r) * wl + (cl - rb)] = value_minus_original;
    if (cl / 6 == r / 6) J.win[size_t(cl) * wl + (r - rb)] = value_minus_original;  // mirrored entry of the diagonal block
  };
  auto junction_wait = [&]() {                                  // job 0: the other end has published its window
    wait_for_partner(T);
  };
  constexpr int nthr = 64 * NCW;
  const int wave = int(threadIdx.x) >> 6, l = int(threadIdx.x) & 63;
  const int cwave = wave < 3 ? wave : 3;        // compute wave index of waves 0, 1, 2 (and 6)
  const int tid = wave < 3 ? int(threadIdx.x) : wave == 6 ? 192 + l : int(threadIdx.x);  // compute lanes: 0 .. nthr - 1, contiguous
  constexpr int PC = 2;  // columns of the pivot row per panel lane: 6 * bw + 1 <= 128 (bw <= 20)
  const int bw = T.bw, ncb = 6 * bw, ld = ncb + 2;
  const int n_blk = T.np / 6;
  double* rowbuf = smem;            // row r (published by its owners, updated through X_(r-2)) in rowbuf[r & 1]
  double* xbuf = smem + 12 * ld;    // [U_rr | X_r | y_r] in xbuf[r & 1]
  double* stage = smem + 24 * ld;   // block row r staged by the loader in stage[r % 3] (three buffers: a row is read by its new owners
                                    // one step after it was staged for them, see the compute waves)
  double* xs = smem + 42 * ld;      // np : y (forward solve)
  double* dscr = xs + T.np;         // 36 : (unused: the panel broadcasts its diagonal block with v_readlane)
  double* dinv = dscr + 36;         // 2 x 6 : 1 / diag(U_rr) in dinv[r & 1]
  __shared__ int fail;
  if (threadIdx.x == 0) fail = 0;
  if (wave == 4 || wave == 5) {  // ================================ IO waves ================================
    // lane l owns columns l and l + 64 of a block row ([band | rhs], 6 * bw + 1 <= 128 columns), all six rows: no index tables
    int c_col[2];
    bool c_ok[2], c_rhs[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int c = l + 64 * m;
      c_ok[m] = c <= ncb, c_rhs[m] = c == ncb;
      c_col[m] = c_ok[m] ? c : 0;
    }
    if (wave == 4) {
      // two register sets: a block row is requested two steps before it is staged (the rows were written by another XCD's
      // workgroups and come from HBM / MALL; one step of prefetch distance does not always cover that)
      double va[12], vb[12];
      auto fetch = [&](double* v, int r) {
        const bool in = r < n_blk;
        const int rr = in ? r : 0;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const double* src = c_rhs[m] ? J.g_s + 6 * rr : J.Sb + size_t(6 * rr) * ncb + c_col[m];
          const int stride = c_rhs[m] ? 1 : ncb;
#pragma unroll
          for (int a = 0; a < 6; ++a) v[6 * m + a] = (c_ok[m] && in) ? src[a * stride] : 0.0;
        }
      };
      auto put = [&](const double* v, double* dst) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if (c_ok[m]) {
#pragma unroll
            for (int a = 0; a < 6; ++a) dst[a * ld + c_col[m]] = v[6 * m + a];
          }
      };
      // tail blocks (band blocks bw - 2, bw - 1: 6 x 12 entries) of the row that is published this step: never modified before
      // the row becomes the panel row, so they bypass the register tiles
      const double* t_base[2];
      int t_lds[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int e = l + 64 * m;
        const int a = e < 72 ? e / 12 : 0, c = 6 * (bw - 2) + (e < 72 ? e % 12 : 0);
        t_lds[m] = e < 72 ? a * ld + c : -1;
        t_base[m] = J.Sb + a * ncb + c;
      }
      double ta[2], tb[2];
      auto tfetch = [&](double* v, int r) {
#pragma unroll
        for (int m = 0; m < 2; ++m) v[m] = (t_lds[m] >= 0 && r < n_blk) ? t_base[m][size_t(r < n_blk ? r : 0) * 6 * ncb] : 0.0;
      };
      auto tput = [&](const double* v, double* dst, int r) {
        // two-ended job 0: the tail blocks of the middle rows couple to the end that the other workgroup eliminates -> zero
        // (rows m, m + 1 are in LDS at the junction and are fixed there)
        const bool zero = m_at >= 0 && r >= m_at + 2 && r < m_at + w_mid;
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if (t_lds[m] >= 0) dst[t_lds[m]] = zero ? 0.0 : v[m];
      };
      auto junction_io = [&](int i_done) {  // after the barrier that ends step i_done
        if (m_at >= 0 && i_done + 1 == m_at) {
          junction_wait();
          lds_barrier();  // merge done
          lds_barrier();  // panel(m) done
        }
      };
      fetch(va, 0), fetch(vb, 1);
      put(va, rowbuf), put(vb, rowbuf + 6 * ld);
      fetch(va, bw + 2), fetch(vb, bw + 3);
      tfetch(tb, 2), tfetch(ta, 3);
      put(va, stage + ((bw + 2) % 3) * 6 * ld);
      fetch(va, bw + 4);
      lds_barrier();  // init
      lds_barrier();  // prologue
      for (int i = 0; i < n_steps; i += 2) {
        put(vb, stage + ((i + 3 + bw) % 3) * 6 * ld);
        tput(tb, rowbuf + (i & 1) * 6 * ld, i + 2);
        fetch(vb, i + 5 + bw);
        tfetch(tb, i + 4);
        if (prof_enabled(T.debug_flags, 16) && l == 0) reinterpret_cast<long long*>(T.xpart)[8 * i + 7] = wall_clock64();
        lds_barrier();
        junction_io(i);
        if (i + 1 < n_steps) {
          put(va, stage + ((i + 4 + bw) % 3) * 6 * ld);
          tput(ta, rowbuf + ((i + 1) & 1) * 6 * ld, i + 3);
          fetch(va, i + 6 + bw);
          tfetch(ta, i + 5);
          lds_barrier();
          junction_io(i + 1);
        }
      }
      lds_barrier();
      if (dump) lds_barrier();  // window written by the panel / compute waves
    } else {
      if (T.bookkeep && blockIdx.x == 0) factor_bookkeep(T, l);
      lds_barrier();  // init
      lds_barrier();  // prologue: X_0 complete
      for (int i = 0; i < n_steps; ++i) {
        const double* xb = xbuf + (i & 1) * 6 * ld;
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if (c_ok[m]) {
            double x[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) x[a] = xb[a * ld + c_col[m]];
            if (c_rhs[m]) {  // y stays in LDS until the end
#pragma unroll
              for (int a = 0; a < 6; ++a) xs[6 * i + a] = x[a];
            } else {
              double* dst = J.Ub + size_t(6 * i) * ncb + c_col[m];
#pragma unroll
              for (int a = 0; a < 6; ++a) dst[a * ncb] = x[a];
            }
          }
        {  // W = U_ii^-1 (upper triangular) for the backward sweep (x_i = W y_i, no divisions there): lane c < 6 solves U w = e_c;
           // 1 / u_aa comes from the panel wave (dinv), entries below the diagonal come out as exact zeros
          const double* di = dinv + (i & 1) * 6;
          const int c = l < 6 ? l : 0;
          double w[6];
#pragma unroll
          for (int a = 5; a >= 0; --a) {
            double t = a == c ? 1.0 : 0.0;
#pragma unroll
            for (int k = a + 1; k < 6; ++k) t = fma(-xb[a * ld + k], w[k], t);
            w[a] = t * di[a];
          }
          if (l < 6) {
            // packed upper storage index of (a, c), a <= c
#pragma unroll
            for (int a = 0; a < 6; ++a)
              if (a <= c) J.Ubk[size_t(i) * 24 + (a * 6 - a * (a - 1) / 2 + (c - a))] = w[a];
          }
        }
        if (prof_enabled(T.debug_flags, 16) && l == 0) reinterpret_cast<long long*>(T.xpart)[8 * i + 6] = wall_clock64();
        lds_barrier();
        if (m_at >= 0 && i + 1 == m_at) {
          junction_wait();
          lds_barrier();  // merge done
          lds_barrier();  // panel(m) done
        }
      }
      lds_barrier();
      if (dump) lds_barrier();
      for (int rho = l; rho < 6 * n_steps; rho += 64) J.ybuf[rho] = xs[rho];  // y = U^-T g
      if (l == 0 && fail) st->chol_failed = 1;  // (consumed and cleared by decide_step, kernels_update.hpp: an invalid step)
    }
    return;
  }

  if (wave == 3) {  // ================================ panel wave (alone on SIMD 3) ================================
#define UIDX(a, c) ((a) * 6 - (a) * ((a)-1) / 2 + ((c) - (a)))
    const bool pprof = prof_enabled(T.debug_flags, 16) && l == 0;
    long long* plog = reinterpret_cast<long long*>(T.xpart);
    // column bookkeeping of this lane (loop invariant): c = l + 64 m; lanes past the row write to the pad column ncb + 1
    int c_rd[PC], c_src[PC], c_wr[PC];
    bool c_live[PC];
#pragma unroll
    for (int m = 0; m < PC; ++m) {
      const int c = l + 64 * m;
      c_rd[m] = c <= ncb ? c : ncb + 1;
      c_wr[m] = c_rd[m];
      const int cs = c == ncb ? ncb : 6 + c;  // column of X_(r-1) that lands on column c of row r
      c_live[m] = c <= ncb && (c == ncb || cs < ncb);
      c_src[m] = c_live[m] ? cs : ncb + 1;
    }
    // Branch-free on purpose: a taken branch costs ~40 cycles on this chain (tools/microbench/clock_probe.hip).
    auto panel = [&](int r, bool update) {  // pivot row r: rowbuf[r & 1] (- X_(r-1),1' X_(r-1)) -> xbuf[r & 1]
      if (pprof) plog[8 * r + 2] = wall_clock64();
      const double* row = rowbuf + (r & 1) * 6 * ld;
      const double* xp = xbuf + ((r - 1) & 1) * 6 * ld;
      double* xo = xbuf + (r & 1) * 6 * ld;
      double v[PC][6];
#pragma unroll
      for (int m = 0; m < PC; ++m)
#pragma unroll
        for (int a = 0; a < 6; ++a) v[m][a] = row[a * ld + c_rd[m]];
      if (update) {
        double B[6][6];  // block 1 of X_(r-1): couples row r - 1 to row r
#pragma unroll
        for (int ap = 0; ap < 6; ++ap)
#pragma unroll
          for (int a = 0; a < 6; a += 2) {
            const double2 t = *reinterpret_cast<const double2*>(&xp[ap * ld + 6 + a]);
            B[ap][a] = t.x, B[ap][a + 1] = t.y;
          }
#pragma unroll
        for (int m = 0; m < PC; ++m) {
          double xc[6];
#pragma unroll
          for (int ap = 0; ap < 6; ++ap) {
            const double t = xp[ap * ld + c_src[m]];
            xc[ap] = c_live[m] ? t : 0.0;
          }
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int ap = 0; ap < 6; ++ap) v[m][a] = fma(-B[ap][a], xc[ap], v[m][a]);
        }
      }
      if (pprof) plog[8 * r + 3] = wall_clock64();
      // The updated diagonal block sits in lanes 0 - 5 (lane c: column c). Every lane factors it redundantly: the 21 entries are
      // broadcast with v_readlane (wave-uniform values in SGPRs, each the addend of one FMA chain below) instead of a round trip
      // through LDS (6 stores, wait, 21 loads behind the compute waves' traffic: 0.08 us of the chain per block row).
      double U[21], inv[6], dmin;
      {
        int pidx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int c = a; c < 6; ++c)
            U[pidx++] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v[0][a]), c), __builtin_amdgcn_readlane(__double2loint(v[0][a]), c));
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double d = U[UIDX(a, a)];
#pragma unroll
        for (int k = 0; k < a; ++k) d = fma(-U[UIDX(k, a)], U[UIDX(k, a)], d);
        // A non-positive pivot is not patched on this chain: it turns the rest of the factor into NaN / inf, `fail` is raised
        // below and the step is rejected as invalid (k_decide also requires a finite model cost change).
        dmin = a == 0 ? d : fmin(dmin, d);  // fmin drops a NaN operand only if the other is a number: checked with !(x > 0)
        // 1 / sqrt(d): hardware estimate (2^-24 relative) + one third-order step, e = 1 - d y^2, y (1 + e/2 + 3 e^2/8): error ~ e^3
        const double y = __builtin_amdgcn_rsq(d);
        const double e = fma(-d * y, y, 1.0);
        const double rs = fma(y * e, fma(0.375, e, 0.5), y);
        inv[a] = rs;
        const double nrs = -rs;  // the off-diagonal entries are kept NEGATED: products of two of them are unchanged, and the
                                 // column solves below become plain multiply-adds without sign flips
#pragma unroll
        for (int c = a + 1; c < 6; ++c) {
          double t = U[UIDX(a, c)];
#pragma unroll
          for (int k = 0; k < a; ++k) t = fma(-U[UIDX(k, a)], U[UIDX(k, c)], t);
          U[UIDX(a, c)] = t * nrs;
        }
      }
#pragma unroll
      for (int a = 0; a < 6; a += 2) *reinterpret_cast<double2*>(&dinv[(r & 1) * 6 + a]) = make_double2(inv[a], inv[a + 1]);  // every lane, same value
      if (!(dmin > 0.0) && l == 0) fail = 1;
      if (pprof) plog[8 * r + 4] = wall_clock64();
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        // x = U^-T v. For the diagonal-block columns (c < 6) this reproduces column c of U itself in its upper part (same
        // operations as the factorisation above); the diagonal and the part below it are never read (the backward sweep and the
        // border use U_ii^-1 from Ubk, W is built from the strict upper part and dinv).
        double x[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          double t = v[m][a];
#pragma unroll
          for (int k = 0; k < a; ++k) t = fma(U[UIDX(k, a)], x[k], t);  // U holds -u_ka
          x[a] = t * inv[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) xo[a * ld + c_wr[m]] = x[a];
      }
      if (pprof) plog[8 * r + 5] = wall_clock64();
    };
    lds_barrier();  // init: rows 0, 1 in rowbuf
    panel(0, false);
    lds_barrier();  // prologue
    for (int i = 0; i < n_steps; ++i) {
      const bool junction = m_at >= 0 && i + 1 == m_at;  // no look-ahead across the junction: row m changes there
      if (i + 1 < n_steps && !junction) panel(i + 1, true);
      lds_barrier();
      if (junction) {
        junction_wait();
        lds_barrier();  // merge done (compute waves)
        panel(m_at, true);
        lds_barrier();
      }
    }
    lds_barrier();
    if (dump) {
      // trailing window rows 0 and 1 (block rows n_steps, n_steps + 1) are in LDS: row 0 still needs the update by X_(n_steps-1)
      const int r = n_steps;
      const double* row = rowbuf + (r & 1) * 6 * ld;
      const double* xp = xbuf + ((r - 1) & 1) * 6 * ld;
      const double* row1 = rowbuf + ((r + 1) & 1) * 6 * ld;
      double B[6][6];
#pragma unroll
      for (int ap = 0; ap < 6; ++ap)
#pragma unroll
        for (int a = 0; a < 6; ++a) B[ap][a] = xp[ap * ld + 6 + a];
      double o0[PC][6], o1[PC][6];  // the originals of both rows first (the stores below may alias them as far as the compiler knows)
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const int c = l + 64 * m;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          o0[m][a] = c > ncb ? 0.0 : (c == ncb ? J.g_s[6 * r + a] : J.Sb[size_t(6 * r + a) * ncb + c]);
          o1[m][a] = c > ncb ? 0.0 : (c == ncb ? J.g_s[6 * (r + 1) + a] : J.Sb[size_t(6 * (r + 1) + a) * ncb + c]);
        }
      }
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const int c = l + 64 * m;
        if (c > ncb) continue;
        double xc[6];
#pragma unroll
        for (int ap = 0; ap < 6; ++ap) xc[ap] = c_live[m] ? xp[ap * ld + c_src[m]] : 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          double v = row[a * ld + c];
#pragma unroll
          for (int ap = 0; ap < 6; ++ap) v = fma(-B[ap][a], xc[ap], v);
          if (c == ncb) {  // right-hand side: row vr of the reversed middle block is row dm - 1 - vr of the natural one
            J.win[size_t(6 * w_mid - 1 - a) * (ncb + 1) + ncb] = v - o0[m][a];
            J.win[size_t(6 * w_mid - 1 - (6 + a)) * (ncb + 1) + ncb] = row1[a * ld + c] - o1[m][a];
          } else {
            hand_over(a, c, v - o0[m][a]);
            hand_over(6 + a, c, row1[a * ld + c] - o1[m][a]);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (every wave: its stores have reached the L2; the ONE agent-scope release — an L2 write-back on this part — is lane 0's below)
      lds_barrier();
      if (l == 0) {
        __threadfence();
        __hip_atomic_store(T.join_flag, T.join_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#undef UIDX
    return;
  }

  // ================================ compute waves: static tile ownership ================================
  // Ring slot s (block rows s, s + bw, s + 2 bw, ...) lives in compute wave s % NCW, lanes (s / NCW) (bw - 2) + kk: consecutive block rows sit in
  // different waves. The wave that owns block row i + 2 is the last one at the step barrier (it publishes the row on top of its update
  // pass); with consecutive rows in one wave it also had to refill the slot it published the step before.
  static_assert(TPT == 1, "one tile per lane");
  int t_kk[TPT], t_row[TPT];
  bool t_ok[TPT], t_refill[TPT];  // t_refill: the slot was published in the previous step and takes its next block row from the stage
  double acc[TPT][36], rhs[TPT][6];
#pragma unroll
  for (int m = 0; m < TPT; ++m) {
    const int slot = cwave + NCW * (l / (bw - 2));
    t_ok[m] = slot < bw;
    t_kk[m] = l % (bw - 2);
    t_row[m] = slot < 2 ? slot + bw : slot;  // rows 0 and 1 start in LDS (loader); their slots prefetch rows bw, bw + 1
    t_refill[m] = false;
  }
#pragma unroll
  for (int m = 0; m < TPT; ++m) {
    const int r = t_row[m];
    const bool in = t_ok[m] && r < n_blk;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double* src = J.Sb + size_t(6 * (in ? r : 0) + a) * ncb + 6 * t_kk[m];
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[m][6 * a + c] = in ? src[c] : 0.0;
      rhs[m][a] = (in && t_kk[m] == 0) ? J.g_s[6 * r + a] : 0.0;
    }
  }
  const bool cprof = prof_enabled(T.debug_flags, 16) && tid == 0;  // coarse phases of this workgroup -> tlog[8 (200 + 10 job) + ..]
  long long* clog = reinterpret_cast<long long*>(T.xpart) + 8 * (200 + 10 * blockIdx.x);
  if (cprof) clog[0] = wall_clock64();  // tiles requested
  wait_vmem();
  if (cprof) clog[1] = wall_clock64();  // tiles loaded
  lds_barrier();  // init
  lds_barrier();  // prologue: X_0 complete
  if (cprof) clog[2] = wall_clock64();
  const bool prof = prof_enabled(T.debug_flags, 16) && tid == 0 && blockIdx.x == 0;
  const bool prof12 = prof_enabled(T.debug_flags, 16) && (tid == 64 || tid == 128) && blockIdx.x == 0;  // compute waves 1, 2 -> tlog[8 (512 + i) + 2 wave ..]
  long long* tlog = reinterpret_cast<long long*>(T.xpart);
  for (int i = 0; i < n_steps; ++i) {
    if (prof) tlog[8 * i + 0] = wall_clock64();
    if (prof12) tlog[8 * (512 + i) + 2 * (tid >> 6)] = wall_clock64();
    const double* xb = xbuf + (i & 1) * 6 * ld;
#pragma unroll
    for (int m = 0; m < TPT; ++m) {
      const int j = t_row[m] - i;
      if (t_refill[m]) {  // published in the previous step: the next block row of this slot (first update three steps from now)
        const int r = t_row[m];
        const double* src = stage + (r % 3) * 6 * ld + 6 * t_kk[m];
        const double first = t_kk[m] == 0 ? 1.0 : 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = 0; c < 6; c += 2) {
            const double2 t = *reinterpret_cast<const double2*>(&src[a * ld + c]);
            acc[m][6 * a + c] = t.x, acc[m][6 * a + c + 1] = t.y;
          }
          rhs[m][a] = first * stage[(r % 3) * 6 * ld + a * ld + ncb];
        }
        t_refill[m] = false;
      }
      if (!t_ok[m] || j < 2 || j > bw - 1) continue;
      if (j + t_kk[m] <= bw - 1 && !(T.debug_flags & 2)) {  // rank-6 update  S_(i+j),kk -= X_j' X_(j+kk)
        const int ca = 6 * j, cb = 6 * (j + t_kk[m]);
        // The right-hand side rides along as a seventh column in EVERY lane (only the lanes with kk = 0 ever use theirs): as a branch
        // around six loads and FMAs per row of X it cost a full drain of the LDS queue per row (s_waitcnt lgkmcnt(0) inside the
        // predicated block), ~0.3 us per step in every compute wave.
        double y[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) y[a] = xb[a * ld + ncb];
        // software pipelined over the six rows of X: the operands of row a + 1 are requested before the 36 FMAs of row a
        double xa[2][6], xc[2][6];
        auto fetch_x = [&](int a, int b) {
#pragma unroll
          for (int c = 0; c < 6; c += 2) {
            const double2 va = *reinterpret_cast<const double2*>(&xb[a * ld + ca + c]);
            const double2 vb = *reinterpret_cast<const double2*>(&xb[a * ld + cb + c]);
            xa[b][c] = va.x, xa[b][c + 1] = va.y, xc[b][c] = vb.x, xc[b][c + 1] = vb.y;
          }
        };
        fetch_x(0, 0);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const int b = a & 1;
          if (a < 5) fetch_x(a + 1, b ^ 1);
#pragma unroll
          for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[m][6 * r + c] = fma(-xa[b][r], xc[b][c], acc[m][6 * r + c]);
            rhs[m][r] = fma(-xa[b][r], y[a], rhs[m][r]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (j == 2) {  // row i + 2 becomes the panel row of the next step: publish; the slot is refilled in the next step
        double* dst = rowbuf + (i & 1) * 6 * ld;
        const int c_rhs = t_kk[m] == 0 ? ncb : ncb + 1;  // lanes without a right-hand side write theirs to the pad column
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = 0; c < 6; c += 2)
            *reinterpret_cast<double2*>(&dst[a * ld + 6 * t_kk[m] + c]) = make_double2(acc[m][6 * a + c], acc[m][6 * a + c + 1]);
          dst[a * ld + c_rhs] = rhs[m][a];
        }
        t_row[m] = i + 2 + bw;
        t_refill[m] = true;
      }
    }
    if (prof) tlog[8 * i + 1] = wall_clock64();
    if (prof12) tlog[8 * (512 + i) + 2 * (tid >> 6) + 1] = wall_clock64();
    lds_barrier();
    if (m_at >= 0 && i + 1 == m_at) {  // ---- junction: add the other end's Schur contribution to the middle rows ----
      if (cprof) clog[3] = wall_clock64();
      junction_wait();
      if (cprof) clog[4] = wall_clock64();
      const int dm = 6 * w_mid, wl = ncb + 1;
      const double* WD = J.win;  // correction in this job's own band layout, local to the middle rows (see hand_over)
      // The window was written by a workgroup on another XCD: every load misses this L2 (~2 us). All operands are requested first —
      // the two LDS rows (up to six entries per lane) and the register tiles — so that the merge costs one round trip, not seven.
      constexpr int RU = 7;  // 2 * 6 * (ncb + 1) <= RU * nthr  (ncb <= 96)
      double wrow[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int e = tid + u * nthr;
        const bool ok = e < 2 * 6 * (ncb + 1);
        const int jr = ok ? e / (6 * (ncb + 1)) : 0, rem = ok ? e % (6 * (ncb + 1)) : 0, a = rem / (ncb + 1), c = rem % (ncb + 1);
        wrow[u] = WD[size_t(6 * jr + a) * wl + c];
      }
      double wt[TPT][36], wh[TPT][6];
#pragma unroll
      for (int m = 0; m < TPT; ++m) {
        const int jr = t_row[m] - m_at;
        const bool ok = t_ok[m] && jr >= 2 && jr < w_mid;
        const double* src = WD + (ok ? size_t(6 * jr) * wl + 6 * t_kk[m] : 0);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = 0; c < 6; ++c) wt[m][6 * a + c] = src[size_t(a) * wl + c];
          wh[m][a] = WD[(ok ? size_t(6 * jr + a) * wl : 0) + ncb];
        }
      }
      // rows m and m + 1 sit in LDS
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int e = tid + u * nthr;
        if (e < 2 * 6 * (ncb + 1)) {
          const int jr = e / (6 * (ncb + 1)), rem = e % (6 * (ncb + 1)), a = rem / (ncb + 1), c = rem % (ncb + 1);
          double* dst = rowbuf + ((m_at + jr) & 1) * 6 * ld + a * ld + c;
          *dst = (c == ncb || 6 * jr + c < dm) ? *dst + wrow[u] : 0.0;  // columns beyond the middle couple to the eliminated end
        }
      }
#pragma unroll
      for (int m = 0; m < TPT; ++m) {
        const int jr = t_row[m] - m_at;
        if (!t_ok[m] || jr < 2 || jr >= w_mid) continue;
        const bool inside = jr + t_kk[m] <= w_mid - 1;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = 0; c < 6; ++c) acc[m][6 * a + c] = inside ? acc[m][6 * a + c] + wt[m][6 * a + c] : 0.0;
          if (t_kk[m] == 0) rhs[m][a] += wh[m][a];
        }
      }
      if (cprof) clog[5] = wall_clock64();
      lds_barrier();  // merge done
      lds_barrier();  // panel(m) done
      if (cprof) clog[6] = wall_clock64();
    }
  }
  lds_barrier();
  if (cprof) clog[7] = wall_clock64();  // last block row done
  if (dump) {  // rows n_steps + 2 .. n_steps + w - 1 of the trailing window live in the register tiles
#pragma unroll
    for (int m = 0; m < TPT; ++m) {
      const int jr = t_row[m] - n_steps;
      if (!t_ok[m] || jr < 2 || jr >= w_mid) continue;
      double so[36], go[6];  // the originals first: the stores below may alias them as far as the compiler knows, which serialised
                             // 36 load -> subtract -> store round trips
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = 0; c < 6; ++c) so[6 * a + c] = J.Sb[size_t(6 * t_row[m] + a) * ncb + 6 * t_kk[m] + c];
        go[a] = t_kk[m] == 0 ? J.g_s[6 * t_row[m] + a] : 0.0;
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = 0; c < 6; ++c) hand_over(6 * jr + a, 6 * t_kk[m] + c, acc[m][6 * a + c] - so[6 * a + c]);
        if (t_kk[m] == 0) J.win[size_t(6 * w_mid - 1 - (6 * jr + a)) * (ncb + 1) + ncb] = rhs[m][a] - go[a];
      }
    }
    __threadfence();
    lds_barrier();
    if (cprof) clog[8] = wall_clock64();  // window handed over
  }
}

/// Backward sweeps of the two-ended factorisation (grid = 2). Block 0: the top system (block rows 0 .. m + w - 1), ordinary sweep,
/// publishes the middle solution (raises the flag once block row m is done). Block 1: the reversed bottom system: its first w
/// block rows in sweep order are the middle rows (given), the others are solved as usual. Both write the solution in natural
/// order to T.xsol; k_step_outputs finishes.
struct BackJob {
  const double* Ub;
  const double* Ubk;
  const double* ybuf;
  const double* Vb;   // block-row-scaled factor diag(U_jj^-1) U (k_band_backward_w)
  const double* yt;   // diag(U_jj^-1) y
  int n_rows;   // block rows of this factor
  int given;    // block rows above them in sweep order whose solution comes from the other job
  int reversed; // solution index = np - 1 - rho
};


/// Factorisation for wide bands (long feature tracks: more tiles than the register-resident kernels can hold): same algorithm and
/// outputs (Ub, U_ii^-1, y). The trailing window stays in HBM / L2 (in place in Sb), one 6x6 tile per lane and step:
///   P1  every lane factors the 6x6 diagonal block of the pivot row redundantly in registers (the row sits in LDS), lane c solves
///       column c of X = U_ii^-T [S_i,: | g_i] -> LDS (for the update) and HBM (the factor row);           --- barrier ---
///   P2  lane (j, kk), 1 <= j < bw, j + kk <= bw - 1: tile (i + j, kk) -= X_j' X_(j+kk) (load from L2, 216 FMAs, store back);
///       the lanes of row i + 1 also publish their tile as the next pivot row in LDS;  stores drained   --- barrier ---
/// One workgroup of 512 lanes, two tiles per lane (bw <= 42: at most 862 tiles; 1024 lanes would leave 128 registers per lane and
/// spill the 6x6 accumulator). LDS (doubles): rowbuf 6 x ld | xbuf 6 x ld.
constexpr int kWideThreads = 512, kWideTiles = 2;

__global__ void __launch_bounds__(kWideThreads) k_band_factor_wide(Tables T) {
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const int tid = threadIdx.x;
  const int bw = T.bw, ncb = 6 * bw, ld = ncb + 2;
  const int n_blk = T.np / 6;
  double* rowbuf = smem;          // 6 x ld : pivot row [band | rhs]
  double* xbuf = smem + 6 * ld;   // 6 x ld : [U_ii | X | y_i]
  __shared__ int fail;
  if (tid == 0) fail = 0;
  // lane -> update tile (j, kk): row i + j, band block kk; row j holds bw - j tiles (kk <= bw - 1 - j), plus for j = 1 the tile
  // kk = bw - 1 that is only copied into the next pivot row
  int tj[kWideTiles], tk[kWideTiles];
  bool t_ok[kWideTiles], t_copy[kWideTiles];
#pragma unroll
  for (int m = 0; m < kWideTiles; ++m) {
    tj[m] = 1, tk[m] = 0, t_ok[m] = false, t_copy[m] = false;
    int rem = tid + m * kWideThreads;
    for (int j = 1; j < bw; ++j) {
      const int cnt = bw - j + (j == 1 ? 1 : 0);
      if (rem < cnt) {
        tj[m] = j, tk[m] = rem, t_ok[m] = true, t_copy[m] = (j == 1 && rem == bw - 1);
        break;
      }
      rem -= cnt;
    }
  }
  for (int e = tid; e < 6 * (ncb + 1); e += kWideThreads) {  // pivot row 0
    const int a = e / (ncb + 1), c = e % (ncb + 1);
    rowbuf[a * ld + c] = c < ncb ? T.Sb[size_t(a) * ncb + c] : T.g_s[a];
  }
  __syncthreads();
#define UIDX(a, c) ((a) * 6 - (a) * ((a)-1) / 2 + ((c) - (a)))
  for (int i = 0; i < n_blk; ++i) {
    // ---- P1 ----
    {
      double U[21], inv[6], dmin = 1.0;
      {
        int pidx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int c = a; c < 6; ++c) U[pidx++] = rowbuf[a * ld + c];
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double d = U[UIDX(a, a)];
#pragma unroll
        for (int k = 0; k < a; ++k) d = fma(-U[UIDX(k, a)], U[UIDX(k, a)], d);
        dmin = a == 0 ? d : fmin(dmin, d);
        const double y = __builtin_amdgcn_rsq(d);
        const double e = fma(-d * y, y, 1.0);
        const double rs = fma(y * e, fma(0.375, e, 0.5), y);
        inv[a] = rs;
#pragma unroll
        for (int c = a + 1; c < 6; ++c) {
          double t = U[UIDX(a, c)];
#pragma unroll
          for (int k = 0; k < a; ++k) t = fma(-U[UIDX(k, a)], U[UIDX(k, c)], t);
          U[UIDX(a, c)] = t * rs;
        }
      }
      if (!(dmin > 0.0) && tid == 0) fail = 1;
      if (tid <= ncb) {  // column tid of [U_ii | X | y]
        const int c = tid;
        double x[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          double t = rowbuf[a * ld + c];
#pragma unroll
          for (int k = 0; k < a; ++k) t = fma(-U[UIDX(k, a)], x[k], t);
          x[a] = t * inv[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          xbuf[a * ld + c] = x[a];
          if (c < ncb)
            T.Ub[size_t(6 * i + a) * ncb + c] = x[a];
          else
            T.ybuf[6 * i + a] = x[a];
        }
      } else if (tid >= kWideThreads - 6) {  // W = U_ii^-1 (upper): lane c solves U w = e_c
        const int c = tid - (kWideThreads - 6);
        double w[6];
#pragma unroll
        for (int a = 5; a >= 0; --a) {
          double t = a == c ? 1.0 : 0.0;
#pragma unroll
          for (int k = a + 1; k < 6; ++k) t = fma(-U[UIDX(a, k)], w[k], t);
          w[a] = t * inv[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
          if (a <= c) T.Ubk[size_t(i) * 24 + UIDX(a, c)] = w[a];
      }
    }
    __syncthreads();
    // ---- P2 ---- (the lane's tiles one after the other: two accumulators at once do not fit 256 registers without spilling)
#pragma unroll 1
    for (int m = 0; m < kWideTiles; ++m) {
      // (register selects: indexing the bookkeeping arrays with the runtime m would put them in scratch)
      const int tjm = m == 0 ? tj[0] : tj[1], tkm = m == 0 ? tk[0] : tk[1];
      const bool okm = m == 0 ? t_ok[0] : t_ok[1], copym = m == 0 ? t_copy[0] : t_copy[1];
      if (!okm || i + tjm >= n_blk) continue;
      double* tile = T.Sb + size_t(6) * (i + tjm) * ncb + 6 * tkm;
      double acc[36], rhs[6];
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; c += 2) {
          const double2 t = *reinterpret_cast<const double2*>(tile + size_t(a) * ncb + c);
          acc[6 * a + c] = t.x, acc[6 * a + c + 1] = t.y;
        }
#pragma unroll
      for (int a = 0; a < 6; ++a) rhs[a] = tkm == 0 ? T.g_s[6 * (i + tjm) + a] : 0.0;
      if (!copym) {
        const int ca = 6 * tjm, cb = 6 * (tjm + tkm);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          double xa[6], xc[6];
#pragma unroll
          for (int c = 0; c < 6; c += 2) {
            const double2 va = *reinterpret_cast<const double2*>(&xbuf[a * ld + ca + c]);
            const double2 vb = *reinterpret_cast<const double2*>(&xbuf[a * ld + cb + c]);
            xa[c] = va.x, xa[c + 1] = va.y, xc[c] = vb.x, xc[c + 1] = vb.y;
          }
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[6 * r + c] = fma(-xa[r], xc[c], acc[6 * r + c]);
          if (tkm == 0) {
            const double y = xbuf[a * ld + ncb];
#pragma unroll
            for (int r = 0; r < 6; ++r) rhs[r] = fma(-xa[r], y, rhs[r]);
          }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int c = 0; c < 6; c += 2) *reinterpret_cast<double2*>(tile + size_t(a) * ncb + c) = make_double2(acc[6 * a + c], acc[6 * a + c + 1]);
        if (tkm == 0)
#pragma unroll
          for (int a = 0; a < 6; ++a) T.g_s[6 * (i + tjm) + a] = rhs[a];
      }
      if (tjm == 1) {  // next pivot row
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = 0; c < 6; c += 2) *reinterpret_cast<double2*>(&rowbuf[a * ld + 6 * tkm + c]) = make_double2(acc[6 * a + c], acc[6 * a + c + 1]);
          if (tkm == 0) rowbuf[a * ld + ncb] = rhs[a];
        }
      }
    }
    __threadfence_block();  // the tiles stored above are read by other lanes in the next step
    __syncthreads();
  }
#undef UIDX
  if (tid == 0 && fail) st->chol_failed = 1;  // (consumed and cleared by decide_step, kernels_update.hpp; an earlier kernel of this iteration may have raised it)
}

/// Backward sweep U x = y (y in T.ybuf, possibly corrected by the border solve) + step outputs and model-cost reductions.
/// The sweep stops above block row j_lo: the leading block rows of constant control points are decoupled with a zero right-hand side
/// (k_factor_decoupled_rows), their part of the solution is zero (half of the block rows of a full sliding window).
__global__ void __launch_bounds__(kCholThreads) k_band_backward(Tables T, int j_lo) {
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const int tid = threadIdx.x;
  constexpr int nthr = kCholThreads;
  const int bw = T.bw, ncb = 6 * bw;
  const int n_blk = T.np / 6;
  double* xs = smem;         // np : pending rows
  double* xout = xs + T.np;  // np : final x
  __shared__ double Wl[2][24];
  for (int rho = tid; rho < T.np; rho += nthr) xs[rho] = T.ybuf[rho], xout[rho] = 0.0;
  if (T.debug_flags & 1) return;  // timing experiments only (HS_DEBUG_FLAGS)

  // ---- backward solve U x = y, column oriented: once x_j is final every pending row above subtracts U[rho][x_j] ----
  // thread t owns pending row rho = 6 j - 1 - t of step j; its six U entries (contiguous in the band row) and U_jj^-1
  // are prefetched three steps ahead. One barrier per block row.
  const int n_above = 6 * (bw - 1);
  auto load_u = [&](int j, double* u) {
    const int rho = 6 * j - 1 - tid;
    const bool ok = j >= 0 && tid < n_above && rho >= 0;
    const double* src = T.Ub + (ok ? size_t(rho) * ncb + (6 * j - 6 * (rho / 6)) : 0);
#pragma unroll
    for (int a = 0; a < 6; ++a) u[a] = ok ? src[a] : 0.0;
  };
  auto load_w = [&](int j) -> double { return (j >= 0 && tid < 21) ? T.Ubk[size_t(j) * 24 + tid] : 0.0; };
  __syncthreads();
  // Four register sets, renamed by unrolling (a rotating set costs ~40 v_mov per block row: 0.08 us of the 0.5 us step). Ablation at
  // configs[2] (128 rows, 69 us): the bare loop (barrier, y, x store) 34 us, W y 15 us, operand loads 14 us, pending-row update 6 us —
  // the step is instruction-issue bound, not a latency chain: neither two block rows per barrier (81 us), nor a deeper prefetch, nor one
  // wave instead of four changed it.
  double ua[6], ub[6], uc[6], ud[6], wa, wb, wc, wd;
  load_u(n_blk - 1, ua), load_u(n_blk - 2, ub), load_u(n_blk - 3, uc);
  wa = load_w(n_blk - 1), wb = load_w(n_blk - 2), wc = load_w(n_blk - 3);
  auto body = [&](int j, const double* u, double w, double* u_next, double* w_next) {
    if (j < j_lo) return;
    if (tid < 21) Wl[j & 1][tid] = w;
    load_u(j - 3, u_next), *w_next = load_w(j - 3);
    lds_barrier();  // publishes Wl and the pending-row updates of the previous step
    const double* W = Wl[j & 1];
    double y[6], x[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) y[a] = xs[6 * j + a];
    {
      int p = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double v = 0.0;
#pragma unroll
        for (int c = a; c < 6; ++c) v = fma(W[p++], y[c], v);
        x[a] = v;
      }
    }
    if (tid < 6) xout[6 * j + tid] = x[tid];
    if (tid < n_above && 6 * j - 1 - tid >= 0) {
      double sacc = 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a) sacc = fma(u[a], x[a], sacc);
      xs[6 * j - 1 - tid] -= sacc;
    }
  };
  for (int j = n_blk - 1; j >= j_lo; j -= 4) {
    body(j, ua, wa, ud, &wd);
    body(j - 1, ub, wb, ua, &wa);
    body(j - 2, uc, wc, ub, &wb);
    body(j - 3, ud, wd, uc, &wc);
  }
  __syncthreads();

  // ---- outputs: step = -x, delta = scale o step, reductions for the model cost change ---------------------------------
  __shared__ double red[kCholThreads / 64];
  double gd = 0.0, dd = 0.0;
  for (int rho = tid; rho < T.np; rho += nthr) {
    const double step = -xout[rho];
    T.step_p[rho] = step;
    T.delta_p[rho] = T.scale_p[rho] * step;
    gd = fma(T.g_full[rho], step, gd);
    dd = fma(T.D2p[rho] * step, step, dd);
  }
  for (int b = tid; b < T.nb; b += nthr) {
    const double step = -T.xb[b];
    T.delta_b[b] = T.scale_b[b] * step;
    gd = fma(T.gb_s[b], step, gd);
    dd = fma(T.D2b[b] * step, step, dd);
  }
  gd = block_sum(gd, red);
  dd = block_sum(dd, red);
  if (tid == 0) {
    st->g_dot_step_pose = gd;
    st->d2_step2_pose = dd;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Factorisation of SHORT systems with window-wide bands (the sliding-window replay: ~33 free block rows, every landmark track as long
// as the window, so the "band" is the whole matrix and nothing slides): every 6x6 tile of the upper block triangle inside the band
// lives in the registers of ONE lane for the whole factorisation — no trailing window in L2 (k_band_factor_wide: two global round
// trips per step, 6.4 us per block row), no loader, no tile ownership that moves. Seven waves x 2 tiles own the off-diagonal band tiles and
// the right-hand side (an extra tile column whose columns 1 .. 5 are zero), the eighth wave owns the diagonal tiles. Per block row k:
//   A  the owners of row k (one contiguous run of <= bw lanes) solve their tile in place, X(k, j) = U_kk^-T S(k, j) (U_kk read from LDS
//      with uniform addresses) and write it to LDS, the operand of the update                                          --- barrier ---
//   B  band lanes: tile (i, j) -= X(k, i)' X(k, j) for their tiles with k < i <= k + bw - 1 (216 FMAs, 36 ds_read_b128; ONE code path for
//      band and right-hand-side tiles). Diagonal wave: S(i, i) -= X(k, i)' X(k, i), upper triangle (18 reads, 126 FMAs); the owner of
//      (k + 1, k + 1) then factors it in place (6x6 Cholesky, one lane) and publishes U, 1 / diag, the diagonal tile of X_(k+1) — in the
//      shadow of the band lanes' updates (X rows double buffered, U_kk per block row)                                   --- barrier ---
// Every tile stays in the registers of its owner to the end: the factor (band storage), y and U_kk^-1 are written after the last step.
// Chain per block row (tools/dense_phase_timing.py): A 0.64 us, diagonal update 0.56 us, pivot 0.68 us, barriers: 2.1 us — 3.5 us while
// lanes of one wave owned tiles of three kinds (the update ran once per kind) and the stores and U_kk^-1 were inside the steps.
// (A redundant register Cholesky in every row owner, as in the panel of k_band_factor_la, needs 60 more live registers next to the two
//  resident tiles and spilled: 6.5 us per block row, parked 83 % of the time. A right-looking pivot with U_kk read from the X row: pivot
//  0.56 us, A 0.88 us — no gain.)
// Same outputs as the other factorisation kernels (Ub in band storage, U_ii^-1 packed, y = U^-T g).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kDenseThreads = 512, kDenseTiles = 2;
constexpr int kDenseBand = kDenseThreads - 64;  // lanes that own band / right-hand-side tiles; the last wave owns the diagonal tiles

HSD void factor_decoupled_row(const Tables& T, int i, int lane);

/// Number of tile slots the dense kernel needs in its band lanes for n block rows of band width bw (off-diagonal band tiles + one
/// right-hand-side tile per row; the n diagonal tiles live in the last wave).
__host__ __device__ constexpr int dense_factor_tiles(int n, int bw) {
  int t = 0;
  for (int i = 0; i < n; ++i) t += (n - i < bw ? n - i : bw);
  return t;
}
__host__ __device__ constexpr bool dense_factor_fits(int n, int bw) {
  return n <= 64 * kDenseTiles && dense_factor_tiles(n, bw) <= kDenseBand * kDenseTiles;
}

/// Workgroups 1 .. n_decoupled (first wave only) write the decoupled leading block rows -n_decoupled .. -1 (factor_decoupled_row): their
/// own launch in front of this kernel cost 6 us on the chain.
__global__ void __launch_bounds__(kDenseThreads) k_dense_factor(Tables T, int n_decoupled) {
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {
    if (tid < 64) factor_decoupled_row(T, int(blockIdx.x) - 1 - n_decoupled, tid);
    return;
  }
  const int bw = T.bw, ncb = 6 * bw, n = T.np / 6;
  const int ldx = ncb + 8;           // row stride of the X row in LDS: [6 x (bw tiles) | y]
  double* xrow = smem;               // 2 x 6 x ldx : X_k in band order (column 6 (j - k) + c), right-hand side at column ncb
  double* dscr = smem + 12 * ldx;    // n x 32 : U_kk (upper, packed, 21) and 1 / diag (6), published by the owner of the diagonal tile
  __shared__ int fail;
  if (tid == 0) fail = 0;
  // HS_DEBUG_FLAGS 16: phase timestamps (100 MHz clock) -> hs_debug_read, tools/dense_phase_timing.py. Per block row k (8 slots):
  // [0] barrier after A, [1] barrier after B (thread 0); [2] row owner (k, k + 1): solve done, [3] stores issued; diagonal owner
  // (k + 1, k + 1): [4] update done, [5] factorised and published
  const bool prof = prof_enabled(T.debug_flags, 16);
  long long* tlog = reinterpret_cast<long long*>(T.xpart) + 8 * 300;
  if (prof && tid == 0) tlog[-1] = wall_clock64();
  // ---- static tile ownership: slot t = tid + 512 m -> (i, j), row major over rows i with columns j = i .. min(i + bw, n) - 1 and
  //      the right-hand-side column (encoded as j = n) ----
  int ti[kDenseTiles], tj[kDenseTiles];
  double acc[kDenseTiles][36];
#pragma unroll
  for (int m = 0; m < kDenseTiles; ++m) {
    ti[m] = -1, tj[m] = -1;
    if (tid >= kDenseBand) {  // diagonal tiles: one wave of their own (its code path is the next pivot's: half the operands, no other kind)
      const int i = tid - kDenseBand + 64 * m;
      if (i < n) ti[m] = tj[m] = i;
    } else {
      int rem = tid + m * kDenseBand;
      for (int i = 0; i < n; ++i) {
        const int cnt = n - i < bw ? n - i : bw;  // columns i + 1 .. i + cnt - 1 and the right-hand side
        if (rem < cnt) {
          ti[m] = i, tj[m] = rem == cnt - 1 ? n : i + 1 + rem;
          break;
        }
        rem -= cnt;
      }
    }
#pragma unroll
    for (int e = 0; e < 36; ++e) acc[m][e] = 0.0;
    if (ti[m] >= 0) {
      if (tj[m] == n) {
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[m][6 * a] = T.g_s[6 * ti[m] + a];  // right-hand side: column 0 of the tile
      } else {
        const double* src = T.Sb + size_t(6 * ti[m]) * ncb + 6 * (tj[m] - ti[m]);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int c = 0; c < 6; ++c) acc[m][6 * a + c] = (tj[m] > ti[m] || c >= a) ? src[size_t(a) * ncb + c] : 0.0;
      }
    }
  }
  for (int e = tid; e < 12 * ldx; e += kDenseThreads) xrow[e] = 0.0;  // (the pad columns behind y stay zero: a right-hand-side tile is
                                                                      //  an ordinary tile whose columns 1 .. 5 are zero)
  __syncthreads();
#define UIDX(a, c) ((a) * 6 - (a) * ((a)-1) / 2 + ((c) - (a)))
  // X rows and the published U_kk are double buffered by k & 1: the owner of (k + 1, k + 1) factors its tile at the end of its part of
  // update k (nothing else depends on that lane), while the other lanes still read X_k
  auto factor_diagonal = [&](double (&t)[36], int k) {  // in place; publishes U_kk, 1 / diag and the diagonal tile of X_k
    double* ub = dscr + 32 * k;
    double* xr = xrow + (k & 1) * 6 * ldx;
    double dmin = 1.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double d = t[6 * a + a];
#pragma unroll
      for (int q = 0; q < a; ++q) d = fma(-t[6 * q + a], t[6 * q + a], d);
      dmin = a == 0 ? d : fmin(dmin, d);
      const double y = __builtin_amdgcn_rsq(d);
      const double e = fma(-d * y, y, 1.0);
      const double rs = fma(y * e, fma(0.375, e, 0.5), y);
      ub[21 + a] = rs;
      t[6 * a + a] = d * rs;  // u_aa
#pragma unroll
      for (int c = a + 1; c < 6; ++c) {
        double v = t[6 * a + c];
#pragma unroll
        for (int q = 0; q < a; ++q) v = fma(-t[6 * q + a], t[6 * q + c], v);
        t[6 * a + c] = v * rs;
      }
    }
    if (!(dmin > 0.0)) fail = 1;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (c < a) t[6 * a + c] = 0.0;
        if (c >= a) ub[UIDX(a, c)] = t[6 * a + c];
      }
#pragma unroll
      for (int c = 0; c < 6; c += 2) {  // (rows are 16-byte aligned: ldx and ncb are even)
        *reinterpret_cast<double2*>(xr + a * ldx + c) = make_double2(t[6 * a + c], t[6 * a + c + 1]);
      }
    }
  };
#pragma unroll
  for (int m = 0; m < kDenseTiles; ++m)
    if (ti[m] == 0 && tj[m] == 0) factor_diagonal(acc[m], 0);
  lds_barrier();
  for (int k = 0; k < n; ++k) {
    const double* ubuf = dscr + 32 * k;
    double* xk = xrow + (k & 1) * 6 * ldx;
    // ---- A: the owners of row k other than the diagonal one solve their tile in place, X(k, j) = U_kk^-T S(k, j); U_kk is read from
    //      LDS (uniform addresses: broadcast), so the only registers involved are the tile's own ----
#pragma unroll
    for (int m = 0; m < kDenseTiles; ++m) {
      if (ti[m] != k || tj[m] == k) continue;
      const int col0 = tj[m] == n ? ncb : 6 * (tj[m] - k);  // (right-hand side: column ncb, its zero columns land in the pad)
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const double ia = ubuf[21 + a];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double t = acc[m][6 * a + c];
#pragma unroll
          for (int q = 0; q < a; ++q) t = fma(-ubuf[UIDX(q, a)], acc[m][6 * q + c], t);
          acc[m][6 * a + c] = t * ia;
        }
      }
      if (prof && tj[m] == k + 1) tlog[8 * k + 2] = wall_clock64();
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; c += 2)
          *reinterpret_cast<double2*>(xk + a * ldx + col0 + c) = make_double2(acc[m][6 * a + c], acc[m][6 * a + c + 1]);
      if (prof && tj[m] == k + 1) tlog[8 * k + 3] = wall_clock64();
    }
    lds_barrier();
    if (prof && tid == 0) tlog[8 * k] = wall_clock64();
    // ---- B: trailing update ----
    if (tid >= kDenseBand) {
      // diagonal tiles: S(i, i) -= X(k, i)' X(k, i), upper triangle (18 operand loads, 126 FMAs); the owner of the next pivot factors it
#pragma unroll
      for (int m = 0; m < kDenseTiles; ++m) {
        const int i = ti[m];
        if (i <= k || i - k >= bw) continue;
        const double* A = xk + 6 * (i - k);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          double av[6];
#pragma unroll
          for (int a = 0; a < 6; a += 2) {
            const double2 ta = *reinterpret_cast<const double2*>(A + q * ldx + a);
            av[a] = ta.x, av[a + 1] = ta.y;
          }
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = a; c < 6; ++c) acc[m][6 * a + c] = fma(-av[a], av[c], acc[m][6 * a + c]);
        }
        if (i == k + 1) {  // (the next pivot: its tile is final now)
          if (prof) tlog[8 * k + 4] = wall_clock64();
          factor_diagonal(acc[m], k + 1);
          if (prof) tlog[8 * k + 5] = wall_clock64();
        }
      }
    } else {  // band columns past the end of the matrix: zero in the factor row (the sweeps read whole rows); the band lanes share the stores
      const int used = 6 * (n - k < bw ? n - k : bw), nz = ncb - used;
      for (int e = tid; e < 6 * nz; e += kDenseBand) T.Ub[size_t(6 * k + e / nz) * ncb + used + e % nz] = 0.0;
    }
#pragma unroll
    for (int m = 0; m < kDenseTiles; ++m) {
      const int i = ti[m], j = tj[m];
      if (tid >= kDenseBand || i <= k || i - k >= bw) continue;  // diagonal wave; finished rows; rows X_k does not reach
      if (j != n && j - k >= bw) continue;      // (inside the band of row i but beyond the band of row k: untouched by X_k)
      const double* A = xk + 6 * (i - k);
      const double* B = xk + (j == n ? ncb : 6 * (j - k));
      // (one code path for band, diagonal and right-hand-side tiles: lanes of a wave that own different kinds would otherwise run the
      //  216 FMAs once per kind — measured 1.76 us for this loop with three kinds, tools/dense_phase_timing.py)
#pragma unroll 2  // (fully unrolled, the scheduler hoists all 36 operand loads above the FMAs: 144 more live registers -> scratch)
      for (int q = 0; q < 6; ++q) {
        double av[6], bv[6];
#pragma unroll
        for (int a = 0; a < 6; a += 2) {
          const double2 ta = *reinterpret_cast<const double2*>(A + q * ldx + a);
          const double2 tb = *reinterpret_cast<const double2*>(B + q * ldx + a);
          av[a] = ta.x, av[a + 1] = ta.y, bv[a] = tb.x, bv[a + 1] = tb.y;
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int c = 0; c < 6; ++c) acc[m][6 * a + c] = fma(-av[a], bv[c], acc[m][6 * a + c]);
      }
    }
    lds_barrier();
    if (prof && tid == 0) tlog[8 * k + 1] = wall_clock64();
  }
  // ---- the factor: every tile is still in the registers of its owner (X(i, j), U_ii with a zero lower triangle, y_i in column 0 of the
  //      right-hand-side tiles). Stored here, off the chain: inside the steps the 18 + 18 stores of a row owner took 0.8 us ----
#pragma unroll
  for (int m = 0; m < kDenseTiles; ++m) {
    const int i = ti[m], j = tj[m];
    if (i < 0) continue;
    if (j == n) {
#pragma unroll
      for (int a = 0; a < 6; ++a) T.ybuf[6 * i + a] = acc[m][6 * a];
    } else {
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; c += 2)
          *reinterpret_cast<double2*>(T.Ub + size_t(6 * i + a) * ncb + 6 * (j - i) + c) = make_double2(acc[m][6 * a + c], acc[m][6 * a + c + 1]);
    }
  }
  // ---- U_kk^-1 (upper, packed; operand of the backward sweep) for every block row at once, one lane each: on the chain it cost
  //      ~0.3 us per block row in the lane that owns the next pivot ----
  for (int k = tid; k < n; k += kDenseThreads) {
    const double* ub = dscr + 32 * k;
    double u[21];
#pragma unroll
    for (int e = 0; e < 21; ++e) u[e] = ub[e];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double w[6];
#pragma unroll
      for (int a = 5; a >= 0; --a) {
        double v = a == c ? 1.0 : 0.0;
#pragma unroll
        for (int q = a + 1; q < 6; ++q) v = fma(-u[UIDX(a, q)], w[q], v);
        w[a] = a <= c ? v * ub[21 + a] : 0.0;
      }
#pragma unroll
      for (int a = 0; a < 6; ++a)
        if (a <= c) T.Ubk[size_t(k) * 24 + UIDX(a, c)] = w[a];
    }
  }
#undef UIDX
  if (tid == 0 && fail) st->chol_failed = 1;
}

/// Leading block rows of CONSTANT control points (the sliding window freezes every control point at or before its lower bound,
/// optimizer.cpp:319-328, and keeps them while residuals still reach them): their Jacobian columns are zero, so the block rows are
/// decoupled from everything — S_i,: = [D_i | 0] with the damping on the diagonal, g_i = 0. The factorisation kernels start behind
/// them (pointer offsets at launch, the band storage is row relative); this kernel writes their part of the factor, one wave per block
/// row: U_ii = chol(S_ii), the rest of the row zero, U_ii^-1, y_i = U_ii^-T g_i.
/// (one wave: the lanes clear block row i, lane 0 factors the 6 x 6 block. `i` may be negative: k_dense_factor runs on pointers that
///  were moved past the decoupled rows and reaches back)
HSD void factor_decoupled_row(const Tables& T, const int i, const int lane) {
  const ptrdiff_t ncb = 6 * T.bw;
  for (int e = lane; e < 6 * ncb; e += 64)
    if (e % ncb >= 6) T.Ub[ptrdiff_t(6 * i) * ncb + e] = 0.0;
  if (lane != 0) return;
  double U[6][6], y[6];
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 6; ++c) U[a][c] = c >= a ? T.Sb[ptrdiff_t(6 * i + a) * ncb + c] : 0.0;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double d = U[a][a];
#pragma unroll
    for (int k = 0; k < a; ++k) d = fma(-U[k][a], U[k][a], d);
    ok = ok && d > 0.0;
    const double u = sqrt(d), r = 1.0 / u;
    U[a][a] = u;
#pragma unroll
    for (int c = a + 1; c < 6; ++c) {
      double t = U[a][c];
#pragma unroll
      for (int k = 0; k < a; ++k) t = fma(-U[k][a], U[k][c], t);
      U[a][c] = t * r;
    }
    double t = T.g_s[6 * i + a];
#pragma unroll
    for (int k = 0; k < a; ++k) t = fma(-U[k][a], y[k], t);
    y[a] = t * r;
    T.ybuf[6 * i + a] = y[a];
  }
  if (!ok) T.st->chol_failed = 1;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 6; ++c) T.Ub[ptrdiff_t(6 * i + a) * ncb + c] = U[a][c];  // (zero below the diagonal)
#pragma unroll
  for (int c = 0; c < 6; ++c) {  // W = U_ii^-1, upper, packed like the factorisation kernels do
    double w[6];
#pragma unroll
    for (int a = 5; a >= 0; --a) {
      double t = a == c ? 1.0 : 0.0;
#pragma unroll
      for (int k = a + 1; k < 6; ++k) t = fma(-U[a][k], w[k], t);
      w[a] = a <= c ? t / U[a][a] : 0.0;
    }
#pragma unroll
    for (int a = 0; a <= c; ++a) T.Ubk[ptrdiff_t(i) * 24 + (a * 6 - a * (a - 1) / 2 + (c - a))] = w[a];
  }
}

__global__ void __launch_bounds__(64) k_factor_decoupled_rows(Tables T, int n_rows) {
  if (T.st->done) return;
  if (int(blockIdx.x) < n_rows) factor_decoupled_row(T, blockIdx.x, threadIdx.x);
}

}  // namespace hs
