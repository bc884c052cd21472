// kernels_backward_sb.hpp — backward sweep U x = y of the two-ended factorisation in SUPER-BLOCKS of four block rows (part of kernels.hpp).
//
// The sweep is a dependency chain over the rows of the factor. One block row per step (k_band_backward2) costs 0.55 us per step —
// barrier, operands through LDS, a 6 x 6 triangular product, one update pass — whatever the arithmetic, and a window of 128 control
// points has 57 such steps on the chain of each end. Here a step solves 24 scalar rows at once:
//     x_J = Winv_J y_J,            Winv_J = (U_JJ)^-1, the inverse of the 24 x 24 upper-triangular diagonal super-block,
//     y_rho -= U[rho, J] x_J       for the 6 (bw - 1) pending rows above,
// so the chain has a quarter of the steps, and each step is two short dot products per lane (12 multiply-adds + one cross-lane add)
// with two barriers. The inverses do not depend on one another: extra workgroups of the same launch (one per super-block, one wave
// each) compute them concurrently while the sweep workgroups stage their operands, and publish them through agent-scope flags; the
// sweeps consume them from the bottom row up, four steps after requesting them.
//
// Launch: grid = 2 + n_sb(job 0) + n_sb(job 1), 256 lanes. Workgroups 0, 1 are the two sweeps (roles and hand-over exactly as in
// k_band_backward2: block 0 solves the top system and publishes the middle solution, block 1 takes it as given and solves the reversed
// bottom system); the remaining workgroups are the inverse builders (first wave only).
// One-ended systems (bordered: bias splines + gravity; n_jobs = 1): grid = 1 + n_sb, workgroup 0 sweeps the whole factor from the last
// block row up to block row j_lo (the block rows of leading constant control points above it are decoupled with a zero right-hand
// side) and writes the step outputs itself, including those of the border unknowns, like k_band_backward.
#pragma once
#include "kernels_factor.hpp"

namespace hs {

constexpr int kSb = 4;            // block rows per super-block
constexpr int kSbN = 6 * kSb;     // scalar rows per super-block
constexpr int kSbPrefetch = 5;    // super-steps between requesting the operands of a step from HBM / MALL and using them (~2 us)
constexpr int kSbFlagBase = 4;    // T.join_flag[kSbFlagBase + 512 job + s] = epoch once Winv of super-block s of that job is in memory
constexpr int kSbOut = 2;        // step-output operands a lane keeps from the start of the kernel (2 x 256 scalar rows per end: 170 control points per window)
constexpr int kSbMaxBlocks = 512; // super-blocks per job the flag table has room for (n_cp <= 1024 control points)

HSD int sb_count(int n_rows) { return (n_rows + kSb - 1) / kSb; }
/// LDS of the far sweep's phase A (the near factor's top super-blocks solved again, k_band_backward_sb): pending rows + solution of at most
/// kSbPrefetch super-blocks and the 6 (bw - 1) rows above them, + the zero tail a partial last super-block reads, + alignment.
__host__ __device__ inline size_t sb_phase_a_doubles(int bw) { return size_t(2) * (kSbN * kSbPrefetch + 6 * (bw - 1)) + kSbN + 8; }

/// Entry (rho, col) of a factor in band storage (row rho holds columns 6 floor(rho / 6) .. + 6 bw - 1), zero outside the band / matrix.
HSD double band_entry(const double* __restrict__ Ub, int ncb, int n_own, int rho, int col) {
  const int off = col - 6 * (rho / 6);
  return (rho >= 0 && rho < n_own && col < n_own && off >= 0 && off < ncb) ? Ub[size_t(rho) * ncb + off] : 0.0;
}

/// Sum of a value over a lane pair (lanes 2 i, 2 i + 1) through the DPP cross bar (quad_perm [1 0 3 2]): two moves and an add instead
/// of the ~150 cycles of an LDS permute on the chain of the sweep.
HSD double pair_sum(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, false);
  return v + __hiloint2double(hi, lo);
}

/// Winv of super-block s of job J -> J.Vb + 576 s (row-major 24 x 24, zeros below the diagonal and in the rows / columns of a partial
/// last super-block), then the flag. One wave. The 6 x 6 inverses W_j = U_jj^-1 of the diagonal blocks come from the factorisation
/// (J.Ubk); the blocks above the diagonal follow by block back substitution, one block diagonal per level:
///     V_jj = W_j,     V_ij = -W_i sum_(k = i + 1 .. j) U_ik V_kj     (i < j, level j - i),
/// so the dependent chain is three levels of two 6 x 6 products instead of 24 scalar rows.
/// (raise_flag = false: the wave belongs to a workgroup that uses the inverse itself — the fused factor + sweep kernel — and synchronises on its own)
HSD void sb_inverse(const Tables& T, const BackJob& J, int job, int s, double* lds /* 3 x 24 x 25, this wave's */, bool raise_flag = true) {
  const int l = threadIdx.x & 63;
  constexpr int LD = kSbN + 1;
  // one wave: LDS operations complete in program order; only the compiler and the counters must keep it
#define HS_WAVE_SYNC() wait_lds()
  double* U = lds;
  double* V = lds + kSbN * LD;
  double* Tm = lds + 2 * kSbN * LD;
  const int ncb = 6 * T.bw, n_own = 6 * J.n_rows, r0 = kSbN * s;
  const int nr = min(kSbN, n_own - r0);
  const bool iprof = prof_enabled(T.debug_flags, 16) && l == 0 && job == 0 && s == sb_count(J.n_rows) - 1;  // -> xpart[8 * 260 ..]
  long long* ilog = reinterpret_cast<long long*>(T.xpart) + 8 * 260;
  if (iprof) ilog[0] = wall_clock64();
  {  // every load first (nine + five per lane), then the stores: a load -> store loop pays the memory latency once per trip
    constexpr int NU = (kSbN * kSbN + 63) / 64;
    double v[NU], w[2];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = l + 64 * u, a = e / kSbN, c = e % kSbN;
      v[u] = (e < kSbN * kSbN && a < nr && c < nr && c >= a) ? band_entry(J.Ub, ncb, n_own, r0 + a, r0 + c) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {  // W_j: 21 packed entries per block row (upper triangle, row-major)
      const int e = l + 64 * u, jb = e / 21;
      w[u] = (e < kSb * 21 && 6 * jb < nr) ? J.Ubk[size_t(kSb * s + jb) * 24 + e % 21] : 0.0;
    }
    for (int e = l; e < 2 * kSbN * LD; e += 64) lds[e] = 0.0;
    wait_lds();  // one wave: LDS is in order; only the compiler must keep the order
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = l + 64 * u;
      if (e < kSbN * kSbN) U[(e / kSbN) * LD + e % kSbN] = v[u];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = l + 64 * u, jb = e / 21, t = e % 21;
      if (e < kSb * 21) {
        int a = 0, rem = t;  // packed index -> (a, c): row a holds 6 - a entries
        while (rem >= 6 - a) rem -= 6 - a, ++a;
        V[(6 * jb + a) * LD + 6 * jb + a + rem] = w[u];
      }
    }
  }
  HS_WAVE_SYNC();
  if (iprof) ilog[1] = wall_clock64();  // block in LDS
  for (int d = 1; d < kSb; ++d) {
    const int n_e = (kSb - d) * 36;
    for (int e = l; e < n_e; e += 64) {  // Tm_ij = sum_k U_ik V_kj
      const int i = e / 36, r = (e % 36) / 6, c = e % 6, j = i + d;
      double t = 0.0;
      for (int k = i + 1; k <= j; ++k)
#pragma unroll
        for (int m = 0; m < 6; ++m) t = fma(U[(6 * i + r) * LD + 6 * k + m], V[(6 * k + m) * LD + 6 * j + c], t);
      Tm[(6 * i + r) * LD + 6 * j + c] = t;
    }
    HS_WAVE_SYNC();
    for (int e = l; e < n_e; e += 64) {  // V_ij = -W_i Tm_ij
      const int i = e / 36, r = (e % 36) / 6, c = e % 6, j = i + d;
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) v = fma(V[(6 * i + r) * LD + 6 * i + m], Tm[(6 * i + m) * LD + 6 * j + c], v);  // (W_i is upper triangular: zeros below)
      V[(6 * i + r) * LD + 6 * j + c] = -v;
    }
    HS_WAVE_SYNC();
  }
  if (iprof) ilog[2] = wall_clock64();  // inverse in LDS
  double* dst = const_cast<double*>(J.Vb) + size_t(s) * (kSbN * kSbN);
  for (int e = l; e < kSbN * kSbN; e += 64) {
    const int a = e / kSbN, c = e % kSbN;
    dst[e] = (a < nr && c < nr) ? V[a * LD + c] : 0.0;
  }
  if (!raise_flag) return;
  // (the release store waits for every store of the wave — one instruction stream, one counter — and writes the L2 back: no fence of its own in front)
  if (l == 0) {
    __hip_atomic_store(T.join_flag + kSbFlagBase + kSbMaxBlocks * job + s, T.join_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (iprof) ilog[3] = wall_clock64();  // flag raised
  }
#undef HS_WAVE_SYNC
}

/// Bounded wait for a flag word (see wait_for_partner): 2 s, then the factorisation is marked as failed and the caller carries on.
HSD void sb_wait(const Tables& T, const unsigned* flag) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < T.join_epoch) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > 200000000ll) {
      give_up(T.st);
      break;
    }
  }
}

/// The sweep of one job (the first kCholThreads lanes of the workgroup; the others must have left). prebuilt: the inverses of the super-blocks
/// are in memory (fused factor + sweep kernel: built by the same workgroup) — no builder flags to wait for; near_ready: job 1 of the fused
/// kernel waits on it before it touches the near job's factor (phase A).
HSD void sb_sweep(const Tables& T, const BackJob& j0, const BackJob& j1, const int m_mid, const int n_jobs, const int j_lo, const int job, double* smem,
                  const bool prebuilt, const unsigned* near_ready) {
  DevState* st = T.st;
  const int tid = threadIdx.x;
  const BackJob J = job == 0 ? j0 : j1;
  constexpr int nthr = kCholThreads;
  const int bw = T.bw, ncb = 6 * bw, np = T.np;
  const bool cprof = prof_enabled(T.debug_flags, 16) && tid == 0;  // coarse phases -> xpart[8 (230 + 10 block) + ..] (tools/chol_phase_timing.py)
  long long* clog = reinterpret_cast<long long*>(T.xpart) + 8 * (230 + 10 * job);
  if (cprof) clog[0] = wall_clock64();
  const int n_own = 6 * J.n_rows, n_all = 6 * (J.n_rows + J.given);
  double* xs = smem;              // n_all : pending rows (own) / given solution
  double* xout = smem + n_all;    // n_own : solution of the own rows
  double* xj = smem + 2 * np;     // 24 (+ pad 8) : solution of the super-block of the current step
  double* G = smem + 2 * np + 32; // given-column block of the far sweep, see k_band_backward2
  const int n_above = 6 * (bw - 1);
  const bool merged = J.given > 0;  // (the two-ended launch always has 6 given = n_above <= n_own)
  const int ldg = n_above | 1;
  for (int rho = tid; rho < n_own; rho += nthr) xs[rho] = J.ybuf[rho];
  // Two-ended: the far sweep needs the solution of the middle rows, which are the top rows of the NEAR factor. It does not wait for the near
  // sweep to publish them (flag + release / acquire + a round of loads: ~3 us on its chain) — it solves the near factor's top super-blocks
  // itself first, the same steps on the same operands as block 0 (phase A: at most kSbPrefetch super-steps), in arrays of its own.
  const int nA_own = 6 * j0.n_rows, sA_top = sb_count(j0.n_rows) - 1, sA_pub = m_mid >= 0 ? m_mid / kSb : 0;
  const bool redo_mid = n_jobs == 2 && m_mid >= 0 && sA_top - sA_pub + 1 <= kSbPrefetch;
  const bool phase_a = redo_mid && job == 1;
  const int baseA = max(0, kSbN * sA_pub - n_above);  // first row phase A touches
  double* xsA = G + n_above * ldg + 2;                // nA_own - baseA (+ 24 zeros: partial last super-block), indexed from baseA
  double* xoutA = xsA + (nA_own - baseA) + kSbN;      // nA_own - baseA
  // operands of the step outputs at the end of a two-ended sweep (the rows this block solves), requested now
  double o_sc[kSbOut], o_gf[kSbOut], o_d2[kSbOut];
#pragma unroll
  for (int u = 0; u < kSbOut; ++u) {
    const int rho = tid + u * nthr;
    const bool ok = n_jobs == 2 && rho < n_own;
    const int nat = ok ? (J.reversed ? np - 1 - rho : rho) : 0;
    o_sc[u] = ok ? T.scale_p[nat] : 0.0, o_gf[u] = ok ? T.g_full[nat] : 0.0, o_d2[u] = ok ? T.D2p[nat] : 0.0;
  }
  if (n_jobs == 1)
    for (int rho = tid; rho < n_own; rho += nthr) xout[rho] = 0.0;  // (rows above j_lo are not swept)
  if (tid < kSbN && J.given == 0) smem[n_all + tid] = 0.0;  // a partial last super-block reads 24 entries from its first row on
  if (merged) {
    const int n_g = n_above * n_above;
    constexpr int GU = 16;  // (loads in flight per lane: the far sweep's staging is on the chain since it no longer waits for the near one)
    for (int e0 = tid; e0 < n_g; e0 += GU * nthr) {
      double v[GU];
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int e = e0 + u * nthr, r = e / n_above, c = e - r * n_above;
        const int rho = n_own - n_above + r, off = n_own + c - 6 * (rho / 6);  // band offset of column n_own + c in row rho
        v[u] = (e < n_g && off < ncb) ? J.Ub[size_t(rho) * ncb + off] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int e = e0 + u * nthr, r = e / n_above, c = e - r * n_above;
        if (e < n_g) G[c * ldg + r] = v[u];
      }
    }
  }
  if (phase_a && near_ready) {  // fused kernel: wait until the near workgroup has published its factor and its top inverses
    sb_wait(T, near_ready);
    __threadfence();
  }
  if (phase_a) {
    for (int rho = baseA + tid; rho < nA_own; rho += nthr) xsA[rho - baseA] = j0.ybuf[rho];
    if (tid < kSbN) xsA[nA_own - baseA + tid] = 0.0;
  }
  // ---- lane roles ----
  // waves 0 .. 2 (192 lanes), lane (p, q) = (tid / 2, tid & 1): pending row rho = r0 - 1 - p of the step (p < n_above), columns
  //   r0 + 12 q .. + 11 of the super-block: 12 multiply-adds, the two halves are added across the lane pair;
  // wave 3, lane (r, q) = (l / 2, l & 1), r < 24: row r of Winv_J, columns 12 q .. + 11.
  const int wave = tid >> 6, l = tid & 63;
  const bool solver = wave == 3;
  const int q = tid & 1;
  const int p_row = solver ? (l >> 1) : (tid >> 1);
  const int n_sb = sb_count(J.n_rows);
  const unsigned* flags = T.join_flag + kSbFlagBase + kSbMaxBlocks * job;
  double ring[kSbPrefetch][12];  // operands of the next kSbPrefetch steps, oldest first (register renaming by full unrolling below)
  auto request_of = [&](const double* Ub_, const double* Vb_, int n_own_, int s, double* dst) {
    if (s < 0) {
#pragma unroll
      for (int c = 0; c < 12; ++c) dst[c] = 0.0;
      return;
    }
    const int r0 = kSbN * s;
    if (solver) {
      const bool ok = p_row < kSbN;
      // (16-byte loads: rows of Winv are 192 bytes, rows of the band 8 * 6 bw bytes and every offset below is even)
      const double2* src = reinterpret_cast<const double2*>(Vb_ + size_t(s) * (kSbN * kSbN) + (ok ? p_row : 0) * kSbN + 12 * q);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double2 t = ok ? src[c] : make_double2(0.0, 0.0);
        dst[2 * c] = t.x, dst[2 * c + 1] = t.y;
      }
    } else {
      const int rho = r0 - 1 - p_row;
      const bool ok = p_row < n_above && rho >= 0;
      const int base = r0 + 12 * q - 6 * ((ok ? rho : 0) / 6);  // band offset of the first of the 12 columns (even)
      const double* src = Ub_ + size_t(ok ? rho : 0) * ncb;
#pragma unroll
      for (int c = 0; c < 12; c += 2) {
        const int off = base + c;
        const double2 t = (ok && off < ncb && r0 + 12 * q + c < n_own_) ? *reinterpret_cast<const double2*>(src + off) : make_double2(0.0, 0.0);
        dst[c] = t.x, dst[c + 1] = t.y;
      }
    }
  };
  auto request = [&](int s, double* dst) { request_of(J.Ub, J.Vb, n_own, s, dst); };
  // The solver wave needs the inverses in memory before it requests them. All builders run concurrently and finish within a few
  // microseconds of the launch, so the wave waits for ALL of its job's flags once, lane i polling flag i (an agent-scope acquire load
  // costs ~1 us: one per step on the chain doubled the step, five in a row delayed the first step by 8 us).
  auto request_checked = [&](int s, double* dst) { request(s, dst); };
  if (solver && !prebuilt) {
    for (int i = l; i < n_sb; i += 64) sb_wait(T, flags + i);
    if (phase_a && l <= sA_top - sA_pub) sb_wait(T, T.join_flag + kSbFlagBase + sA_pub + l);  // the near job's top super-blocks
    // (every poll is an acquire load of the wave: the caches it must not read stale lines from were invalidated by the last of them)
  }
  const int s_top = n_sb - 1;
  double ringA[kSbPrefetch][12];  // phase A: operands of the near factor's top super-blocks, all requested at once
  if (phase_a) {
#pragma unroll
    for (int d = 0; d < kSbPrefetch; ++d) request_of(j0.Ub, j0.Vb, nA_own, sA_top - d >= sA_pub ? sA_top - d : -1, ringA[d]);
  }
#pragma unroll
  for (int d = 0; d < kSbPrefetch; ++d) request_checked(s_top - d, ring[d]);
  if (cprof) clog[1] = wall_clock64();  // operands staged
  if (J.given && !phase_a) {  // wait for the middle solution
    wait_for_partner(T);
    if (cprof) clog[2] = wall_clock64();  // middle solution arrived
    for (int rho = n_own + tid; rho < n_all; rho += nthr) xs[rho] = T.xsol[J.reversed ? np - 1 - rho : rho];
  }
  __syncthreads();
  // one super-step; `slot` is the ring entry that holds its operands (compile-time after unrolling)
  auto step_of = [&](double* xs, double* xout, int n_own, int rho_lo, int s, double* op) {  // (rows below rho_lo are not kept: phase A)
    const int r0 = kSbN * s;
    if (solver) {
      // x_J[r] = sum_c Winv[r][c] y[c]
      double acc = 0.0, acc1 = 0.0;
#pragma unroll
      for (int c = 0; c < 12; c += 2) {
        const double2 y = *reinterpret_cast<const double2*>(&xs[r0 + 12 * q + c]);
        acc = fma(op[c], y.x, acc);
        acc1 = fma(op[c + 1], y.y, acc1);
      }
      acc = pair_sum(acc + acc1);
      if (q == 0 && p_row < kSbN) {
        xj[p_row] = acc;
        if (r0 + p_row < n_own) xout[r0 + p_row] = acc;
      }
    }
    lds_barrier();
    if (!solver) {
      const int rho = r0 - 1 - p_row;
      double acc = 0.0, acc1 = 0.0;
#pragma unroll
      for (int c = 0; c < 12; c += 2) {
        const double2 x = *reinterpret_cast<const double2*>(&xj[12 * q + c]);
        acc = fma(op[c], x.x, acc);
        acc1 = fma(op[c + 1], x.y, acc1);
      }
      acc = pair_sum(acc + acc1);
      if (q == 0 && p_row < n_above && rho >= rho_lo) xs[rho] -= acc;
    }
    lds_barrier();
  };
  auto step = [&](int s, double* op) { step_of(xs, xout, n_own, 0, s, op); };
  if (phase_a) {  // the middle rows, solved here as block 0 solves them; then they are the given part of this sweep
#pragma unroll
    for (int d = 0; d < kSbPrefetch; ++d)
      if (sA_top - d >= sA_pub) step_of(xsA - baseA, xoutA - baseA, nA_own, baseA, sA_top - d, ringA[d]);
    for (int rho = n_own + tid; rho < n_all; rho += nthr) xs[rho] = xoutA[(np - 1 - rho) - baseA];
    __syncthreads();
    if (cprof) clog[2] = wall_clock64();  // middle solution ready
  }
  if (merged) {  // row r = tid / 2, the lane pair takes the even / odd columns (fixed order: even sum + odd sum)
    const int r = tid >> 1;
    double acc = 0.0;
    if (r < n_above)
      for (int c = tid & 1; c < n_above; c += 2) acc = fma(G[c * ldg + r], xs[n_own + c], acc);
    acc = pair_sum(acc);
    if ((tid & 1) == 0 && r < n_above) xs[n_own - n_above + r] -= acc;
    __syncthreads();
  }
  if (cprof) clog[3] = wall_clock64();  // sweep starts
  const int s_pub = (job == 0 && m_mid >= 0) ? m_mid / kSb : 0;  // block 0 publishes the middle solution once block row m_mid is solved
  int s = s_top;
  bool published = !(job == 0 && m_mid >= 0) || redo_mid;  // (redo_mid: block 1 solves the middle rows itself)
  const int s_lo = n_jobs == 1 ? j_lo / kSb : 0;
  while (s >= s_lo) {
#pragma unroll
    for (int d = 0; d < kSbPrefetch; ++d) {  // ring entry d holds the operands of super-block s (rotation by unrolling: no register moves)
      if (s < s_lo) break;
      step(s, ring[d]);
      request_checked(s - kSbPrefetch, ring[d]);
      if (!published && s == s_pub) {
        if (cprof) clog[4] = wall_clock64();  // middle rows solved
        for (int rho = 6 * m_mid + tid; rho < n_own; rho += nthr) T.xsol[rho] = xout[rho];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (every wave: its stores have reached the L2; the ONE agent-scope release — an L2 write-back on this part — is lane 0's below)
        lds_barrier();
        if (tid == 0) {
          __threadfence();
          __hip_atomic_store(T.join_flag, T.join_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (cprof) clog[5] = wall_clock64();  // middle solution published
        published = true;
      }
      --s;
    }
  }
  __syncthreads();
  if (cprof) clog[6] = wall_clock64();  // sweep done
  __shared__ double red[kCholThreads / 64];
  if (n_jobs == 1) {  // one-ended: the solution is in LDS; step = -x, delta = scale o step, reductions of the model cost change
    double gd = 0.0, dd = 0.0;
    for (int rho = tid; rho < np; rho += nthr) {
      const double step_v = -xout[rho];
      T.step_p[rho] = step_v;
      T.delta_p[rho] = T.scale_p[rho] * step_v;
      gd = fma(T.g_full[rho], step_v, gd);
      dd = fma(T.D2p[rho] * step_v, step_v, dd);
    }
    for (int b = tid; b < T.nb; b += nthr) {
      const double step_v = -T.xb[b];
      T.delta_b[b] = T.scale_b[b] * step_v;
      gd = fma(T.gb_s[b], step_v, gd);
      dd = fma(T.D2b[b] * step_v, step_v, dd);
    }
    gd = block_sum(gd, red);
    dd = block_sum(dd, red);
    if (tid == 0) {
      st->g_dot_step_pose = gd;
      st->d2_step2_pose = dd;
    }
    return;
  }
  // Two-ended: each sweep turns the rows it solved into the step outputs itself — step = -x, delta = scale o step and its share of the two
  // sums of the model cost change (DevState: block 0 the near share + the border unknowns, block 1 the far share; decide_step adds them) —
  // with the operands requested at the start of the kernel. (Until round 4 the block that finished last did it for all rows behind a
  // ticket: a release, an acquire and a round of loads, ~4 us at the end of the iteration's chain.)
  double gd = 0.0, dd = 0.0;
#pragma unroll
  for (int u = 0; u < kSbOut; ++u) {
    const int rho = tid + u * nthr;
    if (rho < n_own) {
      const int nat = J.reversed ? np - 1 - rho : rho;
      const double step_v = -xout[rho];
      T.xsol[nat] = -step_v;
      T.step_p[nat] = step_v;
      T.delta_p[nat] = o_sc[u] * step_v;
      gd = fma(o_gf[u], step_v, gd);
      dd = fma(o_d2[u] * step_v, step_v, dd);
    }
  }
  for (int rho = tid + kSbOut * nthr; rho < n_own; rho += nthr) {  // (windows beyond kSbOut * 256 scalar rows per end)
    const int nat = J.reversed ? np - 1 - rho : rho;
    const double step_v = -xout[rho];
    T.xsol[nat] = -step_v;
    T.step_p[nat] = step_v;
    T.delta_p[nat] = T.scale_p[nat] * step_v;
    gd = fma(T.g_full[nat], step_v, gd);
    dd = fma(T.D2p[nat] * step_v, step_v, dd);
  }
  if (job == 0)
    for (int b = tid; b < T.nb; b += nthr) {  // border unknowns of a bordered system (bias points, gravity)
      const double step_v = -T.xb[b];
      T.delta_b[b] = T.scale_b[b] * step_v;
      gd = fma(T.gb_s[b], step_v, gd);
      dd = fma(T.D2b[b] * step_v, step_v, dd);
    }
  gd = block_sum(gd, red);
  dd = block_sum(dd, red);
  if (tid == 0) {
    if (job == 0)
      st->g_dot_step_pose = gd, st->d2_step2_pose = dd;
    else
      st->g_dot_step_far = gd, st->d2_step2_far = dd;
  }
  if (cprof) clog[7] = wall_clock64();  // step outputs written (the block that finished last)
}

__global__ void __launch_bounds__(kCholThreads) k_band_backward_sb(Tables T, BackJob j0, BackJob j1, int m_mid, int n_jobs, int j_lo) {
  HS_DYNAMIC_LDS(smem);
  if (T.st->done) return;
  if (int(blockIdx.x) >= n_jobs) {  // ---------------- inverse builders (first wave) ----------------
    if (threadIdx.x >= 64) return;
    const int s0 = int(blockIdx.x) - n_jobs, n0 = sb_count(j0.n_rows);
    if (s0 < n0)
      sb_inverse(T, j0, 0, s0, smem);
    else
      sb_inverse(T, j1, 1, s0 - n0, smem);
    return;
  }
  sb_sweep(T, j0, j1, m_mid, n_jobs, j_lo, blockIdx.x, smem, false, nullptr);
}

}  // namespace hs
