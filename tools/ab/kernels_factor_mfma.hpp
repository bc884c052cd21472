// A/B ALTERNATIVE, PROFILING BUILDS ONLY (tools/build_profiling_lib.sh, -DHS_PROFILE_HOOKS=1; included by hyperslam_amd/csrc/kernels.hpp behind
// that switch): k_band_factor_mfma (round 2: trailing window in LDS-resident f64 MFMA tiles, HS_DEBUG_FLAGS 131072), which lost to k_band_factor_la and was superseded by k_band_factor_mx. Not part of the product library.
// kernels_factor_mfma.hpp — block-banded Cholesky with the rank-6 trailing update on the f64 matrix cores (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include <climits>
#include <utility>

#include "../../hyperslam_amd/csrc/kernels_factor.hpp"

namespace hs {

// ---------------------------------------------------------------------------------------------------------------------
// S = U'U for the block-banded reduced system (6 x 6 blocks, bw band blocks), fused forward solve — same inputs / outputs as
// k_band_factor_la (Ub, U_ii^-1, y = U^-T g; optional two-ended operation), but the rank-6 trailing update runs on the matrix
// cores: v_mfma_f64_16x16x4_f64 has the FMA rate of the vector ALU on gfx950 (64 clk per 16x16x4, tools/microbench/mfma_probe);
// what it removes is the operand traffic — two ds_read_b64 per lane feed 1024 FMAs, where the register-tile VALU update needed 72
// LDS operands per 216 FMAs and was LDS-issue bound (2200 clk per tile, p2_probe) — and with it the reason to keep the window
// in registers: a C tile costs four ds_read_b64 + four ds_write_b64 per lane and step.
//
// Data layout. The trailing window is the symmetric W x W diagonal window (W = 16 NT >= 6 bw + 12) of the partially eliminated
// matrix, resident in LDS in RING coordinates: matrix index rho lives at ring position rho mod W, so the window slides by six
// positions per block row without moving data. The ring is cut into NT x NT tiles of 16 x 16 (row stride 17: rows and columns
// are both conflict free); only tiles I <= J are stored (the entry of the unordered pair {rho, sigma}; diagonal tiles keep both
// orders). The right-hand side is a vector in the same ring coordinates.
//   step i :  C(I, J) -= X_i[:, I]' X_i[:, J]   C tile LDS -> accumulator registers (C/D layout of the instruction: lane l, register r
//             = row (l >> 4) + 4 r, column l & 15), two MFMAs (k = 0..3, k = 4..5 + two zero rows), back to LDS. X_i = row i of the
//             factor in ring layout (zeros at every position outside its trailing band): 2 NT ds_read_b64 per lane.
// Because the window is addressable, nothing is moved between register tiles and hand-over buffers: the panel wave reads its row
// straight from the window, the loader writes entering block rows straight into it. Block rows ENTER as column strips (all pairs
// (rho, sigma) with sigma in the entering block and rho resident): exactly the rows of the LOWER band, i.e. the rows of the upper
// band of the reversed system, which k_finalize_reduced writes for the two-ended factorisation — each job reads the other job's
// array. The two spare block rows (W >= 6 bw + 12) guarantee that an entering block row only takes positions of eliminated rows.
//
// Waves (hardware wave 3 = panel: waves are placed round robin on the four SIMDs, so it has SIMD 3 to itself):
//   compute 0 .. 2      tiles dealt round robin
//   panel               row i + 1: read from the window (phase A), -= X_i,1' X_i, 6 x 6 Cholesky redundantly in registers, column
//                       solves -> X_(i+1) (phase B), one lane per band column, branch free (as in k_band_factor_la)
//   loader              lower-band rows of the entering block rows HBM -> registers (two steps ahead) -> window (phase A)
//   storer              factor row, y, U_ii^-1 -> HBM; right-hand-side update (phase B)
// Two LDS-only barriers per block row: phase A (panel reads row i + 1, loader writes block row i + bw + 1: nobody modifies the window
// tiles) | phase B (tile updates, panel arithmetic). Two-ended operation (grid = 2) as in k_band_factor_la: job 1 eliminates the far
// end of the reversed system and hands over the Schur correction of the middle block rows (it enters zeros for the middle-middle
// pairs, so its window holds the pure correction); job 0 adds it at the junction and restarts its pipeline at the first middle row.
// ---------------------------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

/// Compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>) — tile indices must be constants, a
/// run-time index into the operand / accumulator arrays would put them into scratch memory.
template <class F, int... Is>
HSD void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
HSD void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int W>
HSD int ring_add(int p, int c) {  // (p + c) mod W for p in [0, W), c in [0, W]
  const int x = p + c;
  return x >= W ? x - W : x;
}

/// Tile q of the enumeration of the upper tiles I <= J (row major).
template <int NT>
__host__ __device__ constexpr int mfma_tile_I(int q) {
  int I = 0;
  while (q >= NT - I) q -= NT - I, ++I;
  return I;
}
template <int NT>
__host__ __device__ constexpr int mfma_tile_J(int q) {
  int I = 0;
  while (q >= NT - I) q -= NT - I, ++I;
  return I + q;
}

template <int NT>
struct MfmaGeom {
  static constexpr int W = 16 * NT;
  static constexpr int LDX = W + ((W % 32 == 16) ? 32 : 16);  // row stride of X in LDS (> W: column W = y); rows k, k + 1 in different bank halves
  static constexpr int NTILE = NT * (NT + 1) / 2;
  static constexpr int TS = 16 * 17;                          // doubles per tile (row stride 17)
  static constexpr int PC = (W + 1 + 63) / 64;                // panel columns per lane
  static constexpr int PW = (W + 63) / 64;                    // ring positions per lane (right-hand-side update)
  // LDS (doubles): window NTILE x TS | gring W | xring 2 x 6 x LDX (column W of a row = y) | stage 6 x LDX | rowbuf 2 x 6 x LDX | dscr 36
  static constexpr int kWin = 0, kG = kWin + NTILE * TS, kX = kG + W, kS = kX + 12 * LDX, kR = kS + 6 * LDX, kD = kR + 12 * LDX,
                       kTotal = kD + 36;
  __host__ __device__ static constexpr int tile_row_base(int I) { return I * NT - I * (I - 1) / 2; }  // index of tile (I, I)
};

/// LDS offset (doubles) of the element of the ordered pair of ring positions (a, b) in the upper-tile storage. Pairs in different
/// tiles have one element (either order gives it); a pair inside a diagonal tile has two, (a, b) and (b, a).
template <int NT>
HSD int mfma_pair_addr(int a, int b) {
  using G = MfmaGeom<NT>;
  const int Ia = a >> 4, Ib = b >> 4, ca = a & 15, cb = b & 15;
  const int ta = Ia * NT - ((Ia * (Ia - 1)) >> 1), tb = Ib * NT - ((Ib * (Ib - 1)) >> 1);
  return Ia <= Ib ? (ta + Ib - Ia) * G::TS + ca * 17 + cb : (tb + Ia - Ib) * G::TS + cb * 17 + ca;
}

/// S(rho, sigma) of the job's system for a pair that enters the window (own order), with the job's zero rules.
HSD double mfma_job_value(const MfmaJob& J, int np, int ncb, int rho, int sigma) {
  if (rho > sigma) {
    const int t = rho;
    rho = sigma, sigma = t;
  }
  const int e = sigma / 6, off = 6 * e + 5 - rho;  // band offset in the lower row of sigma
  const bool ok = rho >= 0 && off < ncb && e < J.enter_limit && !(rho / 6 >= J.zero_from && e >= J.zero_from);
  const double v = J.L[ok ? size_t(np - 1 - sigma) * ncb + off : 0];
  return ok ? v : 0.0;
}

/// Phase B of a compute wave: its tiles C -= X_i[:, I]' X_i[:, J]. By-product: the entries of block row i + 2 (ring position p_row), final
/// for the panel after this update (it applies X_(i+1) itself), go from the accumulator registers to rowbuf in band order — the panel
/// never reads the tiled window in the steady state. rowbuf[k][c]: row p_row + k, band column c (ring position p_row + c).
template <int NT, int NC, int WV>
HSD void mfma_update_tiles(double* win, const double* x, int l, int p_i, int p_row, int ncb, double* rowbuf) {
  using G = MfmaGeom<NT>;
  constexpr int LDX = G::LDX, TW = (G::NTILE - WV + NC - 1) / NC;
  const int l15 = l & 15, g4 = l >> 4;
  double xf[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // the pivot's own six positions hold U_ii (read by the storer and the next panel): not part of the trailing update
      // k = 6, 7 of the second MFMA: zero rows (not stored)
      const double v = x[(g == 1 && g4 >= 2 ? 0 : g4 + 4 * g) * LDX + 16 * t + l15];
      xf[t][g] = (unsigned(16 * t + l15 - p_i) < 6u || (g == 1 && g4 >= 2)) ? 0.0 : v;
    }
  f64x4 acc[TW];
  const int el = g4 * 17 + l15;  // element (row g4, column l15) inside a tile; register r adds 4 rows
  static_for<TW>([&](auto mc) {
    constexpr int m = decltype(mc)::value, q = WV + NC * m, I = mfma_tile_I<NT>(q), Jt = mfma_tile_J<NT>(q);
    const double* c = win + (G::tile_row_base(I) + Jt - I) * G::TS + el;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[m][rr] = c[rr * 4 * 17];
  });
  // program order: MFMAs of tile m, then the stores of tile m - 1 (its result is ready by then): the stores issue while the matrix
  // pipe works on tile m instead of after all MFMAs
  static_for<TW + 1>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    if constexpr (m < TW) {
      constexpr int q = WV + NC * m, I = mfma_tile_I<NT>(q), Jt = mfma_tile_J<NT>(q);
#pragma unroll
      for (int g = 0; g < 2; ++g) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xf[I][g], xf[Jt][g], acc[m], 0, 0, 0);
    }
    if constexpr (m >= 1) {
      constexpr int q = WV + NC * (m - 1), I = mfma_tile_I<NT>(q), Jt = mfma_tile_J<NT>(q);
      double* c = win + (G::tile_row_base(I) + Jt - I) * G::TS + el;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) c[rr * 4 * 17] = acc[m - 1][rr];
      // does the tile hold entries of rows p_row .. p_row + 5? (ring interval test, wave uniform)
      constexpr int W = G::W;
      int sr = 16 * I - p_row + 15, sc = 16 * Jt - p_row + 15;
      sr += sr < 0 ? W : 0, sc += sc < 0 ? W : 0;
      if (sr <= 20) {  // as rows of the tile: element (16 I + r, 16 Jt + l15) -> rowbuf[row - p_row][column - p_row]
        int cc = 16 * Jt + l15 - p_row;
        cc += cc < 0 ? W : 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          int k = 16 * I + g4 + 4 * rr - p_row;
          k += k < 0 ? W : 0;
          if (k < 6 && cc < ncb) rowbuf[k * LDX + cc] = acc[m - 1][rr];
        }
      }
      if (I != Jt && sc <= 20) {  // as columns (the stored element also is the pair the other way round); a diagonal tile has both already
        int k = 16 * Jt + l15 - p_row;
        k += k < 0 ? W : 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          int cc = 16 * I + g4 + 4 * rr - p_row;
          cc += cc < 0 ? W : 0;
          if (k < 6 && cc < ncb) rowbuf[k * LDX + cc] = acc[m - 1][rr];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

template <int NT, int NC>
__global__ void __launch_bounds__(64 * (NC + 3)) k_band_factor_mfma(Tables T) {
  using G = MfmaGeom<NT>;
  constexpr int W = G::W, LDX = G::LDX, PC = G::PC, PW = G::PW;
  static_assert(G::LDX >= G::W + 1, "column W of an X row carries y");
  static_assert(NC == 3, "wave roles below assume three compute waves");
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const MfmaJob J = T.mj[blockIdx.x];
  const int tid = threadIdx.x, hw = tid >> 6, l = tid & 63;
  const int bw = T.bw, ncb = 6 * bw, np = T.np, n_steps = J.n_steps, m_at = J.merge_at;
  double* win = smem + G::kWin;
  double* gring = smem + G::kG;
  double* xring = smem + G::kX;
  double* stage = smem + G::kS;    // entering block row in band order: stage[q][t], t == ncb: right-hand side (loader -> helpers)
  double* rowbuf = smem + G::kR;   // next pivot row in band order: rowbuf[k][c], column W: right-hand side (helpers -> panel)
  double* dscr = smem + G::kD;
  __shared__ int fail;
  if (tid == 0) fail = 0;
  constexpr int nthreads = 64 * (NC + 3);
  // ---- prologue (all waves): block rows 0 .. bw at ring position = matrix index ----
  {
    const int n_in = 6 * (bw + 1);  // <= W - 6
    for (int e = tid; e < G::NTILE * 256; e += nthreads) {
      const int q = e >> 8, r = (e >> 4) & 15, c = e & 15;
      int I = 0, qq = q;
      while (qq >= NT - I) qq -= NT - I, ++I;
      const int a = 16 * I + r, b = 16 * (I + qq) + c;
      win[q * G::TS + r * 17 + c] = (a < n_in && b < n_in) ? mfma_job_value(J, np, ncb, a, b) : 0.0;
    }
    for (int a = tid; a < W; a += nthreads) gring[a] = (a < n_in && a / 6 < J.enter_limit && a / 6 < J.zero_from) ? J.g[a] : 0.0;
  }
  __syncthreads();
  // Junction (job 0, every wave): add the other end's Schur correction of the middle block rows to the window (p_m = ring position of
  // block row m). D is dm x (dm + 1) in middle-local coordinates, last column = right-hand side.
  auto junction_merge = [&](int p_m) {
    wait_for_partner(T);
    const int dm = 6 * (bw - 1);
    const double* D = J.win;
    for (int e = tid; e < dm * dm; e += nthreads) {
      const int da = e / dm, db = e - da * dm;
      const int a = ring_add<W>(p_m, da), b = ring_add<W>(p_m, db);
      // a pair in two different tiles has one element: visit it once; inside a diagonal tile both orders are elements of their own
      if (da <= db || (a >> 4) == (b >> 4)) win[mfma_pair_addr<NT>(a, b)] += D[size_t(da) * (dm + 1) + db];
    }
    for (int da = tid; da < dm; da += nthreads) gring[ring_add<W>(p_m, da)] += D[size_t(da) * (dm + 1) + dm];
  };
  // Phase A of step i (all six waves, one row of the entering block row i + bw + 1 each): SCATTER the staged row q (stage[q][t]: pairs
  // (ring position p_lo + t, p_e + q), t < ncb; t == ncb: right-hand side) into the tiled window. The second element of each pair is wave
  // uniform, so the tile address (mfma_pair_addr) splits into a uniform and a lane part. This is the only work of phase A: nobody
  // updates tiles in it, so the plain stores cannot race with a read-modify-write of a compute wave.
  auto scatter_row = [&](int q, int p_next) {
    const int p_lo = ring_add<W>(p_next, 6), p_e = ring_add<W>(p_lo, 6 * (bw - 1));
    const int b = p_e + q, Iu = b >> 4, cu = b & 15, tu = Iu * NT - ((Iu * (Iu - 1)) >> 1);
    const int u1 = Iu * G::TS + cu, u2 = (tu - Iu) * G::TS + cu * 17, um = tu * G::TS + cu * 17;
    double val[PC];
#pragma unroll
    for (int m = 0; m < PC; ++m) val[m] = stage[q * LDX + (l + 64 * m <= ncb ? l + 64 * m : 0)];
#pragma unroll
    for (int m = 0; m < PC; ++m) {
      const int t = l + 64 * m;
      const int a = t < W ? ring_add<W>(p_lo, t) : 0;
      const int Ia = a >> 4, ca = a & 15, ta = __mul24(Ia, NT) - (__mul24(Ia, Ia - 1) >> 1);
      if (t < ncb) {
        win[Ia <= Iu ? __mul24(ta - Ia, G::TS) + ca * 17 + u1 : __mul24(Ia, G::TS) + ca + u2] = val[m];
        if (Ia == Iu) win[um + ca] = val[m];  // diagonal tile: the mirrored element too
      } else if (t == ncb) {
        gring[b] = val[m];
      }
    }
  };
  const bool prof = prof_enabled(T.debug_flags, 16) && l == 0 && blockIdx.x == 0;  // HS_DEBUG_FLAGS: phase timestamps -> hs_debug_read
  long long* tlog = reinterpret_cast<long long*>(T.xpart);

  if (hw < 3) {  // ================================ compute waves ================================
    lds_barrier();  // P1: X_0 published
    int p_i = 0;
    for (int i = 0; i < n_steps; ++i) {
      if (prof && hw == 0) tlog[8 * i + 4] = wall_clock64();
      scatter_row(hw, ring_add<W>(p_i, 6));
      if (prof && hw == 0) tlog[8 * i + 5] = wall_clock64();
      lds_barrier();  // A -> B
      if (prof && hw == 0) tlog[8 * i + 0] = wall_clock64();
      const double* x = xring + (i & 1) * 6 * LDX;
      double* rb = rowbuf + (i & 1) * 6 * LDX;  // block row i + 2 for the panel of the next step
      const int p_row = ring_add<W>(p_i, 12);
      if (hw == 0) mfma_update_tiles<NT, NC, 0>(win, x, l, p_i, p_row, ncb, rb);
      if (hw == 1) mfma_update_tiles<NT, NC, 1>(win, x, l, p_i, p_row, ncb, rb);
      if (hw == 2) mfma_update_tiles<NT, NC, 2>(win, x, l, p_i, p_row, ncb, rb);
      if (prof && hw == 0) tlog[8 * i + 1] = wall_clock64();
      p_i = ring_add<W>(p_i, 6);
      lds_barrier();  // B -> A
      if (m_at >= 0 && i + 1 == m_at) {
        junction_merge(p_i);
        lds_barrier();  // J1: window merged
        lds_barrier();  // J2: X_m published
      }
    }
  } else if (hw == 4) {  // ================================ loader ================================
    // lane l owns band columns t = l + 64 m of an entering block row (t < ncb; t == ncb: right-hand side), all six rows
    double va[PC][6], vb[PC][6];
    auto fetch = [&](double (*v)[6], int e) {  // block row e: S(rho = 6 (e - bw + 1) + t, sigma = 6 e + q) with the job's zero rules
      // entries that enter as zeros are read from a zero in memory: a select on the loaded value would make the wave wait for the data
      // here, at issue time
      const bool ok_e = e < J.enter_limit;
      const long long row0 = ok_e ? (long long)(np - 1 - 6 * e) * ncb : 0;  // row of sigma = 6 e; sigma + q: q rows earlier
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const int t = l + 64 * m;
        const int rb = e - bw + 1 + int(unsigned(t) / 6u);  // block row of rho (only used when rho >= 0)
        const bool band = t < ncb, rhs = t == ncb;
        const bool ok = band && ok_e && 6 * (e - bw + 1) + t >= 0 && !(rb >= J.zero_from && e >= J.zero_from);
        const bool ok_g = rhs && ok_e && e < J.zero_from;
        const double* src = ok ? J.L + row0 + (ncb - 1 - t) : (ok_g ? J.g + 6 * e : J.zero);
        const long long stride = ok ? -(long long)ncb : (ok_g ? 1 : 0);
#pragma unroll
        for (int q = 0; q < 6; ++q) v[m][q] = src[q * stride];
      }
    };
    auto stage_write = [&](const double (*v)[6]) {
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const int t = l + 64 * m;
        if (t <= ncb) {
#pragma unroll
          for (int q = 0; q < 6; ++q) stage[q * LDX + t] = v[m][q];
        }
      }
    };
    auto junction_io = [&](int i_done, int p_next) {
      if (m_at >= 0 && i_done + 1 == m_at) {
        junction_merge(p_next);
        lds_barrier();  // J1
        lds_barrier();  // J2
      }
    };
    fetch(va, bw + 1), fetch(vb, bw + 2);
    stage_write(va);
    fetch(va, bw + 3);
    lds_barrier();  // P1
    int p_i = 0;
    auto step = [&](double (*v)[6], int i) {  // v holds block row i + bw + 2 (fetched two steps ago)
      scatter_row(4, ring_add<W>(p_i, 6));
      lds_barrier();           // A -> B
      stage_write(v);          // scattered in phase A of step i + 1
      fetch(v, i + bw + 4);
      p_i = ring_add<W>(p_i, 6);
      lds_barrier();           // B -> A
      junction_io(i, p_i);
    };
    // no branch between the two halves: with a conditional second half the compiler's vmcnt bookkeeping merges both paths at the loop
    // head and waits for loads issued one step ago instead of two
    int i = 0;
    for (; i + 1 < n_steps; i += 2) step(vb, i), step(va, i + 1);
    if (i < n_steps) step(vb, i);
  } else if (hw == 5) {  // ================================ storer ================================
    lds_barrier();  // P1: X_0 complete
    int p_i = 0;
    for (int i = 0; i < n_steps; ++i) {
      scatter_row(5, ring_add<W>(p_i, 6));
      lds_barrier();  // A -> B
      const double* x = xring + (i & 1) * 6 * LDX;
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const int c = l + 64 * m;
        if (c < ncb) {
          const int pos = ring_add<W>(p_i, c);
          double* dst = J.Ub + size_t(6 * i) * ncb + c;
#pragma unroll
          for (int a = 0; a < 6; ++a) dst[a * ncb] = x[a * LDX + pos];
        }
      }
      double y[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) y[k] = x[k * LDX + W];
      if (l < 6) {
        double yl = y[0];
#pragma unroll
        for (int k = 1; k < 6; ++k) yl = l == k ? y[k] : yl;
        J.ybuf[6 * i + l] = yl;
      }
      // right-hand side of the trailing rows: g -= X_i' y_i (X is zero outside the trailing band; the pivot's own positions are dead)
#pragma unroll
      for (int m = 0; m < PW; ++m) {
        const int pos = l + 64 * m;
        if (pos < W) {
          double acc = gring[pos];
#pragma unroll
          for (int k = 0; k < 6; ++k) acc = fma(-x[k * LDX + pos], y[k], acc);
          gring[pos] = acc;
        }
      }
      wait_lds();  // (same wave: the stores above are visible to the loads below)
      if (l < 6) rowbuf[(i & 1) * 6 * LDX + l * LDX + W] = gring[ring_add<W>(p_i, 12) + l];  // right-hand side of block row i + 2 -> panel
      {  // W = U_ii^-1 (upper triangular, packed): lane c < 6 solves U w = e_c. U_ii sits at the pivot's own positions of X (upper part)
        const int c = l < 6 ? l : 0;
        double w[6];
#pragma unroll
        for (int a = 5; a >= 0; --a) {
          double t = a == c ? 1.0 : 0.0;
#pragma unroll
          for (int k = a + 1; k < 6; ++k) t = fma(-x[a * LDX + p_i + k], w[k], t);
          const double u = x[a * LDX + p_i + a];  // reciprocal of the diagonal: hardware estimate + two Newton steps
          double r = __builtin_amdgcn_rcp(u);
          r = fma(fma(-u, r, 1.0), r, r);
          r = fma(fma(-u, r, 1.0), r, r);
          w[a] = t * r;
        }
        if (l < 6) {
#pragma unroll
          for (int a = 0; a < 6; ++a)
            if (a <= c) J.Ubk[size_t(i) * 24 + (a * 6 - a * (a - 1) / 2 + (c - a))] = w[a];
        }
      }
      p_i = ring_add<W>(p_i, 6);
      lds_barrier();  // B -> A
      if (m_at >= 0 && i + 1 == m_at) {
        junction_merge(p_i);
        lds_barrier();  // J1
        lds_barrier();  // J2
      }
    }
  } else {  // ================================ panel wave (hardware wave 3) ================================
#define UIDX(a, c) ((a) * 6 - (a) * ((a)-1) / 2 + ((c) - (a)))
    // Column c = l + 64 m of the pivot row in band order (c < ncb), ring position (p_r + c) mod W; the last slot of lane 63 carries the
    // right-hand side. Slots with ncb <= c < W only zero their ring position (X is zero outside the trailing band).
    double v[PC][6];
    int pos[PC];
    bool is_rhs[PC], is_band[PC];
    long long* plog = tlog + 8 * 1024;
    auto panel_read = [&](int r, int pr) {  // phase A: row block r from the window (as it stands: updated through X_(r-2) or X_(r-1))
      if (prof) plog[8 * r + 0] = wall_clock64();
      // element (a = pr + k: wave uniform, b = pos: lane): uniform part + lane part for either tile order (mfma_pair_addr)
      int ua1[6], ua2[6], Iu[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int a = pr + k, Ia = a >> 4, ca = a & 15, ta = Ia * NT - ((Ia * (Ia - 1)) >> 1);
        Iu[k] = Ia, ua1[k] = (ta - Ia) * G::TS + ca * 17, ua2[k] = Ia * G::TS + ca;
      }
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const int c = l + 64 * m;
        is_rhs[m] = m == PC - 1 && l == 63;
        is_band[m] = c < ncb && !is_rhs[m];
        pos[m] = is_rhs[m] ? W : (c < W ? ring_add<W>(pr, c) : 0);
        const int b = is_rhs[m] ? 0 : pos[m], Ib = b >> 4, cb = b & 15, tb = Ib * NT - ((Ib * (Ib - 1)) >> 1);
        const int lb1 = Ib * G::TS + cb, lb2 = (tb - Ib) * G::TS + cb * 17;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const double* src = is_rhs[m] ? gring + pr + k : win + (Iu[k] <= Ib ? ua1[k] + lb1 : ua2[k] + lb2);
          v[m][k] = *src;
        }
      }
      if (prof) {
        wait_lds();
        plog[8 * r + 4] = wall_clock64();
      }
    };
    auto panel_slots = [&](int pr) {
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const int c = l + 64 * m;
        is_rhs[m] = m == PC - 1 && l == 63;
        is_band[m] = c < ncb && !is_rhs[m];
        pos[m] = is_rhs[m] ? W : (c < W ? ring_add<W>(pr, c) : 0);
      }
    };
    // update: the row still lacks X_(r-1) (look-ahead); direct: v was read from the window by panel_read (first rows, junction),
    // otherwise the compute waves left the row in rowbuf[r & 1] one step ago (by-product of their update with X_(r-2))
    auto panel_compute = [&](int r, int pr, bool update, bool direct) {  // phase B
      if (prof) plog[8 * r + 1] = wall_clock64();
      if (update && !direct) {
        panel_slots(pr);
        const double* rb = rowbuf + (r & 1) * 6 * LDX;
#pragma unroll
        for (int m = 0; m < PC; ++m) {
          const int c = l + 64 * m, cc = is_rhs[m] ? W : (c < W ? c : 0);
#pragma unroll
          for (int k = 0; k < 6; ++k) v[m][k] = rb[k * LDX + cc];
        }
      }
      const double* xp = xring + ((r - 1) & 1) * 6 * LDX;
      double* xo = xring + (r & 1) * 6 * LDX;
      if (update) {
        double B[6][6];  // block 1 of X_(r-1) = its columns of block row r: six consecutive ring positions, 16-byte aligned
        const double* bp = xp + pr;
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
          for (int a = 0; a < 6; a += 2) {
            const double2 t = *reinterpret_cast<const double2*>(bp + k * LDX + a);
            B[k][a] = t.x, B[k][a + 1] = t.y;
          }
#pragma unroll
        for (int m = 0; m < PC; ++m) {
          double xc[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            const double t = xp[k * LDX + pos[m]];  // (right-hand-side slot: column W = y)
            xc[k] = (is_band[m] || is_rhs[m]) ? t : 0.0;
          }
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int k = 0; k < 6; ++k) v[m][a] = fma(-B[k][a], xc[k], v[m][a]);
        }
      }
      if (l < 6) {
#pragma unroll
        for (int a = 0; a < 6; ++a) dscr[6 * a + l] = v[0][a];
      }
      wait_lds();  // same wave: LDS is in order, only the compiler must not reorder
      double U[21], inv[6], dmin;
      {
        int pidx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int c = a; c < 6; ++c) U[pidx++] = dscr[6 * a + c];
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
        const double nrs = -rs;  // off-diagonal entries are kept negated (products of two of them are unchanged)
#pragma unroll
        for (int c = a + 1; c < 6; ++c) {
          double t = U[UIDX(a, c)];
#pragma unroll
          for (int k = 0; k < a; ++k) t = fma(-U[UIDX(k, a)], U[UIDX(k, c)], t);
          U[UIDX(a, c)] = t * nrs;
        }
      }
      if (!(dmin > 0.0) && l == 0) fail = 1;
      if (prof) plog[8 * r + 2] = wall_clock64();
      // x = U^-T v per column; one unconditional store stream: columns c < 6 reproduce U_rr (upper part incl. the diagonal: read by the
      // storer and the next panel, masked out of the trailing update by the compute waves), band columns X_r, the rest zeros, y at W
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        double x[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          double t = v[m][a];
#pragma unroll
          for (int k = 0; k < a; ++k) t = fma(U[UIDX(k, a)], x[k], t);  // U holds -u_ka
          x[a] = t * inv[a];
        }
        const int c = l + 64 * m;
        if (c < W || is_rhs[m]) {
          const bool live = is_band[m] || is_rhs[m];
#pragma unroll
          for (int a = 0; a < 6; ++a) xo[a * LDX + pos[m]] = live ? x[a] : 0.0;
        }
      }
      if (prof) plog[8 * r + 3] = wall_clock64();
    };
    panel_read(0, 0);
    panel_compute(0, 0, false, true);
    bool direct = true;  // the row after a freshly factored one is read from the window itself: nobody has gathered it
    if (n_steps > 1) panel_read(1, 6);
    lds_barrier();  // P1: X_0 published
    int p_next = 6;  // ring position of block row i + 1
    for (int i = 0; i < n_steps; ++i) {
      const bool junction = m_at >= 0 && i + 1 == m_at;  // no look-ahead across the junction: row m changes there
      const bool ahead = i + 1 < n_steps && !junction;
      scatter_row(3, p_next);
      lds_barrier();  // A -> B
      if (ahead) panel_compute(i + 1, p_next, true, direct), direct = false;
      lds_barrier();  // B -> A
      if (junction) {
        junction_merge(p_next);
        lds_barrier();  // J1: window merged
        panel_read(m_at, p_next);
        panel_compute(m_at, p_next, false, true);
        if (m_at + 1 < n_steps) panel_read(m_at + 1, ring_add<W>(p_next, 6)), direct = true;
        lds_barrier();  // J2: X_m published
      }
      p_next = ring_add<W>(p_next, 6);
    }
#undef UIDX
  }
  __syncthreads();
  if (tid == 0 && fail) st->chol_failed = 1;
  if (J.dump) {  // job 1: the pure correction of the middle block rows, in job 0's coordinates (index reversal inside the middle block)
    const int dm = 6 * (bw - 1);
    const int p_m = (6 * n_steps) % W;
    double* D = J.win;
    for (int e = tid; e < dm * dm; e += nthreads) {
      const int da = e / dm, db = e - da * dm;
      const int a = ring_add<W>(p_m, da), b = ring_add<W>(p_m, db);
      D[size_t(dm - 1 - da) * (dm + 1) + (dm - 1 - db)] = win[mfma_pair_addr<NT>(a, b)];
    }
    for (int da = tid; da < dm; da += nthreads) D[size_t(dm - 1 - da) * (dm + 1) + dm] = gring[ring_add<W>(p_m, da)];
    __threadfence();
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      __hip_atomic_store(T.join_flag, T.join_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace hs
