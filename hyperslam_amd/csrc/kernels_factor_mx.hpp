// kernels_factor_mx.hpp — block-banded Cholesky of the reduced system with the trailing window RESIDENT IN THE ACCUMULATORS of the f64
// matrix cores (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include <climits>
#include <utility>

#include "kernels_factor.hpp"

namespace hs {

// ---------------------------------------------------------------------------------------------------------------------
// S = U'U for the block-banded reduced system (6 x 6 blocks, bw <= 16 band blocks), fused forward solve: same outputs as
// k_band_factor_la (factor rows Ub, inverted diagonal blocks Ubk, y = U^-T g; one- and two-ended operation), so the sweeps
// behind it do not change. Replaces what CHOLMOD does for /root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:46-48.
//
// Why another kernel. k_band_factor_la takes 1.20 us per block row and both sides of its step are instruction-issue bound:
// the compute waves spend 72 LDS operands per 216 FMAs of a 6 x 6 register tile, the panel wave carries two columns per lane.
// Here the rank-6 trailing update is v_mfma_f64_16x16x4_f64 on tiles that never leave the accumulator registers: an MFMA
// wave needs TWELVE ds_read_b64 per block row for all its operands (the A and the B operand of tile (I, J) are the same
// register set: X_i[k][16 I + j] and X_i[k][16 J + j], lane (k, j)). gfx950 executes the f64 MFMA on the SIMD's fp64 pipe — it
// has the vector FMA rate, and a panel wave next to an MFMA wave stretched from 0.84 to 1.2 us (measured) — so the matrix cores do
// not add arithmetic; what they remove are operand traffic and issue slots, and the SIMDs are dealt accordingly (waves are
// placed round robin on the four SIMDs):
//   waves 0, 4 (SIMD 0) and 1, 5 (SIMD 1)   the 21 tiles in four MFMA waves (the operand loads and hand-overs of one wave run while
//                                           the pipe works on the other's tiles); + wave 8: inverts the diagonal blocks for the sweeps
//   waves 2 (SIMD 2) and 3 (SIMD 3)         the PANEL, one column per lane, 48 ring columns each (wave 3 also the right-hand side):
//                                           update of the own column with X_(i-1), the pivot's 6 x 6 diagonal block broadcast with
//                                           v_readlane from six extra lanes that redo its columns in BOTH waves, redundant register
//                                           Cholesky, column solve, publish — and nothing else: the panel is a latency chain (LDS loads
//                                           -> update -> factor -> solve -> LDS writes), 0.85 us per block row; a flag word behind a U
//                                           published by one wave cost + 0.3 us, 480 instead of 272 instructions 1.5 us
//   wave 6 (SIMD 2) storer, wave 7 (SIMD 3) loader: factor row and y -> HBM one iteration later + the right-hand side of the trailing
//                                           rows; block rows HBM -> stage two iterations ahead (next to the MFMA waves the storer took
//                                           1.3 us per block row: their MFMAs keep the fp64 pipes busy for most of a step)
// One column per lane in RING coordinates (matrix index rho lives at ring position rho mod 96 for its whole life): a lane's
// own result of the previous block row is its operand of the next one, the right-hand side of a position lives in a
// register of its lane, and the window slides by six positions per block row without moving anything.
//
// Data. W = 96 = 6 x 6 tiles of 16 x 16; only tiles I <= J are kept (the entry of the unordered pair of ring positions;
// diagonal tiles hold both orders), 21 tiles, five or six per MFMA wave. Tile element of lane l,
// register r: row (l >> 4) + 4 r, column l & 15 (mfma_probe). Per block row i an MFMA wave
//   * applies X_i (zero outside its trailing band, so every tile can be updated blindly),
//   * EXTRACTS block row i + 2 — final for the panel, which applies X_(i+1) itself (look-ahead) — from the registers into
//     rowbuf (per-lane predicated ds_write_b64; both orientations of a pair),
//   * ENTERS block row i + 16 from the loader's stage into the positions of block row i (per-lane predicated ds_read_b64).
// Everything in LDS is in ring coordinates too (X rows, the pivot rows, the staged rows), and the loop over the block rows is unrolled over
// the 16 PHASES of the ring (6 x 16 = 96): which tiles, registers and lanes hold a block row is a compile-time pattern per phase — no
// index arithmetic, no wave-uniform branches (a taken branch costs ~38 cycles: the first version spent 28 of them per block row).
// Block rows enter as column strips (all pairs (rho, sigma) with sigma in the entering block, rho resident) = rows of the
// LOWER band = rows of the upper band of the reversed system, which the assembly writes for the far end of the two-ended
// factorisation anyway: each job reads the other job's array (MfmaJob, problem.hpp).
// One LDS-only barrier per block row and nothing else.
// LDS (doubles): xring 2 x 6 x 112 (X rows, column 96 = y) | rowbuf 2 x 6 x 112 (pivot rows, column 96 = right-hand side) |
// stage 2 x 6 x 112 (entering block rows: st[q][a] = pair (a, p_e + q), column 96 = right-hand side).
// ---------------------------------------------------------------------------------------------------------------------
typedef double mx_f64x4 __attribute__((vector_size(32)));

constexpr int kMxW = 96, kMxLdx = 112, kMxWaves = 9, kMxThreads = 64 * kMxWaves;
constexpr int kMxX = 0, kMxR = kMxX + 12 * kMxLdx, kMxS = kMxR + 12 * kMxLdx, kMxD = kMxS + 12 * kMxLdx, kMxLds = kMxD + 16;
constexpr int kMxDiagLane = 56;  // lanes 56 .. 61 of both panel waves redo the six columns of the pivot's diagonal block

/// Two-ended factorisation: block rows the near end takes more than half of the non-middle rows (launch_factor; the test harness uses the same rule).
constexpr int two_ended_lead(bool mx) { return mx ? 2 : 3; }

/// Bands this kernel holds: 16 block rows of ring = the pivot row, the one in the panel's hands, bw - 2 trailing ones, the entering one.
constexpr bool mx_fits(int bw) { return bw >= 3 && bw <= 16; }
/// Bands of 15 and 16 control points (order-6 splines) use all 16 slots of the ring: see WIDE below.
constexpr bool mx_wide(int bw) { return bw > 14; }

template <class F, int... Is>
HSD void mx_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
/// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): accumulator indices must be compile-time constants.
template <int N, class F>
HSD void mx_static_for(F&& f) {
  mx_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

HSD int mx_add(int p, int c) {  // (p + c) mod W, p in [0, W), c in [0, W]
  const int x = p + c;
  return x >= kMxW ? x - kMxW : x;
}
HSD int mx_sub(int a, int p) {  // (a - p) mod W, both in [0, W)
  const int d = a - p;
  return d < 0 ? d + kMxW : d;
}
/// Tile q of the row-major enumeration of the tiles I <= J.
__host__ __device__ constexpr int mx_tile_I(int q) {
  int I = 0;
  while (q >= 6 - I) q -= 6 - I, ++I;
  return I;
}
__host__ __device__ constexpr int mx_tile_J(int q) {
  int I = 0;
  while (q >= 6 - I) q -= 6 - I, ++I;
  return I + q;
}
/// The 21 tiles are dealt to four MFMA waves, two per SIMD (groups 0, 1 on SIMD 0; 2, 3 on SIMD 1): the operand loads and the hand-overs
/// of one wave run while the matrix pipe works on the other's tiles. The deal (a search over assignments, tools/mx_tile_deal.py) minimises
/// the tiles a SIMD has to update per block row, summed over the 16 phases: a tile whose rows or columns lie entirely outside the
/// trailing band of the pivot row (15 of the 21 in ten of the sixteen phases) is skipped at compile time. 7 - 11 tiles per SIMD and phase.
constexpr int kMxGroupSize[4] = {6, 5, 5, 5};
constexpr int kMxGroupTile[4][6] = {{2, 5, 9, 10, 12, 16}, {3, 6, 14, 18, 19, -1}, {0, 4, 7, 15, 17, -1}, {1, 8, 11, 13, 20, -1}};
/// Does tile index t (16 ring positions) hold a position of the trailing band [12, hi) of the pivot row at ring position p_i? (hi = 84: bands of
/// up to 14 control points; 96: WIDE)
__host__ __device__ constexpr bool mx_index_active(int t, int p_i, int hi) {
  for (int off = 12; off < hi; ++off)
    if ((p_i + off) % kMxW / 16 == t) return true;
  return false;
}

/// S(rho, sigma) of the job's system (own order) with the job's zero rules; the pair may come in either order.
HSD double mx_job_value(const MfmaJob& J, int np, int ncb, int rho, int sigma) {
  if (rho > sigma) {
    const int t = rho;
    rho = sigma, sigma = t;
  }
  const int e = sigma / 6, off = 6 * e + 5 - rho;  // band offset in the lower row of sigma
  const bool ok = rho >= 0 && off < ncb && e < J.enter_limit && !(rho / 6 >= J.zero_from && e >= J.zero_from);
  const double v = J.L[ok ? size_t(np - 1 - sigma) * ncb + off : 0];
  return ok ? v : 0.0;
}
/// The same value through a pointer select (entries that enter as zeros are read from a zero in memory): no instruction depends on the
/// loaded value, so a wave can carry on — through barriers — until it needs the register.
HSD const double* mx_job_address(const MfmaJob& J, int np, int ncb, int rho, int sigma) {
  if (rho > sigma) {
    const int t = rho;
    rho = sigma, sigma = t;
  }
  const int e = sigma / 6, off = 6 * e + 5 - rho;
  const bool ok = rho >= 0 && off < ncb && e < J.enter_limit && !(rho / 6 >= J.zero_from && e >= J.zero_from);
  return ok ? J.L + size_t(np - 1 - sigma) * ncb + off : J.zero;
}
HSD double mx_job_rhs(const MfmaJob& J, int rho) {
  const bool ok = rho / 6 < J.enter_limit && rho / 6 < J.zero_from;
  const double v = J.g[ok ? rho : 0];
  return ok ? v : 0.0;
}

/// Does the tile index t (16 ring positions) meet the six positions p .. p + 5 (p a multiple of six: they never wrap)?
__host__ __device__ constexpr bool mx_hit(int t, int p) { return 16 * t <= p + 5 && 16 * t + 15 >= p; }

// ================================ MFMA waves ================================
/// One block row of an MFMA wave in phase PH = it mod 16 (it >= 1): apply X_(it-1) = x, extract block row it + 1 -> rb, enter block row
/// it + 15 <- st. Ring positions: block row it at 6 PH, X_(it-1)'s own row at 6 PH - 6 = the positions the entering row takes.
///
/// WIDE (bands of 15, 16 control points): the ring's 16 slots are all in use — the entering block row it + 15 takes its positions in the very
/// iteration in which block row it + 1, whose band reaches it (and, at 16, block row it + 16, which enters an iteration later), leaves for the
/// panel. Those last one or two band blocks of a row are never updated before the row becomes the pivot row (no earlier row reaches both
/// members of the pair), so they bypass the accumulators: the loader writes them into rowbuf (k_band_factor_mx), and the extraction below
/// leaves the twelve positions of block rows it - 1 and it alone. No tile is ever outside the band: all 21 are updated in every phase.
template <int G, int PH, int TW, bool WIDE>
HSD void mx_tiles_step(mx_f64x4 (&acc)[TW], const double* x, double* rb, const double* st, int l15, int g4) {
  constexpr int W = kMxW, LDX = kMxLdx, HI = WIDE ? 96 : 84;
  constexpr int p_i = (6 * PH + W - 6) % W, p_row = (6 * PH + 6) % W, p_e = p_i;
  double xf[6][2], nxf[6][2];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const bool kk_ok = g == 0 || g4 < 2;  // k = 6, 7 of the second MFMA: zero rows
      const int pos = 16 * t + l15;
      const double v = x[(kk_ok ? g4 + 4 * g : 0) * LDX + pos];
      // offsets 0 .. 11 of the pivot row belong to block rows it - 1 (U_ii) and it (in the panel's hands): not part of the update
      xf[t][g] = (kk_ok && mx_sub(pos, p_i) >= 12) ? v : 0.0;
      nxf[t][g] = -xf[t][g];
    }
  mx_static_for<TW>([&](auto mc) {
    constexpr int m = decltype(mc)::value, q = kMxGroupTile[G][m], I = mx_tile_I(q), Jt = mx_tile_J(q);
    if constexpr (mx_index_active(I, p_i, HI) && mx_index_active(Jt, p_i, HI))
      acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(nxf[I][0], xf[Jt][0], acc[m], 0, 0, 0);
  });
  // program order: second MFMA of tile m, then the hand-overs of tile m - 1 (its result is ready by then)
  mx_static_for<TW + 1>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    if constexpr (m < TW) {
      constexpr int q = kMxGroupTile[G][m], I = mx_tile_I(q), Jt = mx_tile_J(q);
      if constexpr (mx_index_active(I, p_i, HI) && mx_index_active(Jt, p_i, HI))
        acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(nxf[I][1], xf[Jt][1], acc[m], 0, 0, 0);
    }
    if constexpr (m >= 1) {
      constexpr int mm = m - 1, q = kMxGroupTile[G][mm], I = mx_tile_I(q), Jt = mx_tile_J(q);
      // element (a, b) of register rr: a = 16 I + g4 + 4 rr, b = 16 Jt + l15
      // ---- block row it + 1 -> rowbuf: rb[k][b] = pair (p_row + k, b) ----
      if constexpr (mx_hit(I, p_row)) {  // as rows of the tile
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          constexpr int d = 16 * I - p_row;
          if (d + 4 * rr + 3 >= 0 && d + 4 * rr < 6) {  // (compile time: some g4 is in range)
            const int k = d + 4 * rr + g4;
            if (unsigned(k) < 6u && (!WIDE || mx_sub(16 * Jt + l15, p_i) >= 12)) rb[k * LDX + 16 * Jt + l15] = acc[mm][rr];
          }
        }
      }
      if constexpr (I != Jt && mx_hit(Jt, p_row)) {  // as columns (a diagonal tile holds both orders already)
        const int k = 16 * Jt - p_row + l15;
        if (unsigned(k) < 6u) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            if (!WIDE || mx_sub(16 * I + g4 + 4 * rr, p_i) >= 12) rb[k * LDX + 16 * I + g4 + 4 * rr] = acc[mm][rr];
        }
      }
      // ---- block row it + 15 <- stage: st[q][a] = pair (a, p_e + q) ----
      if constexpr (mx_hit(Jt, p_e)) {
        const int qe = 16 * Jt - p_e + l15;
        if (unsigned(qe) < 6u) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) acc[mm][rr] = st[qe * LDX + 16 * I + g4 + 4 * rr];
        }
      }
      if constexpr (mx_hit(I, p_e)) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          constexpr int d = 16 * I - p_e;
          if (d + 4 * rr + 3 >= 0 && d + 4 * rr < 6) {
            const int qe = d + 4 * rr + g4;
            if (unsigned(qe) < 6u) acc[mm][rr] = st[qe * LDX + 16 * Jt + l15];
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

template <int G, bool WIDE>
HSD void mx_tiles_wave(const Tables& T, const MfmaJob& J, double* smem, int l, int n_iter) {
  constexpr int TW = kMxGroupSize[G], W = kMxW, LDX = kMxLdx;
  const int bw = T.bw, ncb = 6 * bw, np = T.np, m_at = J.merge_at;
  const int l15 = l & 15, g4 = l >> 4;
  double* xring = smem + kMxX;
  double* rowbuf = smem + kMxR;
  const double* stage = smem + kMxS;
  const bool prof = prof_enabled(T.debug_flags, 16) && l == 0 && blockIdx.x == 0;
  long long* tlog = reinterpret_cast<long long*>(T.xpart);
  mx_f64x4 acc[TW];
  // prologue: the pairs among block rows 2 .. 15, ring position = matrix index (block rows 0 and 1 start in rowbuf)
  mx_static_for<TW>([&](auto mc) {
    constexpr int m = decltype(mc)::value, q = kMxGroupTile[G][m], I = mx_tile_I(q), Jt = mx_tile_J(q);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      // (no select on the loaded value: the wave passes the barriers of the prologue and of iteration 0 — the panel's — with its 28 loads
      //  in flight and waits for them in front of its first MFMA)
      const int a = 16 * I + g4 + 4 * rr, b = 16 * Jt + l15;
      acc[m][rr] = *((a >= 12 && b >= 12) ? mx_job_address(J, np, ncb, a, b) : J.zero);
    }
  });
  lds_barrier();  // P0
  int it = 0;
  while (it < n_iter) {
    // the phases up to and including the junction iteration (job 0), or to the end
    const int stop = (m_at >= it && m_at < n_iter) ? m_at + 1 : n_iter;
    while (it < stop) {
      mx_static_for<16>([&](auto phc) {
        constexpr int PH = decltype(phc)::value;
        if ((it & 15) == PH && it < stop) {
          if (prof && G == 0) tlog[8 * it + 0] = wall_clock64();
          if ((PH != 0 || it >= 1) && (it < J.n_steps || J.dump))  // (the last iteration is the storer's; job 1 applies its last row too)
            mx_tiles_step<G, PH, TW, WIDE>(acc, xring + ((it - 1) & 1) * 6 * LDX, rowbuf + ((it + 1) & 1) * 6 * LDX, stage + (it & 1) * 6 * LDX, l15, g4);
          if (prof && G < 3) tlog[8 * it + 2 + G] = wall_clock64();
          lds_barrier();
          ++it;
        }
      });
    }
    if (m_at >= 0 && it == m_at + 1 && m_at < n_iter) {  // ---- junction (job 0): add the other end's Schur correction of the middle block rows ----
      wait_for_partner(T);
      const int dm = 6 * (bw - 1), p_m = (6 * m_at) % W;
      const double* D = J.win;
      mx_static_for<TW>([&](auto mc) {
        constexpr int m = decltype(mc)::value, q = kMxGroupTile[G][m], I = mx_tile_I(q), Jt = mx_tile_J(q);
        double d[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int da = mx_sub(16 * I + g4 + 4 * rr, p_m), db = mx_sub(16 * Jt + l15, p_m);
          const bool ok = da >= 12 && db >= 12 && da < dm && db < dm;
          const double v = D[ok ? size_t(da) * (dm + 1) + db : 0];
          d[rr] = ok ? v : 0.0;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[m][rr] += d[rr];
      });
      lds_barrier();  // merge done
      lds_barrier();  // panel(m) done
    }
  }
  if (J.dump) {  // job 1: the pure correction of the middle block rows in job 0's coordinates (index reversal inside the middle block), both orders
    const int dm = 6 * (bw - 1), p_m = (6 * J.n_steps) % W;
    double* D = J.win;
    mx_static_for<TW>([&](auto mc) {
      constexpr int m = decltype(mc)::value, q = kMxGroupTile[G][m], I = mx_tile_I(q), Jt = mx_tile_J(q);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int da = mx_sub(16 * I + g4 + 4 * rr, p_m), db = mx_sub(16 * Jt + l15, p_m);
        if (da >= 12 && db >= 12 && da < dm && db < dm) {
          D[size_t(dm - 1 - da) * (dm + 1) + (dm - 1 - db)] = acc[m][rr];
          D[size_t(dm - 1 - db) * (dm + 1) + (dm - 1 - da)] = acc[m][rr];
        }
      }
    });
  }
}

// ================================ panel lanes (wave 3: ring columns 48 .. 95 + right-hand side; wave 4: 0 .. 47) ================================
#define MX_UIDX(a, c) ((a) * 6 - (a) * ((a)-1) / 2 + ((c) - (a)))

/// A panel lane: its LDS column (ring position; kMxW = the right-hand side lane; the diagonal lanes move with the pivot).
struct MxLane {
  int col;
  bool ring;    // owns a ring column: publishes it
  bool active;  // ring column or right-hand side: publishes
  bool diag;    // lane kMxDiagLane + c: column c of the pivot's diagonal block, computed for the broadcast only
};

/// v = row it (rowbuf) - X_(it-1),1' X_(it-1)[:, pos].
HSD void mx_lane_update(const MxLane& L, const double* row, const double* xp, int p_it, const double (&xc)[6], double (&v)[6]) {
  constexpr int LDX = kMxLdx;
#pragma unroll
  for (int a = 0; a < 6; ++a) v[a] = row[a * LDX + L.col];
  double B[6][6];  // block 1 of X_(it-1): its columns of block row it — six consecutive ring positions, 16-byte aligned
  const double* bp = xp + p_it;
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int a = 0; a < 6; a += 2) {
      const double2 t = *reinterpret_cast<const double2*>(bp + k * LDX + a);
      B[k][a] = t.x, B[k][a + 1] = t.y;
    }
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int k = 0; k < 6; ++k) v[a] = fma(-B[k][a], xc[k], v[a]);
}

/// (Round 6 tried the DPP scheme of k_dense_solve_mx's panel here — the block handed to every row of sixteen lanes through LDS, right-looking
///  Cholesky with v_fmac_f64_dpp, pivots scheduled by hand: 84 ordered instructions instead of 178 — and factor + solve took 0.52 us where this
///  takes 0.44: on a 6 x 6 block the chain of a pivot (scale, two wait states, DPP multiply-add, two wait states, DPP move, rsq, Halley) is what
///  counts, not the instruction count, and the compiler's schedule of the left-looking form below has the shorter one. Not kept.)
/// The updated diagonal block sits in the diagonal lanes (lane kMxDiagLane + c: column c). Every lane factors it redundantly: the 21 entries
/// are broadcast with v_readlane (wave-uniform values in SGPRs). U comes out with NEGATED off-diagonal entries (products of two of them are
/// unchanged, the column solves become plain multiply-adds), inv = 1 / diag. Returns the smallest pivot.
HSD double mx_lane_factor(const double (&v)[6], double (&U)[21], double (&inv)[6]) {
  {
    int pidx = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = a; c < 6; ++c)
        U[pidx++] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v[a]), kMxDiagLane + c), __builtin_amdgcn_readlane(__double2loint(v[a]), kMxDiagLane + c));
  }
  double dmin = 0.0;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double d = U[MX_UIDX(a, a)];
#pragma unroll
    for (int k = 0; k < a; ++k) d = fma(-U[MX_UIDX(k, a)], U[MX_UIDX(k, a)], d);
    // A non-positive pivot is not patched on this chain: it turns the rest of the factor into NaN / inf, `fail` is raised and the
    // step is rejected as invalid (k_decide also requires a finite model cost change).
    dmin = a == 0 ? d : fmin(dmin, d);
    // 1 / sqrt(d): hardware estimate (2^-24 relative) + one third-order step, e = 1 - d y^2, y (1 + e/2 + 3 e^2/8)
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    const double rs = fma(y * e, fma(0.375, e, 0.5), y);
    inv[a] = rs;
    const double nrs = -rs;
#pragma unroll
    for (int c = a + 1; c < 6; ++c) {
      double t = U[MX_UIDX(a, c)];
#pragma unroll
      for (int k = 0; k < a; ++k) t = fma(-U[MX_UIDX(k, a)], U[MX_UIDX(k, c)], t);
      U[MX_UIDX(a, c)] = t * nrs;
    }
  }
  return dmin;
}

/// x = U^-T v, published at the own ring position (zeros outside the trailing band: the MFMA waves update blindly).
HSD void mx_lane_solve(const MxLane& L, const double (&U)[21], const double (&inv)[6], const double (&v)[6], double* xo, int p_it, int ncb) {
  constexpr int LDX = kMxLdx;
  double x[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double t = v[a];
#pragma unroll
    for (int k = 0; k < a; ++k) t = fma(U[MX_UIDX(k, a)], x[k], t);
    x[a] = t * inv[a];
  }
  if (L.active) {
    const bool live = !L.ring || mx_sub(L.col, p_it) < ncb;
#pragma unroll
    for (int a = 0; a < 6; ++a) xo[a * LDX + L.col] = live ? x[a] : 0.0;
  }
}

/// Junction (job 0), lanes of both panel waves (t = 0 .. 127): block rows m, m + 1 sit in rowbuf.
HSD void mx_lane_merge(const MfmaJob& J, double* rowbuf, int t, int m_at, int p_m, int bw) {
  constexpr int LDX = kMxLdx, W = kMxW;
  const int dm = 6 * (bw - 1);
  const double* D = J.win;
  constexpr int RU = 10;  // 2 * 6 * (W + 1) <= 10 * 128
  double dv[RU];
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    const int e = t + 128 * u;
    const bool in = e < 12 * (W + 1);
    const int j = in ? e / (6 * (W + 1)) : 0, rem = in ? e % (6 * (W + 1)) : 0, k = rem / (W + 1), pos = rem % (W + 1);
    const int da = 6 * j + k, db = pos == W ? dm : mx_sub(pos, p_m);
    // (db >= 6 j: the upper part only. With 16 control points per band the positions of block row m ARE band positions of block row m + 1 —
    //  its pairs with block row m + 16 — and the mirrored entries D[m + 1][m] would be stored, and carried along, as a band block)
    const bool ok = in && (pos == W || (db < dm && db >= 6 * j));
    const double v = D[ok ? size_t(da) * (dm + 1) + db : 0];
    dv[u] = ok ? v : 0.0;
  }
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    const int e = t + 128 * u;
    if (e < 12 * (W + 1)) {
      const int j = e / (6 * (W + 1)), rem = e % (6 * (W + 1)), k = rem / (W + 1), pos = rem % (W + 1);
      rowbuf[((m_at + j) & 1) * 6 * LDX + k * LDX + pos] += dv[u];
    }
  }
}

/// Dump (job 1), panel lanes: block rows n (still to be updated by X_(n-1)) and n + 1 of the trailing window sit in rowbuf (the
/// right-hand sides of the others: storer). D is written in job 0's coordinates (index reversal inside the middle block), both orders.
HSD void mx_lane_dump(const MxLane& L, const MfmaJob& J, const double* rowbuf, const double* xp, int n, int p_n, int bw) {
  constexpr int LDX = kMxLdx, W = kMxW;
  const int dm = 6 * (bw - 1);
  double* D = J.win;
  if (!L.active) return;
  double xc[6], v[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) xc[k] = (!L.ring || mx_sub(L.col, p_n) + 6 < 6 * bw) ? xp[k * LDX + L.col] : 0.0;  // (see the panel: WIDE)
  mx_lane_update(L, rowbuf + (n & 1) * 6 * LDX, xp, p_n, xc, v);
  const double* row1 = rowbuf + ((n + 1) & 1) * 6 * LDX;
  if (!L.ring) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      D[size_t(dm - 1 - k) * (dm + 1) + dm] = v[k];
      D[size_t(dm - 1 - (6 + k)) * (dm + 1) + dm] = row1[k * LDX + W];
    }
    return;
  }
  const int c = mx_sub(L.col, p_n);
  if (c < dm) {  // column c of block row n
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      D[size_t(dm - 1 - k) * (dm + 1) + (dm - 1 - c)] = v[k];
      D[size_t(dm - 1 - c) * (dm + 1) + (dm - 1 - k)] = v[k];
    }
  }
  if (c >= 6 && c < dm) {  // its entries in block row n + 1
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double t = row1[k * LDX + L.col];
      D[size_t(dm - 1 - (6 + k)) * (dm + 1) + (dm - 1 - c)] = t;
      D[size_t(dm - 1 - c) * (dm + 1) + (dm - 1 - (6 + k))] = t;
    }
  }
}

template <bool WIDE, bool PUB = false>  // WIDE: bands of 15 and 16 control points (mx_tiles_step); PUB: publishes its progress (MfmaJob::progress:
                                         // a compile-time switch — as a run-time one it cost the instance without followers 6 us at configs[1])
__global__ void __launch_bounds__(kMxThreads) k_band_factor_mx(Tables T) {
  constexpr int W = kMxW, LDX = kMxLdx;
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const MfmaJob J = T.mj[blockIdx.x];
  const int tid = threadIdx.x, hw = tid >> 6, l = tid & 63;
  const int bw = T.bw, ncb = 6 * bw, np = T.np, n_steps = J.n_steps, m_at = J.merge_at;
  const int n_iter = n_steps + 1;  // the storer writes block row n_steps - 1 in the last one; job 1 applies it too: its window is handed over
  double* xring = smem + kMxX;
  double* rowbuf = smem + kMxR;
  double* stage = smem + kMxS;
  double* dinv = smem + kMxD;  // 2 x 8: 1 / diag(U_ii) of block row i in dinv[i & 1] (panel -> storer)
  const bool prof = prof_enabled(T.debug_flags, 16) && l == 0 && blockIdx.x == 0;  // phase timestamps -> hs_debug_read (tools/mx_phase_timing.py)
  long long* tlog = reinterpret_cast<long long*>(T.xpart);
  long long* plog = tlog + 8 * 1024;
  __shared__ int fail;
  if (tid == 0) fail = 0;

  // Waves are placed round robin on the four SIMDs: SIMD 0 = waves 0, 4 (MFMA groups 0, 1), SIMD 1 = 1, 5 (MFMA groups 2, 3), SIMD 2 = 2, 6
  // (panel, storer), SIMD 3 = 3, 7 (panel, loader); wave 8 (SIMD 0) inverts the diagonal blocks for the sweeps. The f64 MFMAs keep the fp64 pipes of SIMDs 0 and 1 busy for most of a step: the
  // storer took 1.3 us per block row next to them; a panel wave is latency bound and leaves issue slots.
  if (hw == 0 || hw == 1 || hw == 4 || hw == 5) {
    if (hw == 0) mx_tiles_wave<0, WIDE>(T, J, smem, l, n_iter);
    if (hw == 4) mx_tiles_wave<1, WIDE>(T, J, smem, l, n_iter);
    if (hw == 1) mx_tiles_wave<2, WIDE>(T, J, smem, l, n_iter);
    if (hw == 5) mx_tiles_wave<3, WIDE>(T, J, smem, l, n_iter);
  } else if (hw == 7) {  // ================================ loader ================================
    // lane l owns band columns t = l + 64 m of an entering block row (t < ncb; t == W: right-hand side), all six rows; what it stages is the
    // whole ring row: zeros outside the band
    double va[2][6], vb[2][6];
    auto fetch = [&](double (*v)[6], int e) {  // block row e: S(rho = 6 (e - bw + 1) + t, sigma = 6 e + q) with the job's zero rules
      // entries that enter as zeros are read from a zero in memory: a select on the loaded value would make the wave wait for the data
      const bool ok_e = e < J.enter_limit;
      const long long row0 = ok_e ? (long long)(np - 1 - 6 * e) * ncb : 0;  // row of sigma = 6 e; sigma + q: q rows earlier
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int t = l + 64 * m;
        const int rb = e - bw + 1 + int(unsigned(t) / 6u);
        const bool band = t < ncb, rhs = t == W;
        const bool ok = band && ok_e && 6 * (e - bw + 1) + t >= 0 && !(rb >= J.zero_from && e >= J.zero_from);
        const bool ok_g = rhs && ok_e && e < J.zero_from;
        const double* src = ok ? J.L + row0 + (ncb - 1 - t) : (ok_g ? J.g + 6 * e : J.zero);
        const long long stride = ok ? -(long long)ncb : (ok_g ? 1 : 0);
#pragma unroll
        for (int q = 0; q < 6; ++q) v[m][q] = src[q * stride];
      }
    };
    auto stage_write = [&](const double (*v)[6], double* dst, int e) {
      const int p_lo = (6 * (e + 16 - bw + 1)) % W;  // ring position of band offset 0: block row e - bw + 1
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int t = l + 64 * m;
        if (t <= W) {
          const int col = t == W ? W : mx_add(p_lo, t);
#pragma unroll
          for (int q = 0; q < 6; ++q) dst[q * LDX + col] = v[m][q];
        }
      }
    };
    fetch(va, 16), fetch(vb, 17);
    // prologue share of this wave: X_(-1) = 0
    for (int e = l; e < 6 * LDX; e += 64) xring[6 * LDX + e] = 0.0;
    lds_barrier();  // P0
    auto step = [&](double (*v)[6], int it) {  // v holds block row it + 16 (fetched two iterations ago): entered in iteration it + 1
      stage_write(v, stage + ((it + 1) & 1) * 6 * LDX, it + 16);
      if constexpr (WIDE) {
        // The last band blocks of block row it + 1 — its pairs with block rows it + 15 (entering in this iteration) and, at 16 control points,
        // it + 16 (staged above) — are not in the accumulators when the row leaves: straight into rowbuf, at the positions the MFMA waves skip.
        double* rb = rowbuf + ((it + 1) & 1) * 6 * LDX;
        const int p_it = (6 * it) % W, p_prev = (6 * it + W - 6) % W, p_next = (6 * it + 6) % W;
        if (bw == 16 && l < 6) {  // band offset t = l of block row it + 16 is position l of block row it + 1
#pragma unroll
          for (int q = 0; q < 6; ++q) rb[l * LDX + p_it + q] = v[0][q];
        }
        if (it >= 1 && l < 36) {  // (block row 15 is resident from the prologue on: rows 0 and 1 come complete)
          const int k = l / 6, q = l % 6;
          rb[k * LDX + p_prev + q] = stage[(it & 1) * 6 * LDX + q * LDX + p_next + k];
        }
      }
      fetch(v, it + 18);
      if (prof) tlog[8 * it + 5] = wall_clock64();
      lds_barrier();
      if (m_at >= 0 && it == m_at) {  // junction: merge, panel(m)
        lds_barrier();
        lds_barrier();
      }
    };
    int it = 0;
    for (; it + 1 < n_iter; it += 2) step(va, it), step(vb, it + 1);
    if (it < n_iter) step(va, it);
  } else if (hw == 8) {  // ================================ inverse wave ================================
    // A short dependent chain per block row with a whole iteration to run in: it sits next to the MFMA waves of SIMD 0.
    lds_barrier();  // P0
    int p_prev = W - 6;  // ring position of block row it - 1
    constexpr bool pub = PUB;  // (see the storer)
    for (int it = 0; it < n_iter; ++it) {
      // VMEM BUDGET (what the late publication rests on): this wave issues, per iteration, the SIX agent-scope stores of the inverted block below
      // (one instruction each, exec never empty: lanes 0..5) + the progress store — and no load. vmcnt counts instructions of this wave in issue
      // order, so "all but 12 outstanding" leaves at most the stores of the two previous iterations in flight. Anything that makes an iteration issue
      // FEWER than six VMEM instructions (a store moved under a divergent branch, two stores merged into one) breaks the inference without an
      // error: the GPU test test_two_ended_bordered_solve (bit identity against the sequential arrangement, 40 solves) is what would notice.
      if constexpr (pub) {  // >= 6 stores per iteration (+ this one): all but the last twelve are done -> the blocks of rows 0 .. it - 4 are in memory
        if (it >= 4) {
          wait_vmem_all_but<12>();
          if (l == 0) __hip_atomic_store(J.progress + kProgressStride, J.progress_base + unsigned(it - 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (it >= 1 && l < 6) {  // (six lanes: the return path of the LDS is what every wave queues for at the start of a step)
        // W = U_(ii)^-1, i = it - 1 (upper triangular, packed) for the sweeps: lane cw < 6 solves U w = e_cw. U_ii sits at the pivot's own
        // positions of X_i (upper part), 1 / diag comes from the panel (dinv). Entries below the diagonal go to the pad of the 24-double slot.
        const double* xp = xring + ((it - 1) & 1) * 6 * LDX + p_prev;
        const double* di = dinv + ((it - 1) & 1) * 8;
        double Ud[6][6], dv[6];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int k = 0; k < 6; k += 2) {
            const double2 t = *reinterpret_cast<const double2*>(xp + a * LDX + k);
            Ud[a][k] = t.x, Ud[a][k + 1] = t.y;
          }
#pragma unroll
        for (int a = 0; a < 6; a += 2) {
          const double2 t = *reinterpret_cast<const double2*>(di + a);
          dv[a] = t.x, dv[a + 1] = t.y;
        }
        const int cw = l;
        double w[6];
#pragma unroll
        for (int a = 5; a >= 0; --a) {
          double t = a == cw ? 1.0 : 0.0;
#pragma unroll
          for (int k = a + 1; k < 6; ++k) t = fma(-Ud[a][k], w[k], t);
          w[a] = t * dv[a];
        }
        double* dst = J.Ubk + size_t(it - 1) * 24;
        if constexpr (pub) {
#pragma unroll
          for (int a = 0; a < 6; ++a)
            __hip_atomic_store(dst + (a <= cw ? a * 6 - a * (a - 1) / 2 + (cw - a) : 21 + (a >> 1)), w[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
          for (int a = 0; a < 6; ++a) dst[a <= cw ? a * 6 - a * (a - 1) / 2 + (cw - a) : 21 + (a >> 1)] = w[a];
        }
      }
      p_prev = mx_add(p_prev, 6);
      lds_barrier();
      if (m_at >= 0 && it == m_at) {  // junction: merge, panel(m)
        lds_barrier();
        lds_barrier();
      }
    }
    if constexpr (pub) {
      wait_vmem();
      if (l == 0) __hip_atomic_store(J.progress + kProgressStride, J.progress_base + unsigned(n_steps), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (hw == 6) {  // ================================ storer ================================
    // What is not on the chain, one iteration after the panel published a block row: factor row and y -> HBM (the inverted diagonal block:
    // wave 8); right-hand side of the trailing rows (lane l < 48 <-> ring positions 2 l, 2 l + 1: 16-byte loads and
    // stores): g -= X' y, the entries of the block row that leaves the window -> rowbuf, those of the entering one <- stage.
    const bool has = l < 48;
    const int pos = has ? 2 * l : 0;
    double g[2];
    g[0] = has ? mx_job_rhs(J, pos) : 0.0, g[1] = has ? mx_job_rhs(J, pos + 1) : 0.0;
    lds_barrier();  // P0
    int p_prev = W - 6;  // ring position of block row it - 1
    // Somebody follows this job row by row (MfmaJob::progress: the forward sweep of the border columns on the side stream): the factor rows
    // are written with agent-scope stores (twelve of 8 bytes where the plain path has six of 16) and the count of complete rows is published late, behind a wait that leaves the stores of the
    // last iteration in flight (a wait for all of them would put the write-through latency on this wave in every iteration).
    constexpr bool pub = PUB;
    for (int it = 0; it < n_iter; ++it) {
      // VMEM BUDGET: per iteration TWELVE agent-scope 8-byte stores of the factor row (two per row a, exec never empty: the lanes of the band) + the
      // y store + the progress store, no load — "all but 24 outstanding" leaves at most two iterations' stores in flight (see the inverse wave).
      if constexpr (pub) {  // >= 12 stores per iteration (the factor row; + y, + this one): all but the last 24 are done -> everything iteration
                            // it - 3 and the ones before it stored, i.e. rows 0 .. it - 4, is in memory
        if (it >= 4) {
          wait_vmem_all_but<24>();
          if (l == 0) __hip_atomic_store(J.progress, J.progress_base + unsigned(it - 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      const double* xp = xring + ((it - 1) & 1) * 6 * LDX;  // X_(it-1) (zeros for it = 0)
      double* rb_next = rowbuf + ((it + 1) & 1) * 6 * LDX;
      const double* stg = stage + (it & 1) * 6 * LDX;
      double yv[6];
      double2 x[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) yv[k] = xp[k * LDX + W];
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = *reinterpret_cast<const double2*>(xp + k * LDX + pos);
      // right-hand side: block row it + 1 leaves with the update by X_(it-1) (the panel applies X_(it) itself); block row it + 15 enters
      // at the positions of block row it - 1
#pragma unroll
      for (int k = 0; k < 6; ++k) g[0] = fma(-x[k].x, yv[k], g[0]), g[1] = fma(-x[k].y, yv[k], g[1]);
      {
        const int c1 = mx_sub(pos, mx_add(p_prev, 12));  // (even: both positions of the lane are in the block row or neither)
        if (has && c1 < 6) rb_next[c1 * LDX + W] = g[0], rb_next[(c1 + 1) * LDX + W] = g[1];
        const int qe = mx_sub(pos, p_prev);
        const bool enter = has && it >= 1 && qe < 6;
        const double v0 = stg[(enter ? qe : 0) * LDX + W], v1 = stg[(enter ? qe + 1 : 0) * LDX + W];
        g[0] = enter ? v0 : g[0], g[1] = enter ? v1 : g[1];
      }
      if (it >= 1) {  // block row it - 1 of the factor
        const int i = it - 1;
        const int c = mx_sub(pos, p_prev);  // band offset (even)
        if (has && c < ncb) {
          double* dst = J.Ub + size_t(6 * i) * ncb + c;
          if constexpr (pub) {
#pragma unroll
            for (int a = 0; a < 6; ++a) {  // (8-byte atomics: there is no 16-byte one, and an inline-asm dwordx4 store is invisible to the
                                           //  compiler's hazard handling — a later write of its data registers corrupted rows on the device)
              __hip_atomic_store(dst + a * ncb, x[a].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(dst + a * ncb + 1, x[a].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          } else {
#pragma unroll
            for (int a = 0; a < 6; ++a) *reinterpret_cast<double2*>(dst + a * ncb) = x[a];
          }
        }
        if (l < 6) {
          double yl = yv[0];
#pragma unroll
          for (int k = 1; k < 6; ++k) yl = l == k ? yv[k] : yl;
          J.ybuf[6 * i + l] = yl;
        }
      }
      if (prof) tlog[8 * it + 1] = wall_clock64();
      p_prev = mx_add(p_prev, 6);
      lds_barrier();
      if (m_at >= 0 && it == m_at) {  // junction (job 0): right-hand sides of the middle block rows m + 2 ..
        wait_for_partner(T);
        const int dm = 6 * (bw - 1);
        const int da = mx_sub(pos, p_prev);  // (p_prev = position of block row m by now)
        const bool ok = has && da >= 12 && da < dm;
        const double v0 = J.win[ok ? size_t(da) * (dm + 1) + dm : 0], v1 = J.win[ok ? size_t(da + 1) * (dm + 1) + dm : 0];
        g[0] += ok ? v0 : 0.0, g[1] += ok ? v1 : 0.0;
        lds_barrier();  // merge done
        lds_barrier();  // panel(m) done
      }
    }
    if constexpr (pub) {
      wait_vmem();
      if (l == 0) __hip_atomic_store(J.progress, J.progress_base + unsigned(n_steps), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (J.dump) {  // right-hand sides of block rows n + 2 .. of the trailing window
      const int dm = 6 * (bw - 1), c = mx_sub(pos, (6 * n_steps) % W);
      if (has && c >= 12 && c < dm) J.win[size_t(dm - 1 - c) * (dm + 1) + dm] = g[0], J.win[size_t(dm - 2 - c) * (dm + 1) + dm] = g[1];
    }
  } else {  // ================================ panel waves (hw 2 and 3: each alone on its SIMD) ================================
    const bool chain = hw == 3;  // wave 3: the upper half of the ring, the right-hand side, 1 / diag for the storer, `fail`
    const bool cprof = prof_enabled(T.debug_flags, 16) && chain && l == 0;  // coarse phases of this job -> tlog[8 (590 + 4 job) + ..]
    long long* clog = tlog + 8 * (590 + 4 * blockIdx.x);
    if (cprof) clog[0] = wall_clock64();
    MxLane L;
    L.ring = l < 48;
    L.active = L.ring || (chain && l == 48);
    L.diag = l >= kMxDiagLane && l < kMxDiagLane + 6;
    L.col = L.ring ? (chain ? 48 + l : l) : ((chain && l == 48) ? W : 0);
    // prologue: block rows 0 and 1 -> rowbuf (ring position = matrix index)
    {
      const int t = (chain ? 0 : 64) + l;  // 0 .. 127
      constexpr int RU = 10;               // 2 * 6 * (W + 1) <= 10 * 128
      double pv[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int e = t + 128 * u;
        const bool in = e < 12 * (W + 1);
        const int j = in ? e / (6 * (W + 1)) : 0, rem = in ? e % (6 * (W + 1)) : 0, k = rem / (W + 1), pos = rem % (W + 1);
        // (position -> matrix index: the band of block row 1 wraps around the ring when it is 16 control points wide)
        pv[u] = pos == W ? mx_job_rhs(J, 6 * j + k) : mx_job_value(J, np, ncb, 6 * j + k, pos >= 6 * j ? pos : pos + W);
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int e = t + 128 * u;
        if (e < 12 * (W + 1)) {
          const int j = e / (6 * (W + 1)), rem = e % (6 * (W + 1)), k = rem / (W + 1), pos = rem % (W + 1);
          rowbuf[j * 6 * LDX + k * LDX + pos] = pv[u];
        }
      }
    }
    lds_barrier();  // P0
    // panel of block row `it`: X_(it) from rowbuf[it & 1] and X_(it-1) (zeros for it = 0)
    auto panel = [&](int it, int p_it) {
      const double* row = rowbuf + (it & 1) * 6 * LDX;
      const double* xp = xring + ((it - 1) & 1) * 6 * LDX;
      double* xo = xring + (it & 1) * 6 * LDX;
      double U[21], inv[6], v[6], xc[6];
      if (L.diag) L.col = p_it + (l - kMxDiagLane);
#pragma unroll
      for (int k = 0; k < 6; ++k) xc[k] = xp[k * LDX + L.col];
      if constexpr (WIDE) {  // 16 control points: the last band block of this row sits at the positions of block row it - 1, where X_(it-1) holds U
        const bool in_prev = !L.ring || mx_sub(L.col, p_it) + 6 < ncb;
#pragma unroll
        for (int k = 0; k < 6; ++k) xc[k] = in_prev ? xc[k] : 0.0;
      }
      mx_lane_update(L, row, xp, p_it, xc, v);
      if (prof) plog[8 * it + (chain ? 1 : 5)] = wall_clock64();
      const double dmin = mx_lane_factor(v, U, inv);
      if (prof) plog[8 * it + (chain ? 2 : 6)] = wall_clock64();
      mx_lane_solve(L, U, inv, v, xo, p_it, ncb);
      if (chain) {
#pragma unroll
        for (int a = 0; a < 6; a += 2) *reinterpret_cast<double2*>(&dinv[(it & 1) * 8 + a]) = make_double2(inv[a], inv[a + 1]);  // every lane, same value
        if (l == 0 && !(dmin > 0.0)) fail = 1;
      }
    };
    if (cprof) clog[1] = wall_clock64();  // prologue done
    int p_it = 0;
    for (int it = 0; it < n_iter; ++it) {
      const bool junction = m_at >= 0 && it == m_at;  // no look-ahead across the junction: row m changes there
      if (prof) plog[8 * it + (chain ? 0 : 4)] = wall_clock64();
      if (it < n_steps && !junction) panel(it, p_it);
      if (prof) plog[8 * it + (chain ? 3 : 7)] = wall_clock64();
      lds_barrier();
      if (junction) {
        if (cprof) clog[2] = wall_clock64();  // junction reached
        wait_for_partner(T);
        if (cprof) clog[3] = wall_clock64();  // partner arrived
        mx_lane_merge(J, rowbuf, (chain ? 0 : 64) + l, m_at, p_it, bw);
        lds_barrier();  // merge done
        if (cprof) clog[4] = wall_clock64();
        panel(it, p_it);
        lds_barrier();
        if (cprof) clog[5] = wall_clock64();  // X_m published
      }
      p_it = mx_add(p_it, 6);
    }
    if (cprof) clog[6] = wall_clock64();  // last block row done
    if (J.dump) mx_lane_dump(L, J, rowbuf, xring + ((n_steps - 1) & 1) * 6 * LDX, n_steps, (6 * n_steps) % W, bw);
  }
  __syncthreads();
  if (tid == 0 && fail) st->chol_failed = 1;  // (consumed by the decision, decide_step)
  if (J.dump) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (every wave: its stores have reached the L2; the ONE agent-scope release — an L2 write-back on this part — is lane 0's below)
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      __hip_atomic_store(T.join_flag, T.join_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (prof_enabled(T.debug_flags, 16)) tlog[8 * (590 + 4 * blockIdx.x) + 7] = wall_clock64();  // window handed over
    }
  }
  // Iteration bookkeeping of a directly assembled system (Tables::bookkeep, factor_bookkeep): a wave of the FAR end, behind its hand-over —
  // that workgroup is done ~17 us before the near end, in whose prologue the ~2 us of this reduction used to sit. Nothing in this launch
  // reads what it writes (cost, gradient norm, the record, `done`: the sweep and everything behind it do).
  if (T.bookkeep && blockIdx.x == 1 && hw == 3) factor_bookkeep(T, l);
}
#undef MX_UIDX

}  // namespace hs
