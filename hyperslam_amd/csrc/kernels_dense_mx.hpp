// kernels_dense_mx.hpp — the whole solve of a SMALL reduced system in one launch, trailing matrix in the accumulators of the f64 matrix cores
// (part of kernels.hpp; included once by capi.hip through it).
//
// What it is for. The sliding window HyperSLAM actually runs (max_window 3.0 s at 0.1 s separation,
// /root/reference/internal/hyper/optimizers/abstract.cpp:26-28, settings.yaml:145-148) leaves ~33 free control points whose landmark tracks
// are as long as the window: the "band" of the reduced system IS the matrix (~200 x 200), plus, with an IMU, a border of 6 n_bias + 2
// unknowns (~45). Up to round 5 such a window paid, per LM iteration, k_dense_factor (2.1 us per block row, 72 us), then — bordered — the
// forward sweep of the border columns, the border Schur complement, its dense Cholesky and y' = y - Z x_b in four more launches, then
// k_band_backward: six dependent launches, ~150 us, a third of the iteration. Replaces what CHOLMOD does behind
// /root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:46-48 (SPARSE_NORMAL_CHOLESKY) for these windows.
//
// Here the bordered system [S_pp S_pb; S_pb' S_bb] x = [g_p; g_b] of N = 6 n_free + n_b <= 256 unknowns is ONE dense Cholesky solve in ONE
// workgroup of twelve waves: the border is simply the last columns of the matrix — Z = U^-T S_pb, the border Schur complement and its
// factor are what the elimination produces on the way (nothing in the algebra distinguishes them).
//
//   data      16 x 16 tiles (I, J) of the padded matrix (identity on the padding) live in MFMA accumulators for the whole factorisation:
//             register r of lane l = entry (4 r + (l >> 4), l & 15) of the tile. Twelve waves, three per SIMD; tile rows in three classes,
//             tile columns in four (DxRows): every step's tile row and trailing matrix are spread over all SIMDs.
//   step k    (U1) tile row k + 1 gets the update of step k first and goes to LDS                                              --- barrier ---
//             (P)  panel of step k + 1 on the four panel waves (one per SIMD, raised issue priority), ONE column of the row per lane, the
//                  16 columns of the diagonal tile redundantly in every row of sixteen lanes: right-looking Cholesky of the diagonal
//                  block fused with the forward substitution of every column, pivots and multipliers by DPP row broadcast inside the
//                  multiply-add (v_fmac_f64_dpp) and inside the reciprocal square root (v_rsq_f64_dpp), hand-scheduled (dx_panel_pivots).
//                  Sixteen extra columns start as the identity and come out as the rows of W_k = U_kk^-1 (for the backward sweep only);
//                  the right-hand side is one more column. X(k + 1, :) goes back to LDS (operand of the update) and to memory (factor).
//             (U2) next to the panel: tile (I, J) -= X(k, I)' X(k, J) for k + 1 < I: four v_mfma_f64_16x16x4_f64 per tile, A operand
//                  (negated) shared by the tiles of a row                                                                     --- barrier ---
//   sweep     U x = y in 16-row blocks from the last one: x_K = W_K pend_K, then every pending row above subtracts U(:, K) x_K — one barrier
//             per block; the columns of U come back from memory (L2), requested one block ahead.
//   outputs   step_p / delta_p (zero on the decoupled rows of the leading constant control points), x_b / delta_b, the two sums of the
//             model cost change: what k_band_backward wrote (kernels_factor.hpp), so that the update kernels do not change.
// Bring-up: tests/emul/factor_harness.cpp variant 6 (the kernel source on the CPU against numpy), then tests/test_gpu_edge_cases.py.
#pragma once
#include <utility>

#include "dpp_f64.hpp"
#include "kernels_factor.hpp"

namespace hs {

constexpr int kDxThreads = 768;   // twelve waves, three per SIMD
constexpr int kDxTiles = 16;      // N <= 256
constexpr int kDxLd = 272;        // LDS row stride of a panel row (doubles)
// LDS (doubles): panel rows, double buffered | W_k rows (16 x 256) | g | y | x | pend_K (2 x 16) | block sums
constexpr int kDxOffW = 2 * 16 * kDxLd, kDxOffG = kDxOffW + 16 * 256, kDxOffY = kDxOffG + kDxLd, kDxOffX = kDxOffY + kDxLd, kDxOffP = kDxOffX + kDxLd,
              kDxOffRed = kDxOffP + 32, kDxLdsDoubles = kDxOffRed + 32;
typedef double dx_f64x4 __attribute__((vector_size(32)));

/// Systems the kernel holds: n_free free block rows + nb border unknowns within 16 tiles of 16.
/// (+ 1: the right-hand side is column n_dense of the padded matrix)
__host__ __device__ constexpr bool dense_mx_fits(int n_free, int nb) { return n_free >= 1 && 6 * n_free + nb + 1 <= 16 * kDxTiles; }

/// Tiles of a wave. Tile COLUMNS are dealt to four column classes b in a zigzag, col(b, jq) = {b, 7 - b, 8 + b, 15 - b}: one column of every
/// group of four; tile ROWS to three row classes A — rows 7 11 13 15 for the four panel waves (which also hold 64 registers of panel columns and their temporaries:
/// with the ten tiles of rows 3 6 9 13 next to them the compiler spilled 67 - 87 registers around every panel, scratch stores and reloads ON the chain),
/// rows 0 2 4 6 9 12 and 1 3 5 8 10 14 for the others (17 tiles are what fits next to the operands: with 18 the addresses went to scratch). A wave (A, b) holds the tiles (I, J) of its rows and columns with J in a LATER OR THE
/// SAME group of four as I: 7 / 17 / 16 tiles; which of a row's tiles in its own group lie above the diagonal depends on b: a tile below the
/// diagonal is loaded and written out like the others and never read by anybody (17 % of the tiles; the update skips them on a scalar test).
/// That makes the code of a row class the same for every column class: b only enters addresses. (A variant per (A, b) with exactly the tiles
/// I <= J meant a four-way dispatch around every phase of every step, and the compiler merged the accumulator arrays of the variants behind each:
/// 600 - 3 000 spilled registers.) A SIMD holds one wave of each row class, all of one column class: 40 tiles per SIMD.
constexpr int kDxRow[3][6] = {{7, 11, 13, 15, 99, 99}, {0, 2, 4, 6, 9, 12}, {1, 3, 5, 8, 10, 14}};  // tile rows of a row class
constexpr int kDxFirst[3][7] = {{0, 3, 5, 6, 7, 7, 7}, {0, 4, 8, 11, 14, 16, 17}, {0, 4, 8, 11, 13, 15, 16}};  // slot of a row's first tile (4 - row / 4 tiles per row)
template <int A>
struct DxRows {
  // (tables, not loops: the slot of a tile has to fold to a constant wherever it is used — an index the compiler cannot fold puts the accumulator
  //  array into scratch memory, and a recursive constexpr function is not even inlined on the device)
  static constexpr int n_rows = A == 0 ? 4 : 6;
  static constexpr int row(int iq) { return kDxRow[A][iq]; }
  static constexpr int jq_min(int iq) { return kDxRow[A][iq] / 4; }  // first column group of the row's tiles
  static constexpr int count = kDxFirst[A][n_rows];
  static constexpr int slot(int iq, int jq) { return kDxFirst[A][iq] + jq - kDxRow[A][iq] / 4; }
};
HSD int dx_col(int b, int jq) { return 8 * (jq >> 1) + ((jq & 1) ? 7 - b : b); }

/// acc += a b on the f64 matrix core. (The builtin, not inline assembly with the accumulator as a read-write operand — tried, to keep the compiler
/// from writing the result elsewhere: the hazard recogniser does not look inside an asm statement, and four dependent v_mfma_f64_16x16x4_f64 on one
/// accumulator need the wait states the ISA lists between them. Wrong results on the device, right ones in the emulation.)
HSD void dx_mfma(dx_f64x4& acc, double a, double b) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); }

/// An assembler comment that carries a compile-time number: two blocks of code that contain different ones are different code. (The bodies of
/// the phases below for two tile rows with equally many tiles differ in the accumulators' indices only; the compiler merged such pairs into one
/// block with the index as a run-time value — and an accumulator array indexed at run time lives in scratch memory.)
template <int N>
HSD void dx_keep_apart() {
#if !defined(HS_EMULATED_DEVICE)
  asm volatile("; dx %0" ::"n"(N));
#endif
}


/// The three phases that touch the accumulators (registers: every index a compile-time constant). lane: the lane's place in a register row of a
/// tile, (l >> 4) rows down and (l & 15) columns in, in units of the row stride; cb[jq]: first column of the wave's tile column jq.
template <int A>
struct DxWave {
  using R = DxRows<A>;
  typedef dx_f64x4 Acc[R::count];
  /// load: the wave's tiles from the dense copy of the scaled, damped system the finalisation kernels wrote (Tables::dense: row-major, leading
  /// dimension 256, both triangles, identity on the padding) — a scalar row base per register row, the lane's offset, the tile column.
  /// (Round 6 first loaded from the band / border tables directly: a region test and an address select per entry, unrolled, was 60 000 lines of ISA.)
  static HSD void load(const double* D, unsigned lane, const int (&cb)[4], Acc& acc) {
#pragma unroll
    for (int iq = 0; iq < R::n_rows; ++iq) {
      const double* row = D + 16 * R::row(iq) * kDenseLd;  // (wave uniform)
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {  // (constant trip counts: a bound that depends on the outer loop's variable is not unrolled)
        if (jq < R::jq_min(iq)) continue;
        const double* src = row + lane + unsigned(cb[jq]);
        // (the tile as ONE value: written register by register the array stayed an object in memory — scratch)
        acc[R::slot(iq, jq)] = dx_f64x4{src[0], src[4 * kDenseLd], src[8 * kDenseLd], src[12 * kDenseLd]};
      }
    }
  }
  /// (1) tile row k -> LDS. (The rows as template arguments — a fold over an index sequence instead of an unrolled loop — so that each row's code
  /// can carry its own number, dx_keep_apart.)
  template <int IQ>
  static HSD void extract_row(int k, unsigned lane, const int (&cb)[4], double* xb, const Acc& acc) {
    if (R::row(IQ) != k) return;
    dx_keep_apart<R::row(IQ)>();
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      if (jq < R::jq_min(IQ)) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[4 * r * kDxLd + lane + unsigned(cb[jq])] = acc[R::slot(IQ, jq)][r];
    }
  }
  template <int... IQ>
  static HSD void extract_rows(std::integer_sequence<int, IQ...>, int k, unsigned lane, const int (&cb)[4], double* xb, const Acc& acc) {
    (extract_row<IQ>(k, lane, cb, xb, acc), ...);
  }
  static HSD void extract(int k, unsigned lane, const int (&cb)[4], double* xb, const Acc& acc) {
    extract_rows(std::make_integer_sequence<int, R::n_rows>{}, k, lane, cb, xb, acc);
  }
  /// (3) trailing update: tile (I, J) -= X(k, I)' X(k, J) for the rows lo <= I < hi. The A operand (negated) is shared by the tiles of a row.
  template <int IQ>
  static HSD void update_row(int k, int lo, int hi, int nt_cols, unsigned lane, const int (&cb)[4], const double* xb, Acc& acc) {
    constexpr int I = R::row(IQ);
    if (I < lo || I >= hi) return;
    dx_keep_apart<100 + I>();
    double aop[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) aop[s] = -xb[4 * s * kDxLd + lane + 16 * I];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      if (jq < R::jq_min(IQ)) continue;
      // (a scalar test per tile: below the diagonal — nobody reads the tile — or in the identity padding behind tile column nt_cols, where
      //  X(k, J) is exactly zero; together 17 - 40 % of the matrix-core work of the first steps)
      if (cb[jq] < 16 * I || cb[jq] >= 16 * nt_cols) continue;
      double bop[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bop[s] = xb[4 * s * kDxLd + lane + unsigned(cb[jq])];
#pragma unroll
      for (int s = 0; s < 4; ++s) dx_mfma(acc[R::slot(IQ, jq)], aop[s], bop[s]);
    }
  }
  template <int... IQ>
  static HSD void update_rows(std::integer_sequence<int, IQ...>, int k, int lo, int hi, int nt_cols, unsigned lane, const int (&cb)[4], const double* xb, Acc& acc) {
    (update_row<IQ>(k, lo, hi, nt_cols, lane, cb, xb, acc), ...);
  }
  /// the tile rows lo <= I < hi with X(k, :)
  static HSD void update(int k, int lo, int hi, int nt_cols, unsigned lane, const int (&cb)[4], const double* xb, Acc& acc) {
    update_rows(std::make_integer_sequence<int, R::n_rows>{}, k, lo, hi, nt_cols, lane, cb, xb, acc);
  }
};


/// Panel of step k for one wave: lane l holds column l & 15 of the diagonal tile (a copy in every row of sixteen lanes: the source of the DPP
/// broadcasts) AND one column of its own — a trailing column of the tile row (the right-hand side is one of them: column n_dense of the
/// padded system) or a column of the identity (-> a row of W_k = U_kk^-1). Right-looking Cholesky of the diagonal block fused with the forward
/// substitution of every column; plain substitution, no explicit inverse on the forward path. X(k, :) goes back to LDS (operand of the update)
/// and to memory (the factor by columns, for the sweep).
///
/// The chain is the sixteen pivots: entry P + 1 of the diagonal must have its update of pivot P before 1 / sqrt of it can start, and that
/// estimate + Newton-Halley step is six dependent instructions (row broadcast, rsq, y^2, e = 1 - d y^2,
/// y e and 0.5 + 0.375 e, y + y e (...)). In program order — all 2 (15 - P) multiply-adds of pivot P, then the chain of P + 1 — a pivot
/// was ~300 cycles, more than half of them the wave waiting for its own previous instruction. Here pivot P issues the ONE multiply-add pivot
/// P + 1 waits for first, then deals the remaining ones into the gaps of that chain (dx_fm: multiply-add n of pivot P; five gaps).
template <int P, int N>
HSD void dx_fm(double (&ad)[16], double (&at)[16]) {  // n = 0: ad[P + 1], 1: at[P + 1], 2: ad[P + 2], ...
  if constexpr (N < 2 * (15 - P)) {
    constexpr int R = P + 1 + N / 2;
    if constexpr (N % 2 == 0)
      dx_fnma_bcast<R>(ad[R], ad[P], ad[P]);  // u_pr = entry P of the diagonal tile's column R, from the lane of the row that holds it
    else
      dx_fnma_bcast<R>(at[R], ad[P], at[P]);
  }
}
template <int P, int LO, int HI>
HSD void dx_fm_range(double (&ad)[16], double (&at)[16]) {
  if constexpr (LO < HI) {
    dx_fm<P, LO>(ad, at);
    dx_fm_range<P, LO + 1, HI>(ad, at);
  }
}
template <int P>
HSD void dx_panel_pivots(double (&ad)[16], double (&at)[16], double rinv, double k375, bool& fail) {
  fail |= !(rinv < 1e150);  // (a pivot that is not positive: estimate NaN or infinite — the caller zeroes the step of a failed factorisation)
  dx_scale2(ad[P], at[P], rinv);
  if constexpr (P < 15) {
    constexpr int n = 2 * (15 - P) - 1, g = (n + 4) / 5;  // the multiply-adds behind the first, per gap
    dx_fm<P, 0>(ad, at);
    const double d = dx_row_bcast<P + 1>(ad[P + 1]);
    dx_fm<P, 1>(ad, at);
    const double y0 = dx_rsq(d);
    dx_fm_range<P, 2, 1 + g>(ad, at);
    const double y2 = dx_mul(y0, y0);
    dx_fm_range<P, 1 + g, 1 + 2 * g>(ad, at);
    const double e = dx_one_minus(d, y2);
    dx_fm_range<P, 1 + 2 * g, 1 + 3 * g>(ad, at);
    const double ye = dx_mul(y0, e), c = dx_half_plus(e, k375);
    dx_fm_range<P, 1 + 3 * g, 1 + 4 * g>(ad, at);
    const double rn = dx_fma(ye, c, y0);
    dx_fm_range<P, 1 + 4 * g, 1 + 5 * g>(ad, at);
    dx_panel_pivots<P + 1>(ad, at, rn, k375, fail);
  }
}
HSD bool dx_panel(int k, int n_tr, int w, int l, double* xb, double* wk, double* ut, bool prof = false, long long* tlog = nullptr) {
  const int idx = 64 * w + l;
  const bool trailing = idx < n_tr;
  const int unit = (idx >= n_tr && idx < n_tr + 16) ? idx - n_tr : -1;
  const int col = 16 * (k + 1) + (trailing ? idx : 0);
  double ad[16], at[16];
  // (the panel is the chain of the step and shares its SIMD with two waves that feed the matrix core: it goes first whenever it can issue)
  __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    ad[r] = xb[r * kDxLd + 16 * k + (l & 15)];
    at[r] = xb[r * kDxLd + col];  // (idle and identity lanes read a column of the buffer they do not use ...
    dx_pin(at[r]);                //  ... unconditionally: left to itself the compiler wraps each of the sixteen loads into a branch on `trailing`)
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) at[r] = trailing ? at[r] : (r == unit ? 1.0 : 0.0);
  if (prof) tlog[8 * k + 4] = wall_clock64();
  bool fail = false;
  {
    const double d = dx_row_bcast<0>(ad[0]), y0 = dx_rsq(d), y2 = dx_mul(y0, y0), k375 = 0.375, e = dx_one_minus(d, y2);
    dx_panel_pivots<0>(ad, at, dx_fma(dx_mul(y0, e), dx_half_plus(e, k375), y0), k375, fail);
  }
  if (prof) tlog[8 * k + 5] = wall_clock64();
  if (trailing) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xb[r * kDxLd + col] = at[r];
  }  // (the factor's copy in memory is written from LDS by the waves that do not have a panel: dx_store_factor_row)
  if (unit >= 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) wk[(16 * k + r) * 16 + unit] = at[r];  // (W_k by columns: the sweep's lanes read entry (i, c) at 16 c + i, no two lanes of a row of sixteen in one bank)
  }
  if (w == 0 && l < 16) {  // the diagonal tile's columns too (entries above the diagonal): the right-hand side column may be one of them
    double* dst = ut + size_t(16 * k + l) * kDenseLd + 16 * k;
#pragma unroll
    for (int r = 0; r < 16; r += 2) *reinterpret_cast<double2*>(dst + r) = make_double2(ad[r], ad[r + 1]);
  }
  __builtin_amdgcn_s_setprio(0);
  return fail;
}

/// X(k, :) from its LDS buffer to memory (the factor by columns, for the sweep), by the eight waves without a panel while the panel of the next
/// step runs: thread (column, half) moves eight rows — 64 contiguous bytes, a column's two halves in neighbouring lanes. (From the panel's
/// lanes — sixteen rows of one column each, 2 KB apart — the stores were 0.1 - 0.7 us at the end of every panel, on the chain.)
HSD void dx_store_factor_row(int k, int n_pad, int w, int l, const double* xb, double* ut) {
  const int t = 64 * (w - 4) + l, c = t >> 1, h = 8 * (t & 1);
  if (c >= n_pad - 16 * (k + 1)) return;
  const int col = 16 * (k + 1) + c;
  double v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = xb[(h + r) * kDxLd + col];
  double* dst = ut + size_t(col) * kDenseLd + 16 * k + h;
#pragma unroll
  for (int r = 0; r < 8; r += 2) *reinterpret_cast<double2*>(dst + r) = make_double2(v[r], v[r + 1]);
}

/// Factorisation loop of a wave of row class A (0: the panel waves w < 4, 1, 2: the others; the accumulator arrays differ in length, and the panel's
/// registers must not be live next to 24 tiles): load, then per step extract / panel / update with the two barriers. Returns the panel's verdict.
template <int A>
HSD bool dx_factor(const double* D, int nt, int n_pad, int b, int w, int l, double* smem, double* ut, bool prof, long long* tlog) {
  using W = DxWave<A>;
  double* wk = smem + kDxOffW;
  const int cb[4] = {16 * dx_col(b, 0), 16 * dx_col(b, 1), 16 * dx_col(b, 2), 16 * dx_col(b, 3)};
  typename W::Acc acc;
  W::load(D, unsigned((l >> 4) * kDenseLd + (l & 15)), cb, acc);
  const unsigned lane = unsigned((l >> 4) * kDxLd + (l & 15));
  bool fail = false;
  if (prof) tlog[8 * 20 + 3] = wall_clock64();
  // Look-ahead: the panel of step k + 1 runs (on the four panel waves) while everybody else applies X(k, :) to the tile rows behind k + 1 — the
  // panel is ~3 us of one wave per SIMD, the update up to 3.5 us of matrix core in the first steps. Per step and wave:
  //   U1  tile row k + 1 -= X(k, k + 1)' X(k, :), written to the OTHER panel buffer                                       --- barrier ---
  //   P   panel waves: panel of step k + 1 (in place in that buffer)     U2  tile rows > k + 1 -= X(k, .)' X(k, :)  (panel waves: after the panel;
  //       their rows are rows 7 11 13 15: few tiles)                                           --- barrier ---
  // X(k, :) is read from its buffer through U2; the buffer is overwritten in U1 of the next step, behind the barrier.
  W::extract(0, lane, cb, smem, acc);
  lds_barrier();
  if (prof) tlog[0] = wall_clock64();
  if (A == 0 && 64 * w < n_pad - 16 + 16) fail |= dx_panel(0, n_pad - 16, w, l, smem, wk, ut, prof, tlog);
  if (prof) tlog[1] = wall_clock64();
  lds_barrier();
  for (int k = 0; k < nt; ++k) {
    const double* xb = smem + (k & 1) * 16 * kDxLd;      // X(k, :)
    double* xn = smem + ((k + 1) & 1) * 16 * kDxLd;      // tile row k + 1 -> X(k + 1, :)
    if (prof) tlog[8 * k + 2] = wall_clock64();
    if (k + 1 < nt) {
      W::update(k, k + 1, k + 2, nt, lane, cb, xb, acc);
      W::extract(k + 1, lane, cb, xn, acc);
    }
    lds_barrier();
    if (prof) tlog[8 * (k + 1) + 0] = wall_clock64();
    const int n_tr = n_pad - 16 * (k + 2);  // trailing columns of tile row k + 1 (+ 16 columns of the identity: W_(k+1))
    if (A == 0 && k + 1 < nt && 64 * w < n_tr + 16) fail |= dx_panel(k + 1, n_tr, w, l, xn, wk, ut, prof, tlog);
    if (prof) tlog[8 * (k + 1) + 1] = wall_clock64();
    if (A != 0) dx_store_factor_row(k, n_pad, w, l, xb, ut);
    W::update(k, k + 2, nt, nt, lane, cb, xb, acc);
    if (prof) tlog[8 * k + 3] = wall_clock64();
    lds_barrier();
  }
  return fail;
}

/// Sweep: column block K of U for row rho (scalar base + lane offset per column; zero for the rows at or below the block).
HSD void dx_sweep_fetch(const double* ut, int rho, int K, double (&u)[16]) {
#pragma unroll
  for (int c = 0; c < 16; ++c) u[c] = (K >= 0 && rho < 16 * K) ? (ut + (16 * K + c) * kDenseLd)[unsigned(rho)] : 0.0;
}
/// Sweep, block K: x_K = W_K pend_K in lanes 0 .. 15 of every wave (W_K upper triangular: the entries left of the diagonal came out as exact
/// zeros), then every pending row above subtracts U(:, K) x_K. One barrier. (Functions with the register sets as parameters: as lambdas that
/// capture the arrays by reference they put them into scratch memory.)
template <int C>
HSD void dx_sweep_apply(double& s0, double& s1, double xr, const double (&u)[16]) {
  if constexpr (C < 16) {
    dx_fmac_bcast<C>(s0, xr, u[C]);  // (x_K is replicated in every row of sixteen lanes: lane l holds entry l & 15)
    dx_fmac_bcast<C + 1>(s1, xr, u[C + 1]);
    dx_sweep_apply<C + 2>(s0, s1, xr, u);
  }
}
HSD void dx_sweep_block(int K, int rho, int w, int l, const double* ut, const double* wk, double* pk, double* xv, double& pend, const double (&u)[16],
                        double (&u_next)[16], long long* slog = nullptr) {
  if (K < 0) return;
  if (slog) slog[0] = wall_clock64();
  if (rho >= 16 * K && rho < 16 * K + 16) pk[(K & 1) * 16 + (rho - 16 * K)] = pend;
  dx_sweep_fetch(ut, rho, K - 3, u_next);
  lds_barrier();
  if (slog) slog[1] = wall_clock64();
  const double* wr = wk + 16 * 16 * K + (l & 15);
  const double* pr = pk + (K & 1) * 16;
  double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
#pragma unroll
  for (int c = 0; c < 16; c += 4)
    x0 = fma(wr[16 * c], pr[c], x0), x1 = fma(wr[16 * c + 16], pr[c + 1], x1), x2 = fma(wr[16 * c + 32], pr[c + 2], x2), x3 = fma(wr[16 * c + 48], pr[c + 3], x3);
  double xr = (x0 + x1) + (x2 + x3);
  if (w == 0 && l < 16) xv[16 * K + l] = xr;
  if (slog) slog[2] = wall_clock64();
  double s0 = 0.0, s1 = 0.0;
#if !defined(HS_EMULATED_DEVICE)
  // (xr was written by the vector instruction before; the DPP reads below carry no wait states of their own. xr as an operand: the statement
  //  stays between the addition that writes it and the multiply-adds that read it)
  asm volatile("s_nop 4" : "+v"(xr));
#endif
  dx_sweep_apply<0>(s0, s1, xr, u);
  pend -= s0 + s1;
  if (slog) slog[3] = wall_clock64();
}

/// Border mode (Tables::dense_border): the part of the dense copy that k_border_schur does not write — identity on the padding, the huge
/// diagonal entry behind the right-hand side, zeros between — once per structure (launch_factor).
__global__ void __launch_bounds__(kBlock) k_dense_border_init(double* D, int nb) {
  const int n_pad = 16 * ((nb + 1 + 15) / 16);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_pad * n_pad; e += gridDim.x * blockDim.x) {
    const int i = e / n_pad, j = e % n_pad;
    D[size_t(i) * kDenseLd + j] = i != j ? 0.0 : (i == nb ? 1e300 : 1.0);
  }
}

/// One workgroup. f0: leading block rows of constant control points (decoupled, solution zero: the chain starts behind them).
/// ut: 256 x 256 doubles of scratch (the factor by columns, for the sweep).
__global__ void __launch_bounds__(kDxThreads) k_dense_solve_mx(Tables T, int f0, double* ut) {
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  // (the wave index as a SCALAR: with threadIdx.x >> 6 in a vector register the dispatch to the wave's tile phases is a divergent branch for the
  //  compiler, which then merges the accumulator arrays of all variants lane by lane — 3 000 spilled registers)
  const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), i16 = l & 15;
  const int n_pose = T.np - 6 * f0, n_dense = n_pose + T.nb, nt = (n_dense + 1 + 15) / 16, n_pad = 16 * nt;
  double* xv = smem + kDxOffX;
  double* wk = smem + kDxOffW;
  const bool prof = prof_enabled(T.debug_flags, 16) && tid == 0;  // phase stamps (profiling builds, tools/dense_mx_phase_timing.py)
  long long* tlog = reinterpret_cast<long long*>(T.xpart) + 8 * 300;
  if (prof) tlog[-1] = wall_clock64();
  // wave (a, b) = (w / 4, w mod 4): SIMD w mod 4 holds one wave of each row class, all of column class b
  bool fail = false;
  if (tid == 0) smem[kDxOffRed + 16] = 0.0;  // (ordered before the panel waves' verdicts by the barriers of the factorisation)
  if (w < 4)
    fail = dx_factor<0>(T.dense, nt, n_pad, w, w, l, smem, ut, prof, tlog);
  else if (w < 8)
    dx_factor<1>(T.dense, nt, n_pad, w - 4, w, l, smem, ut, false, tlog);
  else
    dx_factor<2>(T.dense, nt, n_pad, w - 8, w, l, smem, ut, false, tlog);
  if (fail) smem[kDxOffRed + 16] = 1.0;  // (a panel wave may have skipped the panel that failed: the verdict goes through LDS)
  if (prof) tlog[8 * 20 + 0] = wall_clock64();
  // ---- backward sweep U x = y, lane = row: the first four waves (one per SIMD) hold the 256 rows; the others have nothing left to do and
  //      leave — a wave that has ended does not count at a barrier, and while they stayed (16 predicated loads and a barrier per block, on the
  //      SIMDs of the row waves) a step of the sweep took longer ----
  wait_vmem();  // this wave's factor rows have left
  // (WORKGROUP scope: writer and readers of the factor are waves of this workgroup, one CU, one L1. An agent-scope fence — __threadfence() — writes the
  //  XCD's L2 back: ~2 us here, from twelve waves)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  lds_barrier();
  // Iteration bookkeeping of a directly assembled system (Tables::bookkeep, factor_bookkeep: no k_finalize_reduced launch in front of this kernel):
  // a wave that would leave here — nobody waits for its verdict (a termination test that fires makes every later kernel exit on `done`).
  if (T.bookkeep && w == 11) factor_bookkeep(T, l);
  if (w >= 4) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  double* pk = smem + kDxOffP;
  const int rho = tid;  // row of the padded system (tid < 256)
  // y = U^-T g is column n_dense of the factor: the right-hand side rode along as a column of the padded matrix (the dense copy carries g there
  // and a huge diagonal entry behind it; its own unknown gets a zero right-hand side, i.e. stays zero)
  double pend = rho < n_dense ? (ut + n_dense * kDenseLd)[unsigned(rho)] : 0.0;
  // Column block K of U for this row comes back from memory (L2; written by the panel): requested three blocks ahead — with one block ahead
  // a step of the sweep was 2.3 us of load latency. Four register sets of sixteen, renamed by unrolling (the accumulators are dead here).
  double ua[16], ub[16], uc[16], ud[16];
  dx_sweep_fetch(ut, rho, nt - 1, ua), dx_sweep_fetch(ut, rho, nt - 2, ub), dx_sweep_fetch(ut, rho, nt - 3, uc);
  for (int K = nt - 1; K >= 0; K -= 4) {
    long long* slog = (prof && nt - 1 - K < 8) ? tlog + 8 * (21 + nt - 1 - K) : nullptr;  // (profiling builds: the first eight blocks, rows 21 .. 28)
    dx_sweep_block(K, rho, w, l, ut, wk, pk, xv, pend, ua, ud, slog);
    dx_sweep_block(K - 1, rho, w, l, ut, wk, pk, xv, pend, ub, ua, slog ? slog + 8 : nullptr);
    dx_sweep_block(K - 2, rho, w, l, ut, wk, pk, xv, pend, uc, ub, slog ? slog + 16 : nullptr);
    dx_sweep_block(K - 3, rho, w, l, ut, wk, pk, xv, pend, ud, uc, slog ? slog + 24 : nullptr);
  }
  lds_barrier();
  if (prof) tlog[8 * 20 + 1] = wall_clock64();
  // ---- outputs: step = -x, delta = scale o step, the two sums of the model cost change (k_band_backward's epilogue) ----
  // A pivot that was not positive: 1 / sqrt of it is NaN or infinite and so is everything behind it — the step of a failed factorisation is zero
  // (decide_step rejects it on chol_failed, the radius shrinks: Ceres' LINEAR_SOLVER_FAILURE branch).
  fail = smem[kDxOffRed + 16] != 0.0;
  if (T.dense_border) {  // the border Schur complement of a two-ended bordered system: x_b for k_border_apply and the sweeps, nothing else
    if (rho < n_dense) T.xb[rho] = fail ? 0.0 : xv[rho];
    if (tid == 0 && fail) st->chol_failed = 1;
    return;
  }
  double gd = 0.0, dd = 0.0;
  for (int r = tid; r < 6 * f0; r += 256) T.step_p[r] = 0.0, T.delta_p[r] = 0.0;
  if (rho < n_dense) {
    const double step = fail ? 0.0 : -xv[rho];
    if (rho < n_pose) {
      const int r = 6 * f0 + rho;
      T.step_p[r] = step, T.delta_p[r] = T.scale_p[r] * step;
      gd = T.g_full[r] * step, dd = T.D2p[r] * step * step;
    } else {
      const int b = rho - n_pose;
      T.xb[b] = -step, T.delta_b[b] = T.scale_b[b] * step;
      gd = T.gb_s[b] * step, dd = T.D2b[b] * step * step;
    }
  }
  double* red = smem + kDxOffRed;  // (sums over the four row waves, in wave order: block_sum counts on the whole workgroup)
  gd = wave_sum(gd), dd = wave_sum(dd);
  if (l == 0) red[w] = gd, red[8 + w] = dd;
  lds_barrier();
  if (tid == 0) {
    gd = ((red[0] + red[1]) + red[2]) + red[3], dd = ((red[8] + red[9]) + red[10]) + red[11];
    st->g_dot_step_pose = gd, st->d2_step2_pose = dd;
    if (fail) st->chol_failed = 1;  // (consumed and cleared by decide_step, kernels_update.hpp)
  }
  if (prof) tlog[8 * 20 + 2] = wall_clock64();
}


}  // namespace hs
