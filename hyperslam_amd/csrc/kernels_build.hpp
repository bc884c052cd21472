// kernels_build.hpp — fused normal-equation build of the visual factors: linearisation, landmark elimination and the landmark-group-major
// Gram accumulation  J_p'J_p - Yh Yh'  in ONE pass, records in LDS only (part of kernels.hpp; included once by capi.hip through it).
//
// Replaces, for short feature tracks (bw (bw + 1) / 2 <= 256 window tiles), the sequence
//   k_linearize_visual (records -> HBM) -> k_landmark (records -> H_ll, b_l, W_l -> Y-hat) -> k_group_gram (-Yh Yh') + k_seg_gram (J_p'J_p)
// i.e. what Ceres does between CostFunction::Evaluate and its J'J / Schur assembly behind
// /root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:278. A residual's state block always lies inside the band window of its
// landmark — the 6 bw x 6 bw window anchored at the first control point the landmark touches — so BOTH Gram terms can be accumulated
// landmark-group-major by the workgroup that linearised the residual; no segment-major pass, no record ever leaves the CU.
//
// Work unit = CHUNK: a run of consecutive device landmarks that share their first control point cf (device order sorts landmarks by cf),
// at most L_max landmarks (W / Y-hat rows in LDS) and about R residuals (one pass; a landmark with more residuals takes several passes).
// Per workgroup (256 lanes):
//   1  lane t linearises residual t of the pass into a COMPACT record in LDS (factors.hpp: 8 + 7K doubles, the translation columns
//      -B_j A are rebuilt by the consumers)
//   2  lane <-> (landmark, row of W): W_l += J_p' J_l over the landmark's records;  9 more lanes per landmark: H_ll, b_l
//   3  lane <-> (record stream s, band tile (rb, cb), cb - rb < K): P += J_p' J_p of the records whose segment covers both blocks —
//      the pass's records are counting-sorted by segment (stable, ballot based: fixed summation order), so a tile's records are one run
//   4  lane <-> (landmark, row): damped 3 x 3 Cholesky (redundant per lane), Y-hat row -> LDS and HBM (k_backsub_retract needs it)
//   5  lane <-> (landmark stream, window tile (rb, cb)): Q -= Yh Yh', q -= Yh yh  (as k_group_gram)
//   6  streams combined through LDS in index order; the chunk's partial [tiles of P + Q | -Yh yh | J_p'r | diag J_p'J_p] -> HBM, summed
//      over the chunks of the overlapping groups by k_assemble in a fixed order: bit-reproducible, no floating-point atomics.
// HBM traffic per residual: 32 B of inputs + its share of Y-hat and of the chunk partial; the 448-byte record (k = 4) is never written.
#pragma once
#include "kernels_common.hpp"

namespace hs {

template <int K>
constexpr int build_rec_stride() { return compact_record<K>() + 2; }  // LDS record stride (16-byte aligned, bank-spread)

/// Band tiles of the chunk window that J_p'J_p touches: (rb, d = cb - rb), d < K, rb + d < bw; index = tiles of smaller d first.
HSD int band_tile_count(int bw, int k) { return k * bw - k * (k - 1) / 2; }
HSD int band_tile_index(int rb, int d, int bw) { return d * bw - d * (d - 1) / 2 + rb; }

/// LDS layout of k_build_visual (doubles unless noted) — the host sizes the launch with the same function.
struct BuildLds {
  int rec, wy, cps, hb, yh, slo, ints, total_doubles;
};
__host__ __device__ inline BuildLds build_lds_layout(int K, int bw, int R, int Lmax) {
  const int cs = 8 + 7 * K + 2, nband = K * bw - K * (K - 1) / 2, ns = kBlock / nband, ntile = bw * (bw + 1) / 2;
  const bool two = ntile <= kBlock / 2;
  BuildLds o;
  int off = 0;
  o.rec = off;  // records of the pass | per-stream P tiles [ns][nband][42] at the end
  off += (R * cs > ns * nband * 42 ? R * cs : ns * nband * 42);
  o.wy = off;   // W rows, then Y-hat rows [Lmax][6 bw][3] | stream 1's Q tiles at the end
  {
    int n = Lmax * 6 * bw * 3;
    if (two && ntile * 42 > n) n = ntile * 42;
    off += n;
  }
  o.cps = off, off += 8 * bw;     // the window's control points
  o.hb = off, off += Lmax * 10;   // H_ll (6), b_l (3) per landmark
  o.yh = off, off += Lmax * 4;    // y-hat per landmark
  o.slo = off, off += Lmax * 4;   // Jacobi scaling of the landmark (previous iterations) + its constant flag
  off += off & 1;
  o.ints = off;                   // int tables (two per double): lp[Lmax + 1] ncp[Lmax] yoff[Lmax] segoff[R] sorted[R] seg_start[bw + 2] wave_cnt[4][bw]
  off += (3 * Lmax + 1 + 2 * R + (bw + 2) + 4 * bw + 1) / 2 + 1;
  o.total_doubles = off;
  return o;
}

template <int K>
__global__ void __launch_bounds__(kBlock) k_build_visual(Tables T, int R, int Lmax, int robustify) {
  HS_DYNAMIC_LDS(smem);
  __shared__ double red[kBlock / 64];
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  DevState* st = T.st;
  if (st->done) return;
  if (w >= T.n_chunk) {  // padding workgroups of the visual section of the cost-partial table
    if (tid == 0) T.cost_part[w] = 0.0;
    return;
  }
  constexpr int CS = build_rec_stride<K>();
  const int bw = T.bw, R6 = 6 * bw, ntile = bw * (bw + 1) / 2, nband = band_tile_count(bw, K), NS = kBlock / nband, nseg = bw - K + 1;
  const BuildLds lay = build_lds_layout(K, bw, R, Lmax);
  double* recs = smem + lay.rec;
  double* Wy = smem + lay.wy;
  double* Hb = smem + lay.hb;
  double* yhs = smem + lay.yh;
  double* slo = smem + lay.slo;
  int* lp = reinterpret_cast<int*>(smem + lay.ints);
  int* l_ncp = lp + Lmax + 1;
  int* l_yoff = l_ncp + Lmax;
  int* segoff = l_yoff + Lmax;
  int* sorted = segoff + R;
  int* seg_start = sorted + R;       // nseg + 1
  int* wave_cnt = seg_start + bw + 2;  // [4][bw]

  // Deferred commit (DevState::spec & 2 or 4): the candidate accepted by the previous iteration is still only in the candidate buffers
  const bool pend = (st->spec == 2 || st->spec == 4) && st->accepted;
  const double* cp_src = pend ? T.cp_cand : T.cp;
  const double* lm_src = pend ? T.lm_cand : T.lm;
  const bool fresh = !st->scaling_ready;
  const double radius = st->radius;

  const int lo = T.ch_ptr[w], hi = T.ch_ptr[w + 1], nl = hi - lo;
  const int cf = T.lm_cfirst[lo];
  const int q0 = T.lm_ptr[lo], nres = T.lm_ptr[hi] - q0;
  // ---- chunk tables + the window's control points ----
  double* cps_l = smem + lay.cps;
  {
    const int ncp_w = min(bw, T.sp.n_cp - cf);
    const double2* s2 = reinterpret_cast<const double2*>(cp_src + 8 * cf);
    for (int e = tid; e < 4 * ncp_w; e += kBlock) reinterpret_cast<double2*>(cps_l)[e] = s2[e];
    if (tid <= nl) lp[tid] = T.lm_ptr[lo + tid] - q0;
    if (tid < nl) {
      const int dl = lo + tid;
      l_ncp[tid] = T.lm_ncp[dl], l_yoff[tid] = T.lm_yoff[dl];
      slo[4 * tid] = fresh ? 1.0 : T.lm_scale[3 * dl], slo[4 * tid + 1] = fresh ? 1.0 : T.lm_scale[3 * dl + 1];
      slo[4 * tid + 2] = fresh ? 1.0 : T.lm_scale[3 * dl + 2], slo[4 * tid + 3] = T.lm_const[dl] ? 0.0 : 1.0;
    }
  }
  __syncthreads();
  const double* cps_v = cps_l - 8 * cf;  // indexed by absolute control point

  // ---- lane roles of the J_p'J_p accumulation (phase 3) ----
  const bool p_lane = tid < NS * nband;
  const int p_s = p_lane ? tid / nband : 0, p_tb = p_lane ? tid % nband : 0;
  int p_d = 0, p_rb = p_tb;
  while (p_rb >= bw - p_d) p_rb -= bw - p_d, ++p_d;  // tiles of offset d: bw - d
  double pacc[36], pg[6];
#pragma unroll
  for (int e = 0; e < 36; ++e) pacc[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 6; ++e) pg[e] = 0.0;

  double cost = 0.0;
  const int n_task_w = nl * R6, n_task = n_task_w + 9 * nl;
  for (int p0 = 0; p0 < nres; p0 += R) {
    const int n_p = min(R, nres - p0);
    // ---- 1: linearise ----
    int my_o = -1;
    if (tid < n_p) {
      int first;
      cost += visual_linearize_compact<K>(T, cps_v, q0 + p0 + tid, robustify != 0, lm_src, recs + tid * CS, &first);
      my_o = first - cf;
      segoff[tid] = my_o;
    }
    // stable counting sort of the pass's record slots by segment offset: rank inside the wave from ballots, wave totals through LDS
    int my_rank = 0;
    for (int o = 0; o < nseg; ++o) {
      const unsigned long long m = __ballot(my_o == o);
      if (my_o == o) my_rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wave_cnt[wave * bw + o] = __popcll(m);
    }
    __syncthreads();  // records, segoff, wave_cnt visible; (pass > 0) the previous pass's consumers are done — see the barrier at the loop end
    if (tid <= nseg) {  // seg_start[o] = records with a smaller segment offset
      int s = 0;
      for (int o = 0; o < tid; ++o) s += wave_cnt[o] + wave_cnt[bw + o] + wave_cnt[2 * bw + o] + wave_cnt[3 * bw + o];
      seg_start[tid] = s;
    }
    __syncthreads();
    if (my_o >= 0) {
      int pos = seg_start[my_o] + my_rank;
      for (int ww = 0; ww < wave; ++ww) pos += wave_cnt[ww * bw + my_o];
      sorted[pos] = tid;
    }
    // (phase 2 below reads the records and segoff only; `sorted` is first read behind the next barrier)
    // ---- 2: W_l rows (lane <-> landmark, row), H_ll / b_l entries (lane <-> landmark, entry) ----
    for (int tau = tid; tau < n_task; tau += kBlock) {
      if (tau < n_task_w) {
        const int l = tau / R6, rho = tau - l * R6;
        if (rho >= 6 * l_ncp[l]) continue;
        const int jb = rho / 6, c = rho - 6 * jb;
        const int t_lo = max(lp[l], p0) - p0, t_hi = min(lp[l + 1], p0 + R) - p0;
        double* wr = Wy + (size_t(l) * R6 + rho) * 3;
        double w0 = p0 ? wr[0] : 0.0, w1 = p0 ? wr[1] : 0.0, w2 = p0 ? wr[2] : 0.0;
        for (int t = t_lo; t < t_hi; ++t) {
          const int jj = jb - segoff[t];
          if (jj < 0 || jj >= K) continue;
          const double* rec = recs + t * CS;
          double j0, j1;
          if (c < 3) {
            j0 = rec[8 + K + 3 * jj + c], j1 = rec[8 + 4 * K + 3 * jj + c];
          } else {
            const double nb = -rec[8 + jj];
            j0 = nb * rec[2 + c - 3], j1 = nb * rec[5 + c - 3];
          }
          w0 = fma(j0, rec[2], fma(j1, rec[5], w0));
          w1 = fma(j0, rec[3], fma(j1, rec[6], w1));
          w2 = fma(j0, rec[4], fma(j1, rec[7], w2));
        }
        wr[0] = w0, wr[1] = w1, wr[2] = w2;
      } else {
        const int tl = tau - n_task_w, l = tl / 9, e = tl - 9 * l;
        const int t_lo = max(lp[l], p0) - p0, t_hi = min(lp[l + 1], p0 + R) - p0;
        // e: 0..5 = H00 H01 H02 H11 H12 H22, 6..8 = b0 b1 b2  (x = row index of the first factor, y = second factor / residual)
        const int x = e < 3 ? 0 : (e < 5 ? 1 : (e < 6 ? 2 : e - 6));
        const int y = e < 3 ? e : (e < 5 ? e - 2 : 2);
        double acc = p0 ? Hb[10 * l + e] : 0.0;
        for (int t = t_lo; t < t_hi; ++t) {
          const double* rec = recs + t * CS;
          if (e < 6)
            acc = fma(rec[2 + x], rec[2 + y], fma(rec[5 + x], rec[5 + y], acc));
          else
            acc = fma(rec[2 + x], rec[0], fma(rec[5 + x], rec[1], acc));
        }
        Hb[10 * l + e] = acc;
      }
    }
    __syncthreads();  // `sorted` complete
    // ---- 3: J_p'J_p band tiles ----
    if (p_lane) {
      const int o_lo = max(0, p_rb + p_d - K + 1), o_hi = min(p_rb, nseg - 1);
      if (o_lo <= o_hi) {
        const int i_hi = seg_start[o_hi + 1];
        for (int idx = seg_start[o_lo] + p_s; idx < i_hi; idx += NS) {
          const int t = sorted[idx];
          const double* rec = recs + t * CS;
          const int o = segoff[t], a = p_rb - o, b = a + p_d;
          const double na = -rec[8 + a], nb = -rec[8 + b];
          double ja[2][6], jb[2][6];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const double Ai = rec[2 + 3 * i + c];
              ja[i][c] = rec[8 + K + 3 * K * i + 3 * a + c], ja[i][3 + c] = na * Ai;
              jb[i][c] = rec[8 + K + 3 * K * i + 3 * b + c], jb[i][3 + c] = nb * Ai;
            }
          }
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) pacc[6 * r + c] = fma(ja[0][r], jb[0][c], fma(ja[1][r], jb[1][c], pacc[6 * r + c]));
          if (p_d == 0) {
            const double r0 = rec[0], r1 = rec[1];
#pragma unroll
            for (int r = 0; r < 6; ++r) pg[r] = fma(ja[0][r], r0, fma(ja[1][r], r1, pg[r]));
          }
        }
      }
    }
    __syncthreads();  // the next pass overwrites the records / the stream tiles below alias them
  }
  // per-stream P tiles -> LDS (aliases the records: everybody is past the barrier above)
  double* Ps = recs;  // [NS][nband][42]
  if (p_lane) {
    double* dst = Ps + (size_t(p_s) * nband + p_tb) * 42;
#pragma unroll
    for (int e = 0; e < 36; e += 2) *reinterpret_cast<double2*>(dst + e) = make_double2(pacc[e], pacc[e + 1]);
#pragma unroll
    for (int e = 0; e < 6; e += 2) *reinterpret_cast<double2*>(dst + 36 + e) = make_double2(pg[e], pg[e + 1]);
  }
  // ---- 4: landmark elimination: V = S_l H_ll S_l + D_l^2 = L L', Y-hat = W S_l L^-T (landmark_finish of k_landmark, per row lane) ----
  const double inv_radius = 1.0 / radius;
  for (int tau = tid; tau < n_task_w; tau += kBlock) {
    const int l = tau / R6, rho = tau - l * R6;
    if (rho >= 6 * l_ncp[l]) continue;
    const int dl = lo + l;
    const double lmf = slo[4 * l + 3];  // 0: constant landmark (J_l = 0)
    const double* h = Hb + 10 * l;
    const double h0 = lmf * h[0], h1 = lmf * h[1], h2 = lmf * h[2], h3 = lmf * h[3], h4 = lmf * h[4], h5 = lmf * h[5];
    const double b0 = lmf * h[6], b1 = lmf * h[7], b2 = lmf * h[8];
    double sl0, sl1, sl2;
    if (fresh)
      sl0 = 1.0 / (1.0 + sqrt(h0)), sl1 = 1.0 / (1.0 + sqrt(h3)), sl2 = 1.0 / (1.0 + sqrt(h5));
    else
      sl0 = slo[4 * l], sl1 = slo[4 * l + 1], sl2 = slo[4 * l + 2];
    double v00 = sl0 * sl0 * h0, v01 = sl0 * sl1 * h1, v02 = sl0 * sl2 * h2;
    double v11 = sl1 * sl1 * h3, v12 = sl1 * sl2 * h4, v22 = sl2 * sl2 * h5;
    const double d0 = fmin(fmax(v00, 1e-6), 1e32) * inv_radius, d1 = fmin(fmax(v11, 1e-6), 1e32) * inv_radius, d2 = fmin(fmax(v22, 1e-6), 1e32) * inv_radius;
    v00 += d0, v11 += d1, v22 += d2;
    auto rsqrt_refined = [](double d) {  // hardware estimate + one third-order correction: full double accuracy, no divide / sqrt sequence
      const double y = __builtin_amdgcn_rsq(d);
      const double e = fma(-d * y, y, 1.0);
      return fma(y * e, fma(0.375, e, 0.5), y);
    };
    const double i00 = rsqrt_refined(v00), l00 = v00 * i00, l10 = v01 * i00, l20 = v02 * i00;
    const double p11 = v11 - l10 * l10, i11 = rsqrt_refined(p11), l11 = p11 * i11, l21 = (v12 - l20 * l10) * i11;
    const double p22 = v22 - l20 * l20 - l21 * l21, i22 = rsqrt_refined(p22), l22 = p22 * i22;
    const bool active = lmf != 0.0;
    double* wr = Wy + (size_t(l) * R6 + rho) * 3;
    const double w0 = lmf * wr[0] * sl0, w1 = lmf * wr[1] * sl1, w2 = lmf * wr[2] * sl2;
    double a0 = w0 * i00, a1 = (w1 - a0 * l10) * i11, a2 = (w2 - a0 * l20 - a1 * l21) * i22;
    if (!active) a0 = a1 = a2 = 0.0;
    wr[0] = a0, wr[1] = a1, wr[2] = a2;
    double* Y = T.Y + l_yoff[l] + 3 * rho;
    Y[0] = a0, Y[1] = a1, Y[2] = a2;
    if (rho == 0) {
      const double sb0 = sl0 * b0, sb1 = sl1 * b1, sb2 = sl2 * b2;
      const double y0 = sb0 * i00, y1 = (sb1 - l10 * y0) * i11, y2 = (sb2 - l20 * y0 - l21 * y1) * i22;
      double* L = T.lm_L + 6 * dl;
      L[0] = l00, L[1] = l10, L[2] = l11, L[3] = l20, L[4] = l21, L[5] = l22;
      const double yy0 = active ? y0 : 0.0, yy1 = active ? y1 : 0.0, yy2 = active ? y2 : 0.0;
      T.lm_yhat[3 * dl] = yy0, T.lm_yhat[3 * dl + 1] = yy1, T.lm_yhat[3 * dl + 2] = yy2;
      yhs[4 * l] = yy0, yhs[4 * l + 1] = yy1, yhs[4 * l + 2] = yy2;
      T.lm_sb[3 * dl] = sb0, T.lm_sb[3 * dl + 1] = sb1, T.lm_sb[3 * dl + 2] = sb2;
      T.lm_D2[3 * dl] = d0, T.lm_D2[3 * dl + 1] = d1, T.lm_D2[3 * dl + 2] = d2;
      T.lm_gmax[dl] = active ? fmax(fabs(b0), fmax(fabs(b1), fabs(b2))) : 0.0;
      if (fresh) T.lm_scale[3 * dl] = sl0, T.lm_scale[3 * dl + 1] = sl1, T.lm_scale[3 * dl + 2] = sl2;
    }
  }
  __syncthreads();
  // ---- 5: Q = - sum_l Yh_l Yh_l' over the window tiles, q = - sum_l Yh_l yh_l (diagonal tiles) ----
  const bool two = ntile <= kBlock / 2;
  const int q_s = two ? tid / (kBlock / 2) : 0, q_ns = two ? 2 : 1, q_t = two ? tid % (kBlock / 2) : tid;
  const bool q_ok = q_t < ntile;
  int q_rb = 0, q_cb = 0;
  {
    int rem = q_ok ? q_t : 0;
    while (rem >= bw - q_rb) rem -= bw - q_rb, ++q_rb;  // row rb holds bw - rb tiles
    q_cb = q_rb + rem;
  }
  double acc[36], qacc[6];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 6; ++e) qacc[e] = 0.0;
  if (q_ok) {
    const bool diag = q_rb == q_cb;
    for (int l = q_s; l < nl; l += q_ns) {
      if (q_cb >= l_ncp[l]) continue;
      const double* Yb = Wy + size_t(l) * R6 * 3;
      double B[18];
#pragma unroll
      for (int e = 0; e < 18; e += 2) {
        const double2 vb = *reinterpret_cast<const double2*>(Yb + 18 * q_cb + e);
        B[e] = vb.x, B[e + 1] = vb.y;
      }
      const double y0 = yhs[4 * l], y1 = yhs[4 * l + 1], y2 = yhs[4 * l + 2];
#pragma unroll
      for (int rp = 0; rp < 3; ++rp) {
        double A[6];
#pragma unroll
        for (int e = 0; e < 6; e += 2) {
          const double2 va = *reinterpret_cast<const double2*>(Yb + 18 * q_rb + 6 * rp + e);
          A[e] = va.x, A[e + 1] = va.y;
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int r = 2 * rp + rr;
#pragma unroll
          for (int c = 0; c < 6; ++c)
            acc[6 * r + c] = fma(-A[3 * rr + 2], B[3 * c + 2], fma(-A[3 * rr + 1], B[3 * c + 1], fma(-A[3 * rr], B[3 * c], acc[6 * r + c])));
          if (diag) qacc[r] = fma(-A[3 * rr + 2], y2, fma(-A[3 * rr + 1], y1, fma(-A[3 * rr], y0, qacc[r])));
        }
      }
    }
  }
  __syncthreads();  // everybody is done with the Y-hat rows: stream 1 hands its tiles over through the same area
  double* xch = Wy;
  if (two && q_s == 1 && q_ok) {
#pragma unroll
    for (int e = 0; e < 36; e += 2) *reinterpret_cast<double2*>(xch + q_t * 42 + e) = make_double2(acc[e], acc[e + 1]);
#pragma unroll
    for (int e = 0; e < 6; e += 2) *reinterpret_cast<double2*>(xch + q_t * 42 + 36 + e) = make_double2(qacc[e], qacc[e + 1]);
  }
  __syncthreads();
  // ---- 6: combine (fixed order: Q stream 0 + stream 1, then the P streams in index order) and write the chunk partial ----
  double* G = T.grpQ + size_t(w) * (size_t(ntile) * 36 + 3 * R6);
  if (q_s == 0 && q_ok) {
    if (two) {
#pragma unroll
      for (int e = 0; e < 36; ++e) acc[e] += xch[q_t * 42 + e];
#pragma unroll
      for (int e = 0; e < 6; ++e) qacc[e] += xch[q_t * 42 + 36 + e];
    }
    const int d = q_cb - q_rb;
    if (d < K) {
      const int bi = band_tile_index(q_rb, d, bw);
      double pt[36], gp[6];
#pragma unroll
      for (int e = 0; e < 36; ++e) pt[e] = 0.0;
#pragma unroll
      for (int e = 0; e < 6; ++e) gp[e] = 0.0;
      for (int s = 0; s < NS; ++s) {
        const double* src = Ps + (size_t(s) * nband + bi) * 42;
#pragma unroll
        for (int e = 0; e < 36; ++e) pt[e] += src[e];
        if (d == 0)
#pragma unroll
          for (int e = 0; e < 6; ++e) gp[e] += src[36 + e];
      }
      if (d == 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          G[size_t(ntile) * 36 + 6 * q_rb + r] = qacc[r];
          G[size_t(ntile) * 36 + R6 + 6 * q_rb + r] = gp[r];
          G[size_t(ntile) * 36 + 2 * R6 + 6 * q_rb + r] = pt[7 * r];
        }
      }
#pragma unroll
      for (int e = 0; e < 36; ++e) acc[e] += pt[e];
    }
#pragma unroll
    for (int e = 0; e < 36; e += 2) *reinterpret_cast<double2*>(G + size_t(q_t) * 36 + e) = make_double2(acc[e], acc[e + 1]);
  }
  const double s = block_sum(cost, red);
  if (tid == 0) T.cost_part[w] = s;
}

}  // namespace hs
