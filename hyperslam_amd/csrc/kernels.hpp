// kernels.hpp — gfx950 kernels of one Levenberg-Marquardt iteration (included once by capi.hip).
//
// Launch sequence per iteration (one HIP stream + one side stream, no host synchronisation; every kernel early-exits once the
// device state machine has terminated):
//   k_linearize_visual / _prior / _inertial  one residual block per lane -> segment-major records + cost partials
//   k_landmark                               one wave per landmark: H_ll, b_l, W_l -> damped 3x3 Cholesky -> Y-hat, y-hat
//   k_seg_gram / k_group_gram / k_assemble   reduced system from per-segment J'J and per-landmark-group Y-hat Y-hat' partials
//   k_border_pb / k_border_bb                border blocks of the inertial factors (bias splines, gravity)
//   k_pack_exchange -> [all-reduce] -> k_finalize_reduced / _border -> k_cost_reduce     scaling, damping, gradient test
//     (single shard without border unknowns: packing + bookkeeping are an extra workgroup of k_finalize_reduced, one launch)
//   k_band_factor_la (look-ahead, one or two ends) | k_band_factor (bw <= 22) | k_band_factor_wide (bw <= 42)   S = U'U, y
//   k_border_forward / _schur / _solve / _apply                                         bordered part of the solve
//   k_band_backward | k_band_backward2       U x = y, step outputs
//   k_backsub_retract                        step for landmarks, candidate point = Plus(x, delta), norm / model-cost partials
//   k_cost_visual / _prior / _inertial       cost at the candidate point
//   k_pack_decision -> [all-reduce] -> k_decide -> k_commit      trust-region logic and acceptance
#pragma once
#include <type_traits>

#include "factors.hpp"

namespace hs {

constexpr int kBlock = 256;

HSD double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

/// Workgroup barrier that only drains LDS traffic: global loads / stores stay in flight across it (the factorisation
/// prefetches the next band row while the current step runs; __syncthreads() would wait for vmcnt(0) every step).
HSD void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/// Deterministic block sum (fixed butterfly inside each wave, waves combined in index order). Result valid on thread 0.
HSD double block_sum(double v, double* lds /* >= blockDim/64 */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) lds[w] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < int(blockDim.x >> 6); ++i) s += lds[i];
  __syncthreads();
  return s;
}

/// Per-lane partial sum / max of a strided array with eight independent loads in flight (a plain `s += p[i]` loop keeps one
/// load in flight per lane and pays the full memory latency per element). Fixed order: bit-reproducible.
HSD double strided_sum(const double* __restrict__ p, int n, int stride = 1, int offset = 0) {
  double s = 0.0;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * blockDim.x) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * blockDim.x;
      v[u] = i < n ? p[size_t(i) * stride + offset] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  return s;
}
template <int U = 8>
HSD double strided_max(const double* __restrict__ p, int n) {  // entries >= 0; U independent loads in flight per lane
  double m = 0.0;
  for (int i0 = threadIdx.x; i0 < n; i0 += U * blockDim.x) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * blockDim.x;
      v[u] = i < n ? p[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) m = fmax(m, v[u]);
  }
  return m;
}

HSD void stage_cps(const double* __restrict__ src, double* dst, int n_doubles) {
  // control points are n x 8 doubles: 16-byte pieces, four loads in flight per lane (one round trip for up to 128 control points
  // per 256 lanes instead of one per piece)
  const double2* s2 = reinterpret_cast<const double2*>(src);
  double2* d2 = reinterpret_cast<double2*>(dst);
  const int n2 = n_doubles / 2;
  for (int i0 = threadIdx.x; i0 < n2; i0 += 4 * blockDim.x) {
    double2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      v[u] = i < n2 ? s2[i] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      if (i < n2) d2[i] = v[u];
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// Linearisation
// ---------------------------------------------------------------------------------------------------------------------
/// out_rec/out_pos: where the record of residual q goes (solver: T.v_rec at T.v_pos[q]; debug export: table order).
/// Records are transposed through a per-wave LDS slab so that the scattered 448-byte (k = 4) records leave the CU as full
/// 16-byte-per-lane stores (7 cache lines per record instead of 56 partial-line writes).
template <int K>
constexpr int lin_block() { return K <= 4 ? 256 : 128; }  // 2 waves at k = 6: the record slab is 42 KB per wave

template <int K>
__global__ void __launch_bounds__(lin_block<K>()) k_linearize_visual(Tables T, double* out_rec, const int* out_pos, int robustify,
                                                                     double* cost_part, double* cost_each) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (T.st->done) return;
  constexpr int REC = 8 + 12 * K, LREC = REC + 2, NCH = REC / 2;  // LDS record stride (16-byte aligned, bank-spread), 16-B chunks
  constexpr int NW = lin_block<K>() / 64;
  // control points: LDS copy when it fits next to the record slabs, otherwise straight from L2 (long windows)
  const bool cps_in_lds = size_t(8) * T.sp.n_cp * sizeof(double) <= 24 * 1024;
  const double* cps = cps_in_lds ? smem : T.cp;
  double* slab = smem + (cps_in_lds ? 8 * T.sp.n_cp : 0) + (threadIdx.x >> 6) * 64 * LREC;  // this wave's 64 records
  const bool lprof = (T.debug_flags & 32) && threadIdx.x == 0 && blockIdx.x < 256;
  long long* llog = reinterpret_cast<long long*>(T.xpart) + 32 * 1024 + 4 * blockIdx.x;
  if (lprof) llog[0] = wall_clock64();
  if (cps_in_lds) stage_cps(T.cp, smem, 8 * T.sp.n_cp);
  if (lprof) llog[1] = wall_clock64();
  __shared__ double red[NW];
  __shared__ int slots[NW * 64];
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0;
  int slot = -1;
  if (q < T.n_vis) {
    VisualOut<K> o;
    visual_linearize<K>(T, cps, q, robustify != 0, &o);
    cost = o.cost;
    slot = out_pos[q];
    double* rec = slab + lane * LREC;
    *reinterpret_cast<double2*>(rec) = make_double2(o.r[0], o.r[1]);
#pragma unroll
    for (int i = 0; i < 6; i += 2) *reinterpret_cast<double2*>(rec + 2 + i) = make_double2(o.Jl[i], o.Jl[i + 1]);
#pragma unroll
    for (int i = 0; i < 12 * K; i += 2) *reinterpret_cast<double2*>(rec + 8 + i) = make_double2(o.Jp[i], o.Jp[i + 1]);
    if (cost_each) cost_each[slot] = cost;
  }
  if (lprof) llog[2] = wall_clock64();
  slots[threadIdx.x] = slot;
  __builtin_amdgcn_wave_barrier();  // LDS is in-order within a wave: the slab written above is visible to the reads below
  const int* wslots = slots + (threadIdx.x & ~63);
  // 64 records x NCH 16-byte chunks per wave; eight chunks per lane are read from LDS before any is stored (the plain loop paid
  // one LDS round trip per chunk: 3.8 us of the kernel's 14)
  constexpr int SU = 8;
  for (int g0 = lane; g0 < 64 * NCH; g0 += SU * 64) {
    double2 v[SU];
    int sl[SU], cc[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int g = g0 + u * 64, gg = g < 64 * NCH ? g : 0;
      const int r = gg / NCH, c = gg % NCH;
      sl[u] = g < 64 * NCH ? wslots[r] : -1, cc[u] = c;
      v[u] = *reinterpret_cast<const double2*>(slab + r * LREC + 2 * c);
    }
#pragma unroll
    for (int u = 0; u < SU; ++u)
      if (sl[u] >= 0) *reinterpret_cast<double2*>(out_rec + size_t(sl[u]) * REC + 2 * cc[u]) = v[u];
  }
  if (lprof) llog[3] = wall_clock64();
  const double s = block_sum(cost, red);
  if (threadIdx.x == 0 && cost_part) cost_part[blockIdx.x] = s;
}

template <int K>
__global__ void __launch_bounds__(kBlock) k_linearize_prior(Tables T, double* out_rec, double* cost_part, double* cost_each) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (T.st->done) return;
  double* cps = smem;
  stage_cps(T.cp, cps, 8 * T.sp.n_cp);
  __shared__ double red[kBlock / 64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0;
  if (i < T.n_pri) {
    PriorOut<K> o;
    prior_linearize<K>(T, cps, i, &o);
    cost = o.cost;
    constexpr int REC = 6 + 36 * K;
    double* rec = out_rec + size_t(i) * REC;
#pragma unroll
    for (int c = 0; c < 6; ++c) rec[c] = o.r[c];
#pragma unroll
    for (int c = 0; c < 36 * K; ++c) rec[6 + c] = o.Jp[c];
    if (cost_each) cost_each[i] = cost;
  }
  const double s = block_sum(cost, red);
  if (threadIdx.x == 0 && cost_part) cost_part[blockIdx.x] = s;
}

/// Inertial residual blocks (inertial.cpp:13-205): record = [r(6) | J_state(6 x 6K) | wg(KB) | wa(KB) | J_gravity(6 x 2)].
template <int K, int KB>
__global__ void __launch_bounds__(kBlock) k_linearize_inertial(Tables T, double* out_rec, int robustify, double* cost_part, double* cost_each) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (T.st->done) return;
  double* cps = smem;
  stage_cps(T.cp, cps, 8 * T.sp.n_cp);
  __shared__ double red[kBlock / 64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0;
  if (i < T.n_ine) {
    InertialOut<K, KB> o;
    inertial_evaluate<K, KB, true>(T, cps, T.bias_g, T.bias_a, T.gravity, i, robustify != 0, &o);
    cost = o.cost;
    constexpr int REC = 18 + 36 * K + 2 * KB;
    double* rec = out_rec + size_t(i) * REC;
#pragma unroll
    for (int c = 0; c < 6; ++c) rec[c] = o.r[c];
#pragma unroll
    for (int c = 0; c < 36 * K; ++c) rec[6 + c] = o.Jp[c];
#pragma unroll
    for (int c = 0; c < KB; ++c) rec[6 + 36 * K + c] = o.wg[c], rec[6 + 36 * K + KB + c] = o.wa[c];
#pragma unroll
    for (int c = 0; c < 12; ++c) rec[6 + 36 * K + 2 * KB + c] = o.Jg[c];
    if (cost_each) cost_each[i] = cost;
  }
  const double s = block_sum(cost, red);
  if (threadIdx.x == 0 && cost_part) cost_part[blockIdx.x] = s;
}

template <int K, int KB>
__global__ void __launch_bounds__(kBlock) k_cost_inertial(Tables T, const double* cp_src, const double* bg, const double* ba, const double* grav,
                                                         double* cost_part) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (T.st->done) return;
  double* cps = smem;
  stage_cps(cp_src, cps, 8 * T.sp.n_cp);
  __shared__ double red[kBlock / 64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0;
  if (i < T.n_ine) {
    InertialOut<K, KB> o;
    inertial_evaluate<K, KB, false>(T, cps, bg, ba, grav, i, false, &o);
    cost = o.cost;
  }
  const double s = block_sum(cost, red);
  if (threadIdx.x == 0) cost_part[blockIdx.x] = s;
}

/// Cost at the candidate point (residual-only branch).
template <int K>
__global__ void __launch_bounds__(kBlock) k_cost_visual(Tables T, const double* cp_src, const double* lm_src, double* cost_part) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (T.st->done) return;
  double* cps = smem;
  stage_cps(cp_src, cps, 8 * T.sp.n_cp);
  __shared__ double red[kBlock / 64];
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const double cost = (q < T.n_vis) ? visual_cost<K>(T, cps, lm_src, q) : 0.0;
  const double s = block_sum(cost, red);
  if (threadIdx.x == 0) cost_part[blockIdx.x] = s;
}
template <int K>
__global__ void __launch_bounds__(kBlock) k_cost_prior(Tables T, const double* cp_src, double* cost_part) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (T.st->done) return;
  double* cps = smem;
  stage_cps(cp_src, cps, 8 * T.sp.n_cp);
  __shared__ double red[kBlock / 64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double cost = (i < T.n_pri) ? prior_cost<K>(T, cps, i) : 0.0;
  const double s = block_sum(cost, red);
  if (threadIdx.x == 0) cost_part[blockIdx.x] = s;
}

/// Second half of the landmark pass: given the wave-reduced H_ll, b_l and
/// this lane's W rows, forms V = S_l H_ll S_l + D_l^2 = L L', stores L, y-hat, the scaled gradient and the Y-hat rows.
template <int PS>
HSD void landmark_finish(const Tables& T, int dl, int lane, bool active, bool fresh, double radius, const double* sl_old, int yoff, int rows,
                         const double* h, const double* b, double (*w)[3]) {
  double sl[3];
  if (fresh) {
    sl[0] = 1.0 / (1.0 + sqrt(h[0])), sl[1] = 1.0 / (1.0 + sqrt(h[3])), sl[2] = 1.0 / (1.0 + sqrt(h[5]));
    if (lane < 3) T.lm_scale[3 * dl + lane] = sl[lane];
  } else {
    sl[0] = sl_old[0], sl[1] = sl_old[1], sl[2] = sl_old[2];
  }
  // V = S H S + clamp(diag)/radius
  double v00 = sl[0] * sl[0] * h[0], v01 = sl[0] * sl[1] * h[1], v02 = sl[0] * sl[2] * h[2];
  double v11 = sl[1] * sl[1] * h[3], v12 = sl[1] * sl[2] * h[4], v22 = sl[2] * sl[2] * h[5];
  const double inv_radius = 1.0 / radius;
  const double d0 = fmin(fmax(v00, 1e-6), 1e32) * inv_radius, d1 = fmin(fmax(v11, 1e-6), 1e32) * inv_radius, d2 = fmin(fmax(v22, 1e-6), 1e32) * inv_radius;
  v00 += d0, v11 += d1, v22 += d2;
  // Cholesky V = L L' with reciprocal pivots: every lane runs this redundantly, and a double-precision divide or square root
  // costs ~35 instructions, so the 3x3 factor and the row solves below use 1 / l_ii from the hardware rsq estimate + one
  // third-order correction (error ~ e^3, full double accuracy) and multiply.
  auto rsqrt_refined = [](double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
  };
  const double i00 = rsqrt_refined(v00), l00 = v00 * i00, l10 = v01 * i00, l20 = v02 * i00;
  const double p11 = v11 - l10 * l10, i11 = rsqrt_refined(p11), l11 = p11 * i11, l21 = (v12 - l20 * l10) * i11;
  const double p22 = v22 - l20 * l20 - l21 * l21, i22 = rsqrt_refined(p22), l22 = p22 * i22;
  const double sb0 = sl[0] * b[0], sb1 = sl[1] * b[1], sb2 = sl[2] * b[2];
  const double y0 = sb0 * i00, y1 = (sb1 - l10 * y0) * i11, y2 = (sb2 - l20 * y0 - l21 * y1) * i22;
  if (lane == 0) {
    double* L = T.lm_L + 6 * dl;
    L[0] = l00, L[1] = l10, L[2] = l11, L[3] = l20, L[4] = l21, L[5] = l22;
    T.lm_yhat[3 * dl] = active ? y0 : 0.0, T.lm_yhat[3 * dl + 1] = active ? y1 : 0.0, T.lm_yhat[3 * dl + 2] = active ? y2 : 0.0;
    T.lm_sb[3 * dl] = sb0, T.lm_sb[3 * dl + 1] = sb1, T.lm_sb[3 * dl + 2] = sb2;
    T.lm_D2[3 * dl] = d0, T.lm_D2[3 * dl + 1] = d1, T.lm_D2[3 * dl + 2] = d2;
    // gradient max norm: per-landmark value, max-reduced by k_pack_exchange (thousands of atomics on one word would
    // serialise at ~12 ns each and dominate this pass)
    T.lm_gmax[dl] = active ? fmax(fabs(b[0]), fmax(fabs(b[1]), fabs(b[2]))) : 0.0;
  }
  // W rows -> Y-hat rows
  double* Y = T.Y + yoff;
#pragma unroll
  for (int ps = 0; ps < PS; ++ps) {
    const int rho = lane + 64 * ps;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (rho < rows) {
      const double w0 = w[ps][0] * sl[0], w1 = w[ps][1] * sl[1], w2 = w[ps][2] * sl[2];
      // y L' = w  (forward substitution on the columns of L')
      a0 = w0 * i00, a1 = (w1 - a0 * l10) * i11, a2 = (w2 - a0 * l20 - a1 * l21) * i22;
      if (!active) a0 = a1 = a2 = 0.0;
      Y[3 * rho] = a0, Y[3 * rho + 1] = a1, Y[3 * rho + 2] = a2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Landmark pass: one wave per landmark.  H_ll = sum Jl'Jl, b_l = sum Jl'r, W_l = sum Jp'Jl over the landmark's
// residuals; V = S_l H_ll S_l + D_l^2 = L L';  Y-hat = W S_l L^-T (pose-side row scaling is applied by the consumer),
// y-hat = L^-1 S_l b_l.  Jacobi scaling S_l is fixed at iteration 0 (TrustRegionMinimizer, jacobi_scaling = true).
// PS = 64-row passes a lane owns (rows of W = 6 * control points the landmark touches <= 64 * PS).
// ---------------------------------------------------------------------------------------------------------------------
template <int K, int PS, int U>
HSD void landmark_eliminate(const Tables& T, int dl, int lane) {
  constexpr int REC = 8 + 12 * K;
  const int q0 = T.lm_ptr[dl], q1 = T.lm_ptr[dl + 1];
  const int c_first = T.lm_cfirst[dl], rows = 6 * T.lm_ncp[dl];
  // operands of the finishing step: requested up front, they do not depend on the records
  const bool fresh = !T.st->scaling_ready;
  const double radius = T.st->radius;
  const bool is_const = T.lm_const[dl];
  const int yoff = T.lm_yoff[dl];
  double sl_old[3] = {1.0, 1.0, 1.0};
  if (!fresh) sl_old[0] = T.lm_scale[3 * dl], sl_old[1] = T.lm_scale[3 * dl + 1], sl_old[2] = T.lm_scale[3 * dl + 2];
  // One pass over the landmark's residuals: lane q of a 64-chunk fetches (first control point, record slot) of residual q
  // once; the chunk is then walked with register broadcasts, every lane accumulating its own W row(s) (rho = lane, lane + 64)
  // and lane q the H_ll / b_l terms of residual q. All loads are unconditional on clamped indices and masked afterwards:
  // straight-line code, so the loads of U records (all 64-row passes) are in flight together instead of one round trip per
  // record and pass. (The kernel is nevertheless bound by instruction issue, not by this chain: U = 1, 2, 4 and the branchy
  // original all take 17.3 us at 5 000 landmarks x 10 records — only 6K of 64 lanes carry a W row of a given record.)
  double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  double w[PS][3];
#pragma unroll
  for (int ps = 0; ps < PS; ++ps) w[ps][0] = w[ps][1] = w[ps][2] = 0.0;
  for (int base = q0; base < q1; base += 64) {
    const int myq = min(base + lane, q1 - 1);
    const bool mine = base + lane < q1;
    const int my_first = T.v_first[myq], my_pos = T.v_pos[myq];
    const double* myrec = T.v_rec + size_t(my_pos) * REC;
    double own[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) own[e] = myrec[e];
    const int cnt = min(64, q1 - base);
    for (int t0 = 0; t0 < cnt; t0 += U) {
      double ja[U][PS], jb[U][PS], jl[U][6];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = min(t0 + u, cnt - 1);
        const int ft = __builtin_amdgcn_readlane(my_first, t), pt = __builtin_amdgcn_readlane(my_pos, t);  // wave-uniform
        const double* rec = T.v_rec + size_t(pt) * REC;
        const int off = 6 * (ft - c_first);
#pragma unroll
        for (int e = 0; e < 6; ++e) jl[u][e] = rec[2 + e];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
          const int c = lane + 64 * ps - off;
          const bool ok = t0 + u < cnt && c >= 0 && c < 6 * K && lane + 64 * ps < rows;
          const int cc = ok ? c : 0;
          const double va = rec[8 + cc], vb = rec[8 + 6 * K + cc];
          ja[u][ps] = ok ? va : 0.0, jb[u][ps] = ok ? vb : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {  // masked slots add exact zeros
          w[ps][0] = fma(ja[u][ps], jl[u][0], fma(jb[u][ps], jl[u][3], w[ps][0]));
          w[ps][1] = fma(ja[u][ps], jl[u][1], fma(jb[u][ps], jl[u][4], w[ps][1]));
          w[ps][2] = fma(ja[u][ps], jl[u][2], fma(jb[u][ps], jl[u][5], w[ps][2]));
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double rr = mine ? own[r] : 0.0, j0 = mine ? own[2 + 3 * r] : 0.0, j1 = mine ? own[3 + 3 * r] : 0.0, j2 = mine ? own[4 + 3 * r] : 0.0;
      h[0] = fma(j0, j0, h[0]), h[1] = fma(j0, j1, h[1]), h[2] = fma(j0, j2, h[2]);
      h[3] = fma(j1, j1, h[3]), h[4] = fma(j1, j2, h[4]), h[5] = fma(j2, j2, h[5]);
      b[0] = fma(j0, rr, b[0]), b[1] = fma(j1, rr, b[1]), b[2] = fma(j2, rr, b[2]);
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) h[i] = wave_sum(h[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) b[i] = wave_sum(b[i]);
  landmark_finish<PS>(T, dl, lane, (q1 > q0) && !is_const, fresh, radius, sl_old, yoff, rows, h, b, w);
}

template <int K, int PS, int U>
__global__ void __launch_bounds__(kBlock) k_landmark(Tables T) {
  if (T.st->done) return;
  const int dl = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (dl >= T.n_lm) return;
  landmark_eliminate<K, PS, U>(T, dl, threadIdx.x & 63);
}

// ---------------------------------------------------------------------------------------------------------------------
// Reduced system  S = Sp (J_p'J_p) Sp + D_p^2 - Sp (sum_l Yh_l Yh_l') Sp,   g = Sp (g_p - sum_l Yh_l yh_l)  (raw, unscaled parts here;
// scaling and damping in k_finalize_reduced). Owner-computes formulation: every record and every Y-hat row is read ONCE.
//   k_seg_gram<K>   : one workgroup per (segment, split): P = sum J_p' J_p (6K x 6K) and J_p' r over the segment's records
//   k_group_gram<NT>: one workgroup per (first control point c, split): Q = - sum_l Yh_l Yh_l' over the landmarks whose
//                     track starts at c (6 bw x 6 bw window, upper 6x6 tiles), q = - sum_l Yh_l yh_l
//   k_assemble<K>   : block row i = sum of the <= K segment partials and <= bw group partials that overlap it, in a fixed
//                     order (bit-reproducible, no floating-point atomics), written straight into the exchange buffer
// (The first version gathered per block row and re-read each record K times and each Y-hat row once per covered control point.)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSegStage = 6144;  // doubles of record data staged in LDS per round (48 KB)

template <int K>
__global__ void __launch_bounds__(kBlock) k_seg_gram(Tables T) {
  extern __shared__ __attribute__((aligned(16))) double stage[];  // kSegStage doubles: a contiguous run of records
  __shared__ __attribute__((aligned(16))) double red[kBlock * 12 + kBlock * 3];
  if (T.st->done) return;
  constexpr int NCA = 6 * K, RG = NCA / 3, CG = NCA / 4, TPS = RG * CG, NS = kBlock / TPS;  // 3x4 register tiles, NS record streams
  constexpr int VREC = 8 + 12 * K, PREC = 6 + 36 * K;
  // work list: workgroup w serves segment sw_seg[w] as split sp of nsp (splits proportional to the segment's record count: the
  // first and last segment of a window collect the clamped stamps)
  const int first = T.sw_seg[blockIdx.x], sp = blockIdx.x - T.sw_ptr[first], nsp = T.sw_ptr[first + 1] - T.sw_ptr[first];
  const int tid = threadIdx.x;
  const int stream = tid / TPS, tb = tid % TPS, rg = tb / CG, cg = tb % CG;
  const bool sprof = (T.debug_flags & 32) && tid == 0 && sp == 0 && first < 128;
  long long* slog = reinterpret_cast<long long*>(T.xpart) + 8 * 1024 + 8 * first;
  if (sprof) slog[0] = wall_clock64();
  // a tile is needed if some column block >= the row block (upper block triangle); column group 0 also carries J'r
  const bool live = stream < NS && (cg == 0 || (4 * cg + 3) / 6 >= (3 * rg) / 6);
  double acc[3][4], gacc[3] = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
  // Records of a segment are contiguous (segment-major), split `sp` takes a contiguous share: one coalesced sweep brings a run
  // of records into LDS (a single HBM round trip instead of one per record), the streams then walk it from LDS.
  auto run = [&](const double* recs, int r0, int r1, int REC, int n_rows, int joff) {
    const int n = r1 - r0, lo = r0 + int((long long)n * sp / nsp), hi = r0 + int((long long)n * (sp + 1) / nsp);
    const int per = kSegStage / REC;
    for (int c0 = lo; c0 < hi; c0 += per) {
      const int cnt = min(per, hi - c0);
      __syncthreads();
      const double2* src = reinterpret_cast<const double2*>(recs + size_t(c0) * REC);
      const int n2 = cnt * REC / 2;
      for (int e0 = tid; e0 < n2; e0 += 8 * kBlock) {  // eight independent 16-byte loads in flight per lane
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kBlock;
          v[u] = e < n2 ? src[e] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kBlock;
          if (e < n2) reinterpret_cast<double2*>(stage)[e] = v[u];
        }
      }
      __syncthreads();
      if (sprof) slog[1] = wall_clock64();
      if (live)
        for (int c = stream; c < cnt; c += NS) {
          const double* rec = stage + c * REC;
#pragma unroll 2
          for (int r = 0; r < n_rows; ++r) {
            const double* j = rec + joff + r * NCA;
            const double a0 = j[3 * rg], a1 = j[3 * rg + 1], a2 = j[3 * rg + 2];
            const double2 b01 = *reinterpret_cast<const double2*>(j + 4 * cg), b23 = *reinterpret_cast<const double2*>(j + 4 * cg + 2);
            acc[0][0] = fma(a0, b01.x, acc[0][0]), acc[0][1] = fma(a0, b01.y, acc[0][1]), acc[0][2] = fma(a0, b23.x, acc[0][2]), acc[0][3] = fma(a0, b23.y, acc[0][3]);
            acc[1][0] = fma(a1, b01.x, acc[1][0]), acc[1][1] = fma(a1, b01.y, acc[1][1]), acc[1][2] = fma(a1, b23.x, acc[1][2]), acc[1][3] = fma(a1, b23.y, acc[1][3]);
            acc[2][0] = fma(a2, b01.x, acc[2][0]), acc[2][1] = fma(a2, b01.y, acc[2][1]), acc[2][2] = fma(a2, b23.x, acc[2][2]), acc[2][3] = fma(a2, b23.y, acc[2][3]);
            if (cg == 0) {
              const double rr = rec[r];
              gacc[0] = fma(a0, rr, gacc[0]), gacc[1] = fma(a1, rr, gacc[1]), gacc[2] = fma(a2, rr, gacc[2]);
            }
          }
        }
    }
  };
  run(T.v_rec, T.v_seg_ptr[first], T.v_seg_ptr[first + 1], VREC, 2, 8);
  if (T.n_pri) run(T.p_rec, T.p_seg_ptr[first], T.p_seg_ptr[first + 1], PREC, 6, 6);
  if (T.n_ine) run(T.i_rec, T.i_seg_ptr[first], T.i_seg_ptr[first + 1], 18 + 36 * K + 2 * T.kb, 6, 6);
  if (sprof) slog[2] = wall_clock64();
  // combine the record streams in index order
  double* racc = red;                 // [stream][TPS][12]
  double* rg3 = red + kBlock * 12;    // [stream][RG][3]
  if (stream < NS) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) racc[(stream * TPS + tb) * 12 + 4 * r + c] = acc[r][c];
    if (cg == 0)
#pragma unroll
      for (int r = 0; r < 3; ++r) rg3[(stream * RG + rg) * 3 + r] = gacc[r];
  }
  __syncthreads();
  double* P = T.segP + size_t(blockIdx.x) * (NCA * NCA + NCA);
  for (int e = tid; e < NCA * NCA; e += kBlock) {
    const int a = e / NCA, c = e % NCA;
    const int t = (a / 3) * CG + c / 4, in = 4 * (a % 3) + c % 4;
    double v = 0.0;
#pragma unroll
    for (int st = 0; st < NS; ++st) v += racc[(st * TPS + t) * 12 + in];
    P[e] = v;  // tiles below the block diagonal were never accumulated (zeros) and are never read
  }
  if (tid < NCA) {
    double v = 0.0;
#pragma unroll
    for (int st = 0; st < NS; ++st) v += rg3[(st * RG + tid / 3) * 3 + tid % 3];
    P[NCA * NCA + tid] = v;
  }
  if (sprof) slog[3] = wall_clock64();
}

HSD int ok_index(int b, int nb) { return b < nb ? b : 0; }

/// Upper 6x6 tiles of the 6 bw x 6 bw window of a landmark group: tile index of (rb, cb), rb <= cb < bw.
HSD int group_tile_index(int rb, int cb, int bw) { return rb * bw - rb * (rb - 1) / 2 + (cb - rb); }

constexpr int kGroupBatch = 16;  // landmarks staged in LDS per round (host caps it so that the stage fits 48 KB)

template <int NT>  // tiles per thread: NT == 1: two landmark streams of 128 lanes (bw <= 15); NT > 1: one stream, bw (bw + 1) / 2 <= NT * kBlock
__global__ void __launch_bounds__(kBlock, NT == 1 ? 3 : 1) k_group_gram(Tables T, int batch) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int m_ncp[kBlock], m_off[kBlock];
  if (T.st->done) return;
  // work list: workgroup w serves group cf = gw_cf[w] as split sp of nsp (splits proportional to the group's landmark count:
  // the first control point of a window collects every track that started before it)
  const int cf = T.gw_cf[blockIdx.x], sp = blockIdx.x - T.gw_ptr[cf], nsp = T.gw_ptr[cf + 1] - T.gw_ptr[cf];
  const int tid = threadIdx.x;
  const int bw = T.bw, R = 6 * bw, ntile = bw * (bw + 1) / 2;
  double* ybuf = smem;                          // batch x (R x 3): Y-hat rows (zero past the landmark's rows)
  double* yh = smem + size_t(batch) * R * 3;    // batch x 4: y-hat
  const bool two = NT == 1 && ntile <= kBlock / 2;  // two landmark streams
  const int stream = two ? tid / (kBlock / 2) : 0, nstream = two ? 2 : 1;
  const int lt = two ? tid % (kBlock / 2) : tid, lthreads = two ? kBlock / 2 : kBlock;
  int t_rb[NT], t_cb[NT];
  bool t_ok[NT];
  double acc[NT][36], qacc[NT][6];
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    const int t = lt + m * lthreads;
    t_ok[m] = t < ntile;
    int rb = 0, rem = t_ok[m] ? t : 0;
    while (rem >= bw - rb) rem -= bw - rb, ++rb;  // row rb holds bw - rb tiles
    t_rb[m] = rb, t_cb[m] = rb + rem;
#pragma unroll
    for (int e = 0; e < 36; ++e) acc[m][e] = 0.0;
#pragma unroll
    for (int e = 0; e < 6; ++e) qacc[m][e] = 0.0;
  }
  const bool gprof = (T.debug_flags & 32) && tid == 0 && sp == 0 && cf < 128;
  long long* glog = reinterpret_cast<long long*>(T.xpart) + 8 * cf;
  if (gprof) glog[0] = wall_clock64();
  const int dl0 = T.cf_ptr[cf], dl1 = T.cf_ptr[cf + 1];
  const int n_mine = dl1 > dl0 + sp ? (dl1 - dl0 - sp + nsp - 1) / nsp : 0;  // landmarks dl = dl0 + sp + t * nsp
  for (int t0 = 0; t0 < n_mine; t0 += kBlock) {  // (one pass unless a group holds more than 256 landmarks per split)
    __syncthreads();
    if (t0 + tid < n_mine) {
      const int dl = dl0 + sp + (t0 + tid) * nsp;
      m_ncp[tid] = T.lm_ncp[dl], m_off[tid] = T.lm_yoff[dl];
    }
    __syncthreads();
    const int n_pass = min(kBlock, n_mine - t0);
    if (gprof) glog[1] = wall_clock64();
    for (int b0 = 0; b0 < n_pass; b0 += batch) {
      const int nb = min(batch, n_pass - b0);
      __syncthreads();
      for (int e0 = tid; e0 < nb * R * 3; e0 += 8 * kBlock) {  // eight independent loads in flight per lane
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kBlock, b = e / (R * 3), w = e % (R * 3);
          const bool ok = e < nb * R * 3 && w < 18 * m_ncp[b0 + (ok_index(b, nb))];
          v[u] = ok ? T.Y[m_off[b0 + ok_index(b, nb)] + w] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kBlock;
          if (e < nb * R * 3) ybuf[e] = v[u];
        }
      }
      if (tid < 3 * nb) yh[4 * (tid / 3) + tid % 3] = T.lm_yhat[3 * (dl0 + sp + (t0 + b0 + tid / 3) * nsp) + tid % 3];
      __syncthreads();
      if (gprof) glog[2] = wall_clock64();
      for (int b = stream; b < nb; b += nstream) {
        const int ncp = m_ncp[b0 + b];
        const double* Yb = ybuf + size_t(b) * R * 3;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          if (!t_ok[m] || t_cb[m] >= ncp) continue;
          double B[18];
#pragma unroll
          for (int e = 0; e < 18; e += 2) {
            const double2 vb = *reinterpret_cast<const double2*>(Yb + 18 * t_cb[m] + e);
            B[e] = vb.x, B[e + 1] = vb.y;
          }
          const bool diag = t_rb[m] == t_cb[m];
          const double y0 = yh[4 * b], y1 = yh[4 * b + 1], y2 = yh[4 * b + 2];
#pragma unroll
          for (int rp = 0; rp < 3; ++rp) {  // two rows of the A operand at a time: 148 instead of 190 registers, three workgroups per CU
            double A[6];
#pragma unroll
            for (int e = 0; e < 6; e += 2) {
              const double2 va = *reinterpret_cast<const double2*>(Yb + 18 * t_rb[m] + 6 * rp + e);
              A[e] = va.x, A[e + 1] = va.y;
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const int r = 2 * rp + rr;
#pragma unroll
              for (int c = 0; c < 6; ++c)
                acc[m][6 * r + c] = fma(-A[3 * rr + 2], B[3 * c + 2], fma(-A[3 * rr + 1], B[3 * c + 1], fma(-A[3 * rr], B[3 * c], acc[m][6 * r + c])));
              if (diag) qacc[m][r] = fma(-A[3 * rr + 2], y2, fma(-A[3 * rr + 1], y1, fma(-A[3 * rr], y0, qacc[m][r])));
            }
          }
        }
      }
    }
  }
  if (gprof) glog[3] = wall_clock64(), glog[5] = n_mine;
  double* Q = T.grpQ + size_t(blockIdx.x) * (size_t(ntile) * 36 + R);
  if (two) {  // stream 1 hands its partial to stream 0 through LDS (fixed order: stream 0 + stream 1)
    __syncthreads();
    double* xch = smem;  // 128 x 42 doubles <= the stage
    if (stream == 1 && t_ok[0]) {
#pragma unroll
      for (int e = 0; e < 36; ++e) xch[lt * 42 + e] = acc[0][e];
#pragma unroll
      for (int e = 0; e < 6; ++e) xch[lt * 42 + 36 + e] = qacc[0][e];
    }
    __syncthreads();
    if (stream == 0 && t_ok[0]) {
#pragma unroll
      for (int e = 0; e < 36; ++e) acc[0][e] += xch[lt * 42 + e];
#pragma unroll
      for (int e = 0; e < 6; ++e) qacc[0][e] += xch[lt * 42 + 36 + e];
    }
    if (stream == 1) return;
  }
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    if (!t_ok[m]) continue;
    const int t = lt + m * lthreads;
#pragma unroll
    for (int e = 0; e < 36; e += 2) *reinterpret_cast<double2*>(Q + size_t(t) * 36 + e) = make_double2(acc[m][e], acc[m][e + 1]);
    if (t_rb[m] == t_cb[m])
#pragma unroll
      for (int r = 0; r < 6; ++r) Q[size_t(ntile) * 36 + 6 * t_rb[m] + r] = qacc[m][r];
  }
  if (gprof) glog[4] = wall_clock64();
}

constexpr int kAsmThreads = 512, kAsmU = 8;  // lanes per scalar row, loads in flight per lane

/// Scalar row rho = 6 i + a of the raw (unscaled, undamped) reduced system from the segment and group partials; writes xbuf
/// directly. Grid (n_cp, 6). The sources of an entry are dealt round-robin to `nsl` thread slices (loads of a slice are
/// issued kAsmU at a time), the slices are combined through LDS in index order: fixed summation order, bit-reproducible.
template <int K>
__global__ void __launch_bounds__(kAsmThreads) k_assemble(Tables T) {
  __shared__ double part[2][kAsmThreads];
  if (T.st->done) return;
  constexpr int NCA = 6 * K;
  const int i = blockIdx.x, a = blockIdx.y, tid = threadIdx.x;
  const int bw = T.bw, ncb = 6 * bw, R = 6 * bw, ntile = bw * (bw + 1) / 2;
  const size_t pstride = NCA * NCA + NCA, qstride = size_t(ntile) * 36 + R;
  const int f0 = max(0, i - K + 1), f1 = min(i, T.n_seg - 1);
  const int c0 = max(0, i - bw + 1);
  const int nent = ncb + 2;  // band entries + [J'r | Y-hat y-hat] of this row
  const int nsl = max(1, kAsmThreads / nent), sl = tid / nent, c = tid % nent;
  double va = 0.0, vb = 0.0;  // J'J part / Schur part
  if (sl < nsl) {
    const int kk = c / 6, cc = c % 6;
    // segment partials: the workgroups of segments f0 .. f1 are contiguous in the work list; landmark-group partials: those of
    // groups c0 .. i. A lane's sources are p = sl, sl + nsl, ... The first kAsmU segment sources and 2 kAsmU group sources are
    // fetched in two rounds (all work-list entries, then all partial values: two memory round trips instead of one pair per
    // batch); the sums run in the same fixed order as a plain loop over p.
    const bool a_live = (c < ncb ? kk < K : c == ncb) && f1 >= f0;
    const bool b_live = T.n_lm > 0 && (c < ncb || c == ncb + 1);
    const int p_lo = a_live ? T.sw_ptr[f0] : 0, np_ = a_live ? T.sw_ptr[f1 + 1] - p_lo : 0;
    const int q_lo = b_live ? T.gw_ptr[c0] : 0, nq = b_live ? T.gw_ptr[i + 1] - q_lo : 0;
    // (plain macros, not lambdas: a by-reference closure kept these operands in scratch memory)
#define HS_SEG_VALUE(p, seg) \
  ((p) < np_ && (c == ncb || i - (seg) + kk < K) \
       ? T.segP[(p_lo + (p)) * int(pstride) + (c == ncb ? NCA * NCA + 6 * (i - (seg)) + a : (6 * (i - (seg)) + a) * NCA + 6 * (i - (seg)) + c)] \
       : 0.0)
#define HS_GRP_VALUE(q, cf) \
  ((q) < nq && (c > ncb || i - (cf) + kk < bw) \
       ? T.grpQ[(q_lo + (q)) * int(qstride) + \
                (c > ncb ? ntile * 36 + 6 * (i - (cf)) + a : group_tile_index(i - (cf), i - (cf) + kk, bw) * 36 + 6 * a + cc)] \
       : 0.0)
    int si[kAsmU], gi[2 * kAsmU];
#pragma unroll
    for (int u = 0; u < kAsmU; ++u) si[u] = sl + u * nsl < np_ ? T.sw_seg[p_lo + sl + u * nsl] : 0;
#pragma unroll
    for (int u = 0; u < 2 * kAsmU; ++u) gi[u] = sl + u * nsl < nq ? T.gw_cf[q_lo + sl + u * nsl] : 0;
    double sv[kAsmU], gv[2 * kAsmU];
#pragma unroll
    for (int u = 0; u < kAsmU; ++u) sv[u] = HS_SEG_VALUE(sl + u * nsl, si[u]);
#pragma unroll
    for (int u = 0; u < 2 * kAsmU; ++u) gv[u] = HS_GRP_VALUE(sl + u * nsl, gi[u]);
#pragma unroll
    for (int u = 0; u < kAsmU; ++u) va += sv[u];
#pragma unroll
    for (int u = 0; u < 2 * kAsmU; ++u) vb += gv[u];
    // the rest (segments / groups split into unusually many workgroups)
    for (int p0 = sl + kAsmU * nsl; p0 < np_; p0 += kAsmU * nsl) {
      double v[kAsmU];
#pragma unroll
      for (int u = 0; u < kAsmU; ++u) {
        const int p = p0 + u * nsl;
        const int seg = p < np_ ? T.sw_seg[p_lo + p] : 0;
        v[u] = HS_SEG_VALUE(p, seg);
      }
#pragma unroll
      for (int u = 0; u < kAsmU; ++u) va += v[u];
    }
    for (int q0 = sl + 2 * kAsmU * nsl; q0 < nq; q0 += kAsmU * nsl) {
      double v[kAsmU];
#pragma unroll
      for (int u = 0; u < kAsmU; ++u) {
        const int q = q0 + u * nsl;
        const int cf = q < nq ? T.gw_cf[q_lo + q] : 0;
        v[u] = HS_GRP_VALUE(q, cf);
      }
#pragma unroll
      for (int u = 0; u < kAsmU; ++u) vb += v[u];
    }
#undef HS_SEG_VALUE
#undef HS_GRP_VALUE
  }
  part[0][tid] = va, part[1][tid] = vb;
  __syncthreads();
  if (tid < nent) {
    double sa = 0.0, sb = 0.0;
    for (int q = 0; q < nsl; ++q) sa += part[0][q * nent + tid], sb += part[1][q * nent + tid];
    const int rho = 6 * i + a;
    if (tid < ncb) {
      T.xbuf[size_t(rho) * ncb + tid] = sa + sb;
      if (tid == a) T.xbuf[T.xo_dj + rho] = sa;
    } else if (tid == ncb) {
      T.xbuf[T.xo_g + rho] = sa;
    } else {
      T.xbuf[T.xo_gs + rho] = sb;
    }
  }
}

/// xbuf[e] = sum over the accumulation splits (fixed order => bit-reproducible). The result is additive across residual shards.
__global__ void __launch_bounds__(kBlock) k_reduce_partials(Tables T, int nsp, int e0) {
  if (T.st->done) return;
  const int n = T.xo_bb;  // [Sraw | g_p | g_schur | diag | Hpb]; e0 = xo_pb when the pose part comes from k_assemble
  for (int e = e0 + blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int k = 0; k < nsp; ++k) s += T.xpart[size_t(k) * T.x_count1 + e];
    T.xbuf[e] = s;
  }
}

HSD void begin_iteration(const Tables& T, double cost, double gmax, bool set_scaling_ready);

/// Local cost and landmark-side gradient max norm into the exchange buffer (slot per rank so that a SUM all-reduce
/// delivers every rank's value to every rank). reduce_here (single shard, no border unknowns): nothing is exchanged, so the
/// iteration bookkeeping of k_cost_reduce is done right here (the pose-side gradient is already in the buffer).
HSD void pack_exchange_body(const Tables& T, int reduce_here) {
  __shared__ double red[kBlock / 64];
  DevState* st = T.st;
  if (st->done) return;
  double s = strided_sum(T.cost_part, T.n_cost_part);
  double gm = strided_max<24>(T.lm_gmax, T.n_obs_lm);  // one value per landmark: a single round of loads at 5 000 landmarks
  if (reduce_here) {
    const double* gp = T.xbuf + T.xo_g;
    for (int i0 = threadIdx.x; i0 < T.np; i0 += 8 * blockDim.x) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * blockDim.x;
        v[u] = i < T.np ? fabs(gp[i]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) gm = fmax(gm, v[u]);
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) T.xbuf[T.xo_cost] = s;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_xor(gm, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gm;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < int(blockDim.x >> 6); ++i) gm = fmax(gm, red[i]);
    for (int r = 0; r < T.world; ++r) T.xbuf[T.xo_gmax + r] = (r == T.rank) ? gm : 0.0;
    if (reduce_here) begin_iteration(T, s, gm, /*set_scaling_ready=*/false);  // k_finalize_reduced of this linearisation still needs the flag
  }
}
__global__ void __launch_bounds__(kBlock) k_pack_exchange(Tables T, int reduce_here) { pack_exchange_body(T, reduce_here); }

// ---------------------------------------------------------------------------------------------------------------------
// Border unknowns (IMU bias-spline control points + gravity; SURVEY a-4): the inertial factor couples every control point of
// the window with a few *dense* unknowns, ordered last:  [gyro bias 3 n_bias | accel bias 3 n_bias | gravity 2] = nb.
//   H_pb (np x nb), H_bb (nb x nb), g_b (nb) are gathered deterministically from the inertial records.
// Record structure exploited: d r_ang / d b_g,j = wg[j] I_3, d r_lin / d b_a,j = wa[j] I_3 (only the weights are stored).
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(128) k_border_pb(Tables T) {
  // block (i, split): rows 6 i .. 6 i + 5 of H_pb, thread <-> border column
  if (T.st->done) return;
  const int i = blockIdx.x, sp = blockIdx.y, nsp = gridDim.y;
  const int kb = T.kb, nbias = T.n_bias, nb = T.nb;
  const int IREC = 18 + 36 * K + 2 * kb;
  double* out = T.xpart + size_t(sp) * T.x_count1 + T.xo_pb;
  for (int beta = threadIdx.x; beta < nb; beta += blockDim.x) {
    double acc[6] = {0, 0, 0, 0, 0, 0};
    // classify the column once
    const int kind = beta < 3 * nbias ? 0 : (beta < 6 * nbias ? 1 : 2);
    const int bb = kind == 2 ? 0 : (beta - 3 * nbias * kind) / 3, cc = kind == 2 ? beta - 6 * nbias : (beta - 3 * nbias * kind) % 3;
    const int f0 = max(0, i - K + 1), f1 = min(i, T.n_seg - 1);
    for (int first = f0; first <= f1; ++first) {
      const int ao = 6 * (i - first);
      if (kind == 2) {
#pragma unroll 2
        for (int pos = T.i_seg_ptr[first] + sp; pos < T.i_seg_ptr[first + 1]; pos += nsp) {
          const double* rec = T.i_rec + size_t(pos) * IREC;
          const double* jp = rec + 6;
          const double* jg = rec + 6 + 36 * K + 2 * kb;
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            const double g = jg[2 * r + cc];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[a] = fma(jp[r * 6 * K + ao + a], g, acc[a]);
          }
        }
      } else {
        // branch-free body (clamped weight index, masked weight) so that the loads of four records are in flight together
#pragma unroll 4
        for (int pos = T.i_seg_ptr[first] + sp; pos < T.i_seg_ptr[first + 1]; pos += nsp) {
          const double* rec = T.i_rec + size_t(pos) * IREC;
          const int j = bb - T.i_first_bias[pos];
          const bool ok = j >= 0 && j < kb;
          const double wv = rec[6 + 36 * K + kind * kb + (ok ? j : 0)];
          const double wgt = ok ? wv : 0.0;
          const double* row = rec + 6 + (3 * kind + cc) * 6 * K + ao;
#pragma unroll
          for (int a = 0; a < 6; ++a) acc[a] = fma(row[a], wgt, acc[a]);
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) out[size_t(6 * i + a) * nb + beta] = acc[a];
  }
}

/// H_bb and J_b' r. One workgroup per bias control point b (gyro and accel parts): the records whose bias window covers b are
/// dealt to 256 lanes, sums are combined wave by wave in a fixed order; each entry of the exchange buffer has a single writer
/// (the region is zero-filled first by k_border_zero). The gravity block is accumulated per b over the records that START at b
/// (every record exactly once) into T.gravity_part[b][5] and summed by k_border_gravity.
template <int K>
__global__ void __launch_bounds__(kBlock) k_border_bb(Tables T) {
  constexpr int NV = 2 * hsd::kMaxOrder + 18 + 5;
  __shared__ double red[kBlock / 64][NV];
  if (T.st->done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int kb = T.kb, nbias = T.n_bias, nb = T.nb;
  const int IREC = 18 + 36 * K + 2 * kb;
  double* Hbb = T.xbuf + T.xo_bb;
  double* gb = T.xbuf + T.xo_gb;
  const int og = 0, oa = 3 * nbias, ogr = 6 * nbias;
  // records whose bias window covers b: first_bias in [b - kb + 1, b]
  const int p0 = T.i_bias_ptr[max(0, b - kb + 1)], p1 = T.i_bias_ptr[b + 1], pown = T.i_bias_ptr[b];
  double v[NV];  // [gg(kMaxOrder) | aa(kMaxOrder) | ggr 6 | agr 6 | rg 3 | ra 3 | gravity h00 h01 h11 g0 g1]
#pragma unroll
  for (int e = 0; e < NV; ++e) v[e] = 0.0;
  double* gg = v, *aa = v + hsd::kMaxOrder, *ggr = v + 2 * hsd::kMaxOrder, *agr = ggr + 6, *rg = agr + 6, *ra = rg + 3, *hg = ra + 3;
  for (int pos = p0 + tid; pos < p1; pos += kBlock) {
    const double* rec = T.i_rec + size_t(pos) * IREC;
    const int j = b - T.i_first_bias[pos];
    const double* wgp = rec + 6 + 36 * K;
    const double* wap = wgp + kb;
    const double* jg = wap + kb;
    const double wgb = wgp[j], wab = wap[j];
#pragma unroll
    for (int d = 0; d < hsd::kMaxOrder; ++d)
      if (d < kb && j + d < kb) gg[d] = fma(wgb, wgp[j + d], gg[d]), aa[d] = fma(wab, wap[j + d], aa[d]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ggr[2 * c] = fma(wgb, jg[2 * c], ggr[2 * c]), ggr[2 * c + 1] = fma(wgb, jg[2 * c + 1], ggr[2 * c + 1]);
      agr[2 * c] = fma(wab, jg[2 * (3 + c)], agr[2 * c]), agr[2 * c + 1] = fma(wab, jg[2 * (3 + c) + 1], agr[2 * c + 1]);
      rg[c] = fma(wgb, rec[c], rg[c]), ra[c] = fma(wab, rec[3 + c], ra[c]);
    }
    if (pos >= pown) {  // j == 0: this record starts at b -> its gravity terms are counted here
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        hg[0] = fma(jg[2 * r], jg[2 * r], hg[0]), hg[1] = fma(jg[2 * r], jg[2 * r + 1], hg[1]), hg[2] = fma(jg[2 * r + 1], jg[2 * r + 1], hg[2]);
        hg[3] = fma(jg[2 * r], rec[r], hg[3]), hg[4] = fma(jg[2 * r + 1], rec[r], hg[4]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < NV; ++e) v[e] = wave_sum(v[e]);
  if (lane == 0)
#pragma unroll
    for (int e = 0; e < NV; ++e) red[wave][e] = v[e];
  __syncthreads();
  if (tid != 0) return;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    double t = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) t += red[w][e];
    v[e] = t;
  }
  for (int d = 0; d < kb && b + d < nbias; ++d)
    for (int c = 0; c < 3; ++c) {
      const int r0 = og + 3 * b + c, c0 = og + 3 * (b + d) + c;
      Hbb[size_t(r0) * nb + c0] = gg[d], Hbb[size_t(c0) * nb + r0] = gg[d];
      const int r1 = oa + 3 * b + c, c1 = oa + 3 * (b + d) + c;
      Hbb[size_t(r1) * nb + c1] = aa[d], Hbb[size_t(c1) * nb + r1] = aa[d];
    }
  for (int c = 0; c < 3; ++c)
    for (int e = 0; e < 2; ++e) {
      Hbb[size_t(og + 3 * b + c) * nb + ogr + e] = ggr[2 * c + e], Hbb[size_t(ogr + e) * nb + og + 3 * b + c] = ggr[2 * c + e];
      Hbb[size_t(oa + 3 * b + c) * nb + ogr + e] = agr[2 * c + e], Hbb[size_t(ogr + e) * nb + oa + 3 * b + c] = agr[2 * c + e];
    }
  for (int c = 0; c < 3; ++c) gb[og + 3 * b + c] = rg[c], gb[oa + 3 * b + c] = ra[c];
  for (int e = 0; e < 5; ++e) T.gravity_part[5 * b + e] = hg[e];
}

/// Gravity-gravity block and J_g' r: sum of the per-bias-point partials in index order.
__global__ void k_border_gravity(Tables T) {
  if (T.st->done || threadIdx.x != 0) return;
  double h[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < T.n_bias; ++b)
    for (int e = 0; e < 5; ++e) h[e] += T.gravity_part[5 * b + e];
  const int nb = T.nb, ogr = 6 * T.n_bias;
  double* Hbb = T.xbuf + T.xo_bb;
  Hbb[size_t(ogr) * nb + ogr] = h[0], Hbb[size_t(ogr) * nb + ogr + 1] = h[1];
  Hbb[size_t(ogr + 1) * nb + ogr] = h[1], Hbb[size_t(ogr + 1) * nb + ogr + 1] = h[2];
  T.xbuf[T.xo_gb + ogr] = h[3], T.xbuf[T.xo_gb + ogr + 1] = h[4];
}

__global__ void __launch_bounds__(kBlock) k_border_zero(Tables T) {
  if (T.st->done) return;
  const int n = T.nb * T.nb + T.nb;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) T.xbuf[T.xo_bb + e] = 0.0;
}

/// Scaling / damping of the border blocks after the exchange:  S_pb = Sp H_pb Sb,  S_bb = Sb H_bb Sb + D_b^2,  g_b = Sb g_b.
__global__ void __launch_bounds__(kBlock) k_finalize_border(Tables T) {
  DevState* st = T.st;
  if (st->done) return;
  const int nb = T.nb, np = T.np;
  const double* X = T.xbuf;
  const bool fresh = !st->scaling_ready;
  const double radius = st->radius;
  auto sb_of = [&](int b) { return fresh ? 1.0 / (1.0 + sqrt(X[T.xo_bb + size_t(b) * nb + b])) : T.scale_b[b]; };
  auto sp_of = [&](int rho) { return fresh ? 1.0 / (1.0 + sqrt(X[T.xo_dj + rho])) : T.scale_p[rho]; };
  const int total = (np + nb) * nb;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int row = e / nb, c = e % nb;
    if (row < np) {
      T.Spb[e] = sp_of(row) * X[T.xo_pb + e] * sb_of(c);
    } else {
      const int b = row - np;
      const double sr = sb_of(b), sc = sb_of(c);
      double out = sr * sc * X[T.xo_bb + size_t(b) * nb + c];
      if (b == c) {
        const double d = X[T.xo_bb + size_t(b) * nb + b];
        if (d > 0.0) {
          const double d2 = fmin(fmax(sr * sr * d, 1e-6), 1e32) / radius;
          out += d2;
          T.D2b[b] = d2;
        } else {
          out = 1.0;
          T.D2b[b] = 0.0;
        }
        const double g = X[T.xo_gb + b];
        T.gb_s[b] = sr * g;
        if (fresh) T.scale_b[b] = sr;
        T.gabs[T.np + b] = fabs(g);
      }
      T.Sbb[size_t(b) * nb + c] = out;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Bordered solve:  [S_pp S_pb; S_bp S_bb][x_p; x_b] = [g_p; g_b] with S_pp = U'U banded.
//   Z = U^-T S_pb  (k_border_forward: one workgroup per group of border columns, column-oriented forward sweep)
//   C = S_bb - Z'Z, h = g_b - Z'y  (k_border_schur, one workgroup per border row)
//   C x_b = h (dense Cholesky in LDS), y' = y - Z x_b  (k_border_solve, one workgroup)   then the banded backward sweep on y'.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kBorderCols = 8;  // right-hand sides per workgroup in the forward sweep

__global__ void __launch_bounds__(kBlock) k_border_forward(Tables T) {  // blockDim = 64 x waves covering the 6 (bw - 1) pending rows (>= 128)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (T.st->done) return;
  const int tid = threadIdx.x;
  const int bw = T.bw, ncb = 6 * bw, nb = T.nb, np = T.np, n_blk = np / 6;
  const int c0 = blockIdx.x * kBorderCols, ncols = min(kBorderCols, nb - c0);
  double* z = smem;  // np x kBorderCols: pending right-hand side rows, overwritten by the solution
  for (int e = tid; e < np * kBorderCols; e += blockDim.x) {
    const int rho = e / kBorderCols, c = e % kBorderCols;
    z[e] = c < ncols ? T.Spb[size_t(rho) * nb + c0 + c] : 0.0;
  }
  __syncthreads();
  __shared__ double zi[6 * kBorderCols];
  const int n_pend = 6 * (bw - 1);
  // Operands of step m are requested D steps ahead (the sweep is a dependency chain over the block rows: a load issued inside
  // the step would put a full L2 round trip on it). Thread t < n_pend: the six factor entries U[6m + a][6 + t]; thread
  // (a, c) < 6 x kBorderCols: column a of W_m = U_mm^-1.
  constexpr int D = 4;
  const bool pend = tid < n_pend, diag = tid < 6 * kBorderCols;
  const int da = diag ? tid / kBorderCols : 0, dc = diag ? tid % kBorderCols : 0;
  double ur[D][6], wr[D][6];
  auto request = [&](int m, double* u, double* w) {
    const int mm = m < n_blk ? m : 0;
    const double* src = T.Ub + size_t(6 * mm) * ncb + 6 + (pend ? tid : 0);
#pragma unroll
    for (int a = 0; a < 6; ++a) u[a] = src[size_t(a) * ncb];
    // (W')[a][k] = W[k][a], k <= a ; packed index of (k, a) = k*6 - k(k-1)/2 + (a - k)
    const double* W = T.Ubk + size_t(mm) * 24;
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = W[k <= da ? k * 6 - k * (k - 1) / 2 + (da - k) : 0];
  };
#pragma unroll
  for (int d = 0; d < D; ++d) request(d, ur[d], wr[d]);
  for (int mb = 0; mb < n_blk; mb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int m = mb + d;
      if (m >= n_blk) break;
      // z_m = U_mm^-T s_m = W' s_m
      if (diag) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) v = fma(k <= da ? wr[d][k] : 0.0, z[(6 * m + k) * kBorderCols + dc], v);
        zi[tid] = v;
      }
      __syncthreads();
      if (diag) z[(6 * m + da) * kBorderCols + dc] = zi[tid];
      // pending rows of blocks m+1 .. m+bw-1: s_(i,c') -= sum_a U[6m+a][6(i-m)+c'] z_m[a]
      if (pend) {
        const int rho = 6 * (m + 1) + tid;
        if (rho < np) {
#pragma unroll
          for (int c = 0; c < kBorderCols; ++c) {
            double sacc = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) sacc = fma(ur[d][a], zi[a * kBorderCols + c], sacc);
            z[rho * kBorderCols + c] -= sacc;
          }
        }
      }
      request(m + D, ur[d], wr[d]);
      __syncthreads();
    }
  }
  for (int e = tid; e < np * kBorderCols; e += blockDim.x) {
    const int rho = e / kBorderCols, c = e % kBorderCols;
    if (c < ncols) T.Zb[size_t(rho) * nb + c0 + c] = z[e];
  }
}

/// C = S_bb - Z'Z (16 x 16 tile per workgroup, upper tile triangle mirrored) and h = g_b - Z'y. Z rows are staged through LDS
/// in chunks (coalesced, eight loads in flight per lane), the tile is accumulated from LDS.
constexpr int kSchurTile = 16, kSchurRows = 128;

__global__ void __launch_bounds__(kBlock) k_border_schur(Tables T) {
  __shared__ double za[kSchurRows][kSchurTile + 1], zc[kSchurRows][kSchurTile + 1], ys[kSchurRows];
  if (T.st->done) return;
  const int nb = T.nb, np = T.np, tid = threadIdx.x;
  const int bt = blockIdx.x, ct = blockIdx.y;
  if (ct < bt) return;  // lower tiles are written by their mirror
  const int ti = tid / kSchurTile, tj = tid % kSchurTile;
  const int b = bt * kSchurTile + ti, c = ct * kSchurTile + tj;
  double acc = 0.0, hacc = 0.0;
  for (int r0 = 0; r0 < np; r0 += kSchurRows) {
    const int nr = min(kSchurRows, np - r0);
    __syncthreads();
    // 2 x (kSchurRows x 16) operand entries + y: 16 + 1 loads per lane, issued together
    double va[8], vc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + u * kBlock, r = e / kSchurTile, k = e % kSchurTile;
      const bool ok = r < nr;
      va[u] = ok && bt * kSchurTile + k < nb ? T.Zb[size_t(r0 + r) * nb + bt * kSchurTile + k] : 0.0;
      vc[u] = ok && ct * kSchurTile + k < nb ? T.Zb[size_t(r0 + r) * nb + ct * kSchurTile + k] : 0.0;
    }
    const double yv = tid < nr ? T.ybuf[r0 + tid] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + u * kBlock, r = e / kSchurTile, k = e % kSchurTile;
      za[r][k] = va[u], zc[r][k] = vc[u];
    }
    if (tid < kSchurRows) ys[tid] = yv;
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < kSchurRows; ++r) {
      const double a = za[r][ti];
      acc = fma(a, zc[r][tj], acc);
      if (ct == bt && tj == 0) hacc = fma(a, ys[r], hacc);
    }
  }
  if (b < nb && c < nb) {
    const double v = T.Sbb[size_t(b) * nb + c] - acc;
    T.Cb[size_t(b) * nb + c] = v;
    if (ct != bt) T.Cb[size_t(c) * nb + b] = v;
  }
  if (ct == bt && tj == 0 && b < nb) T.hb[b] = T.gb_s[b] - hacc;
}

/// Dense Cholesky of the border Schur complement C (nb x nb, in LDS, augmented with h as an extra row so that the forward
/// solve comes out of the elimination), column-oriented backward solve, x_b. One barrier per column in both sweeps.
__global__ void __launch_bounds__(kBlock) k_border_solve(Tables T) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  DevState* st = T.st;
  if (st->done) return;
  const int nb = T.nb, tid = threadIdx.x;
  const int ld = nb + 1, n1 = nb + 1;  // rows 0 .. nb-1: C (lower), row nb: h'
  double* C = smem;                    // (nb + 1) x ld
  for (int e = tid; e < nb * nb; e += blockDim.x) C[(e / nb) * ld + e % nb] = T.Cb[e];
  for (int e = tid; e < nb; e += blockDim.x) C[nb * ld + e] = T.hb[e];
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();
  const int ti = tid / 16, tj = tid % 16;  // 16 x 16 lanes over the trailing (i, c) entries
  for (int j = 0; j < nb; ++j) {           // right-looking on the lower triangle; column j is scaled on the fly
    const double d = C[j * ld + j];
    if (tid == 0 && !(d > 0.0)) bad = 1;
    const double inv = 1.0 / (d > 0.0 ? d : 1.0);  // 1 / l_jj^2
    for (int i = j + 1 + ti; i < n1; i += 16) {
      const double lij = C[i * ld + j];
      for (int c = j + 1 + tj; c <= i && c < nb; c += 16) C[i * ld + c] = fma(-lij * inv, C[c * ld + j], C[i * ld + c]);
    }
    lds_barrier();
    // scale column j (not read again by later columns' updates except through these scaled values in the backward sweep)
    const double rs = sqrt(inv);
    for (int i = j + tid; i < n1; i += blockDim.x) C[i * ld + j] = i == j ? d * rs : C[i * ld + j] * rs;
    // (no barrier needed here: column j is not touched by the update of column j + 1, which reads columns > j only ... except
    //  C[c][j+1] entries, which were finalised by the update above and published by the barrier)
  }
  lds_barrier();
  // backward: L' x = y, y = row nb; column oriented, one barrier per column (x goes to its own array)
  double* y = C + nb * ld;
  double* x = C + n1 * ld;
  for (int j = nb - 1; j >= 0; --j) {
    const double xj = y[j] / C[j * ld + j];
    if (tid == 0) x[j] = xj;
    for (int i = tid; i < j; i += blockDim.x) y[i] = fma(-C[j * ld + i], xj, y[i]);
    lds_barrier();
  }
  if (tid == 0 && bad) st->chol_failed = 1;
  for (int bq = tid; bq < nb; bq += blockDim.x) T.xb[bq] = x[bq];
}

/// y' = y - Z x_b (one wave per row of Z, lanes over the border columns).
__global__ void __launch_bounds__(kBlock) k_border_apply(Tables T) {
  if (T.st->done) return;
  const int lane = threadIdx.x & 63, rho = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (rho >= T.np) return;
  double v = 0.0;
  for (int bq = lane; bq < T.nb; bq += 64) v = fma(T.Zb[size_t(rho) * T.nb + bq], T.xb[bq], v);
  v = wave_sum(v);
  if (lane == 0) T.ybuf[rho] -= v;
}

/// After the (optional) all-reduce: Jacobi scaling (fixed at iteration 0), LM diagonal, inactive coordinates.
///   S = Sp Sraw Sp + D_p^2,  g = Sp (g_p + g_schur),  g_full = Sp g_p,  D_p^2 = clamp(Sp^2 diag(J'J), 1e-6, 1e32) / radius.
/// A workgroup past the last block row (single shard without border unknowns: gridDim.x = n_cp + 1) does the work of
/// k_pack_exchange + k_cost_reduce concurrently: with nothing exchanged, neither side reads what the other writes (the block
/// rows use the radius and the scaling flag, which the bookkeeping leaves alone; `done` only makes them skip unused work).
__global__ void __launch_bounds__(kBlock) k_finalize_reduced(Tables T) {
  DevState* st = T.st;
  if (int(blockIdx.x) >= T.sp.n_cp) {
    pack_exchange_body(T, 1);
    return;
  }
  if (st->done) return;
  const int i = blockIdx.x, tid = threadIdx.x;
  const int ncb = 6 * T.bw;
  const double* X = T.xbuf;
  const double radius = st->radius;
  const bool fresh = !st->scaling_ready;
  auto scale_of = [&](int rho) { return fresh ? 1.0 / (1.0 + sqrt(X[T.xo_dj + rho])) : T.scale_p[rho]; };
  for (int e = tid; e < 6 * ncb; e += kBlock) {
    const int a = e / ncb, c = e % ncb;
    const int rho = 6 * i + a, sigma = 6 * i + c;
    double out = 0.0;
    if (sigma < T.np) {
      const double sr = scale_of(rho), sc = scale_of(sigma);
      out = sr * sc * X[size_t(rho) * ncb + c];
      if (c == a) {
        const double d = X[T.xo_dj + rho];
        if (d > 0.0) {
          const double d2 = fmin(fmax(sr * sr * d, 1e-6), 1e32) / radius;
          out += d2;
          T.D2p[rho] = d2;
        } else {  // structurally zero column (constant / unobserved): keep the system non-singular, step = 0
          out = 1.0;
          T.D2p[rho] = 0.0;
        }
      }
    }
    T.Sb[size_t(rho) * ncb + c] = out;
    if (T.Sb2 && sigma < T.np) {  // reversed copy for the far end of the two-ended factorisation: (rho, sigma) -> (np-1-sigma, np-1-rho)
      const int rv = T.np - 1 - sigma, cv = T.np - 1 - rho;
      T.Sb2[size_t(rv) * ncb + (cv - 6 * (rv / 6))] = out;
    }
  }
  if (tid < 6) {
    const int rho = 6 * i + tid;
    const double sr = scale_of(rho);
    const double gp = X[T.xo_g + rho];
    T.g_full[rho] = sr * gp;
    T.g_s[rho] = sr * (gp + X[T.xo_gs + rho]);
    if (T.Sb2) T.g2[T.np - 1 - rho] = sr * (gp + X[T.xo_gs + rho]);
    if (fresh) T.scale_p[rho] = sr;
    T.gabs[rho] = fabs(gp);
  }
}

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
  extern __shared__ __attribute__((aligned(16))) double smem[];
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
      if (l == 0) st->chol_failed = fail;
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
      rhs[m][a] = t_kk[m] == 0 ? src[a * ld + ncb] : 0.0;
    }
  };
  lds_barrier();  // initial window loaded / staged
  const bool prof = (T.debug_flags & 16) && tid == 0;
  long long* tlog = reinterpret_cast<long long*>(T.xpart);

#define UIDX(a, c) ((a) * 6 - (a) * ((a)-1) / 2 + ((c) - (a)))
  for (int i = 0; i < n_blk; ++i) {
    const int si = i % bw;
    if (prof) tlog[8 * i + 0] = wall_clock64();
    // ---- publish the pivot row, then refill the freed registers with block row i + bw ----
#pragma unroll
    for (int m = 0; m < TPT; ++m)
      if (t_ok[m] && t_slot[m] == si) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = 0; c < 6; c += 2)
            *reinterpret_cast<double2*>(&rowbuf[a * ld + 6 * t_kk[m] + c]) = make_double2(acc[m][6 * a + c], acc[m][6 * a + c + 1]);
          if (t_kk[m] == 0) rowbuf[a * ld + ncb] = rhs[m][a];
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
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 0; c < 6; ++c) acc[m][6 * r + c] = fma(-xa[r], xb[c], acc[m][6 * r + c]);
        if (t_kk[m] == 0) {
          const double y = xbuf[a * ld + ncb];
#pragma unroll
          for (int r = 0; r < 6; ++r) rhs[m][r] = fma(-xa[r], y, rhs[m][r]);
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
// LDS (doubles): rowbuf 2 x 6 x ld | xbuf 2 x 6 x ld | stage 2 x 6 x ld | y np | diagonal scratch 36.
//
// Two-ended mode (grid = 2, visual-only systems): the chain over the block rows is halved by eliminating from both ends at once.
// Workgroup 1 factors the REVERSED system (written by k_finalize_reduced next to the natural one) for the last n - m - w block rows (w = bw - 1), dumps its trailing
// window (the Schur contribution of those rows to the middle block rows m .. m + w - 1) and raises a flag. Workgroup 0 factors
// rows 0 .. m - 1, waits for the flag, adds the other end's contribution to its own trailing window (entries of the middle
// rows that couple to the eliminated end become zero) and simply continues through the middle rows: it ends with the Cholesky
// factor of the system in which the far end has been eliminated. k_band_backward2 solves the top part normally and the bottom
// part in reversed coordinates once the middle solution is known.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kLaCompute = 192;
constexpr int kLaThreads = kLaCompute + 3 * 64;

template <int TPT>  // tiles per compute thread: bw * (bw - 2) <= TPT * kLaCompute
__global__ void __launch_bounds__(kLaThreads) k_band_factor_la(Tables T) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
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
    while (__hip_atomic_load(T.join_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < T.join_epoch) __builtin_amdgcn_s_sleep(8);
  };
  const int tid = threadIdx.x;
  constexpr int nthr = kLaCompute;
  constexpr int PC = 2;  // columns of the pivot row per panel lane: 6 * bw + 1 <= 128 (bw <= 20)
  const int bw = T.bw, ncb = 6 * bw, ld = ncb + 2;
  const int n_blk = T.np / 6;
  double* rowbuf = smem;            // row r (published by its owners, updated through X_(r-2)) in rowbuf[r & 1]
  double* xbuf = smem + 12 * ld;    // [U_rr | X_r | y_r] in xbuf[r & 1]
  double* stage = smem + 24 * ld;   // block row r staged by the loader in stage[r & 1]
  double* xs = smem + 36 * ld;      // np : y (forward solve)
  double* dscr = xs + T.np;         // 36 : updated diagonal block of the panel row
  double* dinv = dscr + 36;         // 2 x 6 : 1 / diag(U_rr) in dinv[r & 1]
  __shared__ int fail;
  if (tid == 0) fail = 0;
  const int wave = tid >> 6, l = tid & 63;
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
      put(va, stage + ((bw + 2) & 1) * 6 * ld);
      fetch(va, bw + 4);
      lds_barrier();  // init
      lds_barrier();  // prologue
      for (int i = 0; i < n_steps; i += 2) {
        put(vb, stage + ((i + 3 + bw) & 1) * 6 * ld);
        tput(tb, rowbuf + (i & 1) * 6 * ld, i + 2);
        fetch(vb, i + 5 + bw);
        tfetch(tb, i + 4);
        if ((T.debug_flags & 16) && l == 0) reinterpret_cast<long long*>(T.xpart)[8 * i + 7] = wall_clock64();
        lds_barrier();
        junction_io(i);
        if (i + 1 < n_steps) {
          put(va, stage + ((i + 4 + bw) & 1) * 6 * ld);
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
        if ((T.debug_flags & 16) && l == 0) reinterpret_cast<long long*>(T.xpart)[8 * i + 6] = wall_clock64();
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
      if (l == 0 && fail) st->chol_failed = 1;  // (cleared by k_finalize_reduced)
    }
    return;
  }

  if (wave == 3) {  // ================================ panel wave (alone on SIMD 3) ================================
#define UIDX(a, c) ((a) * 6 - (a) * ((a)-1) / 2 + ((c) - (a)))
    const bool pprof = (T.debug_flags & 16) && l == 0;
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
      if (l < 6) {
#pragma unroll
        for (int a = 0; a < 6; ++a) dscr[6 * a + l] = v[0][a];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave: LDS is in order, only the compiler must not reorder
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
      for (int a = 0; a < 6; ++a) dinv[(r & 1) * 6 + a] = inv[a];  // every lane, same value
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
            J.win[size_t(6 * w_mid - 1 - a) * (ncb + 1) + ncb] = v - J.g_s[6 * r + a];
            J.win[size_t(6 * w_mid - 1 - (6 + a)) * (ncb + 1) + ncb] = row1[a * ld + c] - J.g_s[6 * (r + 1) + a];
          } else {
            hand_over(a, c, v - J.Sb[size_t(6 * r + a) * ncb + c]);
            hand_over(6 + a, c, row1[a * ld + c] - J.Sb[size_t(6 * (r + 1) + a) * ncb + c]);
          }
        }
      }
      __threadfence();
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
  int t_kk[TPT], t_row[TPT];
  bool t_ok[TPT];
  double acc[TPT][36], rhs[TPT][6];
#pragma unroll
  for (int m = 0; m < TPT; ++m) {
    const int tl = tid + m * nthr;
    t_ok[m] = tl < bw * (bw - 2);
    const int slot = t_ok[m] ? tl / (bw - 2) : 0;
    t_kk[m] = t_ok[m] ? tl % (bw - 2) : 0;
    t_row[m] = slot < 2 ? slot + bw : slot;  // rows 0 and 1 start in LDS (loader); their slots prefetch rows bw, bw + 1
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();  // init
  lds_barrier();  // prologue: X_0 complete
  const bool prof = (T.debug_flags & 16) && tid == 0;
  long long* tlog = reinterpret_cast<long long*>(T.xpart);
  for (int i = 0; i < n_steps; ++i) {
    if (prof) tlog[8 * i + 0] = wall_clock64();
    const double* xb = xbuf + (i & 1) * 6 * ld;
#pragma unroll
    for (int m = 0; m < TPT; ++m) {
      const int j = t_row[m] - i;
      if (!t_ok[m] || j < 2 || j > bw - 1) continue;
      if (j + t_kk[m] <= bw - 1 && !(T.debug_flags & 2)) {  // rank-6 update  S_(i+j),kk -= X_j' X_(j+kk)
        const int ca = 6 * j, cb = 6 * (j + t_kk[m]);
        // software pipelined over the six rows of X: the operands of row a + 1 are requested before the 36 FMAs of row a, and
        // the scheduler may not hoist more than that (all 72 operands in flight at once spills the register tiles at TPT = 2)
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
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[m][6 * r + c] = fma(-xa[b][r], xc[b][c], acc[m][6 * r + c]);
          if (t_kk[m] == 0) {
            const double y = xb[a * ld + ncb];
#pragma unroll
            for (int r = 0; r < 6; ++r) rhs[m][r] = fma(-xa[b][r], y, rhs[m][r]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (j == 2) {  // row i + 2 becomes the panel row of the next step: publish, then prefetch row i + 2 + bw into the registers
        double* dst = rowbuf + (i & 1) * 6 * ld;
        const double* src = stage + ((i + 2 + bw) & 1) * 6 * ld;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = 0; c < 6; c += 2) {
            *reinterpret_cast<double2*>(&dst[a * ld + 6 * t_kk[m] + c]) = make_double2(acc[m][6 * a + c], acc[m][6 * a + c + 1]);
            const double2 t = *reinterpret_cast<const double2*>(&src[a * ld + 6 * t_kk[m] + c]);
            acc[m][6 * a + c] = t.x, acc[m][6 * a + c + 1] = t.y;
          }
          if (t_kk[m] == 0) dst[a * ld + ncb] = rhs[m][a];
          rhs[m][a] = t_kk[m] == 0 ? src[a * ld + ncb] : 0.0;
        }
        t_row[m] = i + 2 + bw;
      }
    }
    if (prof) tlog[8 * i + 1] = wall_clock64();
    lds_barrier();
    if (m_at >= 0 && i + 1 == m_at) {  // ---- junction: add the other end's Schur contribution to the middle rows ----
      junction_wait();
      const int dm = 6 * w_mid, wl = ncb + 1;
      const double* WD = J.win;  // correction in this job's own band layout, local to the middle rows (see hand_over)
      // rows m and m + 1 sit in LDS
      for (int e = tid; e < 2 * 6 * (ncb + 1); e += nthr) {
        const int jr = e / (6 * (ncb + 1)), rem = e % (6 * (ncb + 1)), a = rem / (ncb + 1), c = rem % (ncb + 1);
        double* dst = rowbuf + ((m_at + jr) & 1) * 6 * ld + a * ld + c;
        const double d = WD[size_t(6 * jr + a) * wl + c];
        *dst = (c == ncb || 6 * jr + c < dm) ? *dst + d : 0.0;  // columns beyond the middle couple to the eliminated end
      }
#pragma unroll
      for (int m = 0; m < TPT; ++m) {
        const int jr = t_row[m] - m_at;
        if (!t_ok[m] || jr < 2 || jr >= w_mid) continue;
        const bool inside = jr + t_kk[m] <= w_mid - 1;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = 0; c < 6; ++c) acc[m][6 * a + c] = inside ? acc[m][6 * a + c] + WD[size_t(6 * jr + a) * wl + 6 * t_kk[m] + c] : 0.0;
          if (t_kk[m] == 0) rhs[m][a] += WD[size_t(6 * jr + a) * wl + ncb];
        }
      }
      lds_barrier();  // merge done
      lds_barrier();  // panel(m) done
    }
  }
  lds_barrier();
  if (dump) {  // rows n_steps + 2 .. n_steps + w - 1 of the trailing window live in the register tiles
#pragma unroll
    for (int m = 0; m < TPT; ++m) {
      const int jr = t_row[m] - n_steps;
      if (!t_ok[m] || jr < 2 || jr >= w_mid) continue;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = 0; c < 6; ++c)
          hand_over(6 * jr + a, 6 * t_kk[m] + c, acc[m][6 * a + c] - J.Sb[size_t(6 * t_row[m] + a) * ncb + 6 * t_kk[m] + c]);
        if (t_kk[m] == 0) J.win[size_t(6 * w_mid - 1 - (6 * jr + a)) * (ncb + 1) + ncb] = rhs[m][a] - J.g_s[6 * t_row[m] + a];
      }
    }
    __threadfence();
    lds_barrier();
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
  int n_rows;   // block rows of this factor
  int given;    // block rows above them in sweep order whose solution comes from the other job
  int reversed; // solution index = np - 1 - rho
};

__global__ void __launch_bounds__(kCholThreads) k_band_backward2(Tables T, BackJob j0, BackJob j1, int m_mid) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  DevState* st = T.st;
  if (st->done) return;
  const BackJob J = blockIdx.x == 0 ? j0 : j1;
  const int tid = threadIdx.x;
  constexpr int nthr = kCholThreads;
  const int bw = T.bw, ncb = 6 * bw, np = T.np;
  const int n_own = 6 * J.n_rows, n_all = 6 * (J.n_rows + J.given);
  double* xs = smem;          // n_all : pending rows (own) / given solution
  double* xout = smem + n_all;  // n_own : solution of the own rows (flushed to T.xsol at the end / when the middle is complete)
  __shared__ double Wl[2][24];
  const int n_above = 6 * (bw - 1);
  // The given block rows have no dependencies among themselves: their whole contribution to the pending rows is one
  // (n_above x n_above) matrix-vector product. The matrix block G = U(own rows n_own - n_above .., given columns) does not depend on
  // the other sweep, so it is brought into LDS (transposed: G[c][r], odd leading dimension) BEFORE waiting for the flag; once the
  // middle solution is there, one pass replaces `given` sequential steps of the sweep.
  const bool merged = J.given > 0 && 6 * J.given == n_above && n_own >= n_above;
  const int ldg = n_above | 1;
  double* G = smem + 2 * np;
  for (int rho = tid; rho < n_own; rho += nthr) xs[rho] = J.ybuf[rho];
  if (merged) {
    const int n_g = n_above * n_above;
    for (int e0 = tid; e0 < n_g; e0 += 8 * nthr) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * nthr, r = e / n_above, c = e - r * n_above;
        const int rho = n_own - n_above + r, off = n_own + c - 6 * (rho / 6);  // band offset of column n_own + c in row rho
        v[u] = (e < n_g && off < ncb) ? J.Ub[size_t(rho) * ncb + off] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * nthr, r = e / n_above, c = e - r * n_above;
        if (e < n_g) G[c * ldg + r] = v[u];
      }
    }
  }
  auto load_u = [&](int j, double* u) {
    const int rho = 6 * j - 1 - tid;
    const bool ok = j >= 0 && tid < n_above && rho >= 0 && rho < n_own;
    const double* src = J.Ub + (ok ? size_t(rho) * ncb + (6 * j - 6 * (rho / 6)) : 0);
#pragma unroll
    for (int a = 0; a < 6; ++a) u[a] = ok ? src[a] : 0.0;
  };
  auto load_w = [&](int j) -> double { return (j >= 0 && j < J.n_rows && tid < 21) ? J.Ubk[size_t(j) * 24 + tid] : 0.0; };
  const int jtop = merged ? J.n_rows - 1 : J.n_rows + J.given - 1;
  double u0[6], u1[6], u2[6], u3[6], w0, w1, w2, w3;
  load_u(jtop, u0), load_u(jtop - 1, u1), load_u(jtop - 2, u2);
  w0 = load_w(jtop), w1 = load_w(jtop - 1), w2 = load_w(jtop - 2);
  if (J.given) {  // wait for the middle solution
    while (__hip_atomic_load(T.join_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < T.join_epoch) __builtin_amdgcn_s_sleep(8);
    for (int rho = n_own + tid; rho < n_all; rho += nthr) xs[rho] = T.xsol[J.reversed ? np - 1 - rho : rho];
  }
  __syncthreads();
  if (merged) {
    if (tid < n_above) {
      double acc = 0.0;
      for (int c = 0; c < n_above; ++c) acc = fma(G[c * ldg + tid], xs[n_own + c], acc);
      xs[n_own - n_above + tid] -= acc;
    }
    __syncthreads();
  }
  // one block row of the sweep; `own` is a compile-time tag so that the hot loops below carry no extra control flow
  auto step = [&](int j, auto own_tag) {
    constexpr bool own = decltype(own_tag)::value;
    if (own && tid < 21) Wl[j & 1][tid] = w0;
    load_u(j - 3, u3), w3 = load_w(j - 3);
    lds_barrier();
    double y[6], x[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) y[a] = xs[6 * j + a];
    if (own) {
      const double* W = Wl[j & 1];
      int pidx = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double v = 0.0;
#pragma unroll
        for (int c = a; c < 6; ++c) v = fma(W[pidx++], y[c], v);
        x[a] = v;
      }
      if (tid < 6) xout[6 * j + tid] = x[tid];
    } else {
#pragma unroll
      for (int a = 0; a < 6; ++a) x[a] = y[a];  // given by the other sweep
    }
    const int rho_p = 6 * j - 1 - tid;
    if (tid < n_above && rho_p >= 0 && rho_p < n_own) {
      double sacc = 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a) sacc = fma(u0[a], x[a], sacc);
      xs[rho_p] -= sacc;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) u0[a] = u1[a], u1[a] = u2[a], u2[a] = u3[a];
    w0 = w1, w1 = w2, w2 = w3;
  };
  for (int j = jtop; j >= J.n_rows; --j) step(j, std::false_type{});  // (not merged: given rows one by one)
  const int j_pub = (blockIdx.x == 0 && m_mid >= 0) ? m_mid : 0;  // block 0 publishes the middle solution after block row m_mid
  for (int j = J.n_rows - 1; j >= j_pub; --j) step(j, std::true_type{});
  if (blockIdx.x == 0 && m_mid >= 0) {
    lds_barrier();
    for (int rho = 6 * m_mid + tid; rho < n_own; rho += nthr) T.xsol[rho] = xout[rho];
    __threadfence();
    lds_barrier();
    if (tid == 0) {
      __threadfence();
      __hip_atomic_store(T.join_flag, T.join_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int j = m_mid - 1; j >= 0; --j) step(j, std::true_type{});
  }
  __syncthreads();
  const int flush_to = (blockIdx.x == 0 && m_mid >= 0) ? 6 * m_mid : n_own;  // (the middle rows of block 0 are already out)
  for (int rho = tid; rho < flush_to; rho += nthr) T.xsol[J.reversed ? np - 1 - rho : rho] = xout[rho];
  if (gridDim.x == 1) return;  // (A/B runs on the whole system: k_step_outputs follows)
  // the block that finishes last turns the solution into the step outputs (saves a launch); join_flag[1] advances by two per launch
  __shared__ int is_last;
  __threadfence();
  __syncthreads();
  if (tid == 0) is_last = (atomicAdd(T.join_flag + 1, 1u) & 1u) == 1u;
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  __shared__ double red[kCholThreads / 64];
  double gd = 0.0, dd = 0.0;
  for (int rho = tid; rho < np; rho += nthr) {
    const double step = -__builtin_nontemporal_load(T.xsol + rho);
    T.step_p[rho] = step;
    T.delta_p[rho] = T.scale_p[rho] * step;
    gd = fma(T.g_full[rho], step, gd);
    dd = fma(T.D2p[rho] * step, step, dd);
  }
  gd = block_sum(gd, red);
  dd = block_sum(dd, red);
  if (tid == 0) {
    st->g_dot_step_pose = gd;
    st->d2_step2_pose = dd;
  }
}


/// step = -x, delta = scale o step and the pose-side reductions of the model cost change, from T.xsol (two-ended path).
__global__ void __launch_bounds__(kBlock) k_step_outputs(Tables T) {
  __shared__ double red[kBlock / 64];
  DevState* st = T.st;
  if (st->done) return;
  double gd = 0.0, dd = 0.0;
  for (int rho = threadIdx.x; rho < T.np; rho += kBlock) {
    const double step = -T.xsol[rho];
    T.step_p[rho] = step;
    T.delta_p[rho] = T.scale_p[rho] * step;
    gd = fma(T.g_full[rho], step, gd);
    dd = fma(T.D2p[rho] * step, step, dd);
  }
  gd = block_sum(gd, red);
  dd = block_sum(dd, red);
  if (threadIdx.x == 0) {
    st->g_dot_step_pose = gd;
    st->d2_step2_pose = dd;
  }
}

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
  extern __shared__ __attribute__((aligned(16))) double smem[];
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
  if (tid == 0) st->chol_failed = fail;
}

/// Backward sweep U x = y (y in T.ybuf, possibly corrected by the border solve) + step outputs and model-cost reductions.
__global__ void __launch_bounds__(kCholThreads) k_band_backward(Tables T) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  DevState* st = T.st;
  if (st->done) return;
  const int tid = threadIdx.x;
  constexpr int nthr = kCholThreads;
  const int bw = T.bw, ncb = 6 * bw;
  const int n_blk = T.np / 6;
  double* xs = smem;         // np : pending rows
  double* xout = xs + T.np;  // np : final x
  __shared__ double Wl[2][24];
  for (int rho = tid; rho < T.np; rho += nthr) xs[rho] = T.ybuf[rho];
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
  double u0[6], u1[6], u2[6], u3[6], w0, w1, w2, w3;
  load_u(n_blk - 1, u0), load_u(n_blk - 2, u1), load_u(n_blk - 3, u2);
  w0 = load_w(n_blk - 1), w1 = load_w(n_blk - 2), w2 = load_w(n_blk - 3);
  for (int j = n_blk - 1; j >= 0; --j) {
    if (tid < 21) Wl[j & 1][tid] = w0;
    load_u(j - 3, u3), w3 = load_w(j - 3);
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
      for (int a = 0; a < 6; ++a) sacc = fma(u0[a], x[a], sacc);
      xs[6 * j - 1 - tid] -= sacc;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) u0[a] = u1[a], u1[a] = u2[a], u2[a] = u3[a];
    w0 = w1, w1 = w2, w2 = w3;
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
// Candidate point of the step, one launch. Workgroups [0, n_lm_part): landmark back-substitution (one wave per landmark):
//   y_l = L^-T (yh_l - Yh_l' (Sp o y_p)),  step_l = -y_l, with y_p = -step_p;   candidate = lm + S_l o step_l,
// with the landmark-side terms of the decision (|x|^2, |x - x+|^2, g.step, step'D^2 step) summed per workgroup in a fixed
// order. Workgroups [n_lm_part, n_lm_part + n_norm_part): candidate control points / bias points / gravity = Plus(x, delta) per
// Ceres manifold (quaternion left-multiplicative, R^3 additive, stamp constant, sphere) and their norms.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_backsub_retract(Tables T) {
  if (T.st->done) return;
  __shared__ double red[kBlock / 64][4];
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
          yh[a] = T.lm_yhat[3 * dl + a], x[a] = T.lm[3 * dl + a], sc[a] = T.lm_scale[3 * dl + a];
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
    const double* x = T.cp + 8 * j;
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
  st->cost = c;
  st->gmax = gm;
  st->chol_failed = 0;    // raised by the factorisation kernels of this iteration
  if (set_scaling_ready) st->scaling_ready = 1;  // Jacobi scaling is computed at iteration 0 only (else: set by decide_step)
  if (st->iteration == 0) {
    hs_iteration& r = st->records[0];
    r.iteration = 0, r.step_is_valid = 1, r.step_is_successful = 1, r.cost = c, r.cost_change = 0, r.gradient_max_norm = gm;
    r.step_norm = 0, r.relative_decrease = 0, r.radius = st->radius;
    st->iteration = 1;
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

HSD void decide_step(const Tables& T);

/// Second exchange buffer (5 doubles, additive across shards): candidate cost, |x|^2, |x - x+|^2 and the landmark-side
/// terms of the model cost change. The replicated control-point part of the norms is contributed by rank 0 only.
__global__ void __launch_bounds__(kBlock) k_pack_decision(Tables T, int decide_here /* no exchange between packing and deciding */) {
  __shared__ double red[kBlock / 64];
  DevState* st = T.st;
  if (st->done) return;
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
  cand = block_sum(cand, red), xs = block_sum(xs, red), ss = block_sum(ss, red), gd = block_sum(gd, red), dd = block_sum(dd, red);
  if (threadIdx.x == 0) {
    double* D = T.xbuf + T.xo_dec;
    D[0] = cand, D[1] = xs, D[2] = ss, D[3] = gd, D[4] = dd;
    if (decide_here) decide_step(T);
  }
}

/// Trust-region decision of one LM iteration (single lane): step quality, acceptance, radius update, termination tests.
HSD void decide_step(const Tables& T) {
  DevState* st = T.st;
  const double* D = T.xbuf + T.xo_dec;
  const double cand = D[0], xs = D[1], ss = D[2];
  // model_cost_change = -g.step/2 + step'D^2 step/2 (exact for the solved system; TrustRegionMinimizer evaluates
  // -(J step).(r + J step/2), identical algebraically)
  const double g_step = st->g_dot_step_pose + D[3], d_step = st->d2_step2_pose + D[4];
  const double mcc = -0.5 * g_step + 0.5 * d_step;
  st->model_cost_change = mcc;
  st->scaling_ready = 1;  // a step was computed: the Jacobi scaling of this solve is fixed from here on
  st->step_valid = (isfinite(mcc) && !st->chol_failed && mcc >= 0.0) ? 1 : 0;
  const int it = st->iteration;
  hs_iteration& r = st->records[it];
  r.iteration = it, r.cost = st->cost, r.cost_change = 0, r.gradient_max_norm = st->gmax, r.step_norm = 0, r.relative_decrease = 0;
  r.step_is_valid = st->step_valid, r.step_is_successful = 0;
  st->num_iterations = it;
  st->accepted = 0;
  st->gmax_bits = 0ull, st->gmax_pose_bits = 0ull;  // the next linearisation re-accumulates them
  if (!st->step_valid) {  // HandleInvalidStep
    if (++st->invalid_streak >= 5) {
      st->done = 1, st->termination = HS_FAILURE;
    } else {
      st->radius *= 0.5;
    }
    r.radius = st->radius;
    st->iteration = it + 1;
    return;
  }
  st->invalid_streak = 0;
  st->cand_cost = cand;
  r.step_norm = sqrt(ss);
  // ParameterToleranceReached
  if (r.step_norm <= 1e-8 * (sqrt(xs) + 1e-8)) {
    st->done = 1, st->termination = HS_CONVERGENCE;
    r.radius = st->radius;
    return;
  }
  // FunctionToleranceReached
  r.cost_change = st->cost - cand;
  if (fabs(r.cost_change) <= 1e-6 * st->cost) {
    st->done = 1, st->termination = HS_CONVERGENCE;
    r.radius = st->radius;
    return;
  }
  r.relative_decrease = (st->cost - cand) / st->model_cost_change;
  if (r.relative_decrease > 1e-3) {  // HandleSuccessfulStep
    r.step_is_successful = 1;
    st->accepted = 1;
    st->num_successful++;
    st->cost = cand;
    r.cost = cand;
    const double q = 2.0 * r.relative_decrease - 1.0;
    st->radius = fmin(1e16, st->radius / fmax(1.0 / 3.0, 1.0 - q * q * q));
    st->decrease_factor = 2.0;
  } else {
    st->radius = st->radius / st->decrease_factor;
    st->decrease_factor *= 2.0;
  }
  r.radius = st->radius;
  st->iteration = it + 1;
}

__global__ void __launch_bounds__(kBlock) k_decide(Tables T) {
  if (T.st->done || threadIdx.x != 0) return;
  decide_step(T);
}

/// x <- candidate when the step was accepted.
__global__ void __launch_bounds__(kBlock) k_commit(Tables T) {
  // note: reads `accepted` even when `done` was just set by a convergence test (those leave accepted = 0)
  if (!T.st->accepted) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < 8 * T.sp.n_cp) T.cp[idx] = T.cp_cand[idx];
  for (int l = idx; l < 3 * T.n_lm; l += gridDim.x * blockDim.x) T.lm[l] = T.lm_cand[l];
  if (T.nb > 0) {
    for (int e = idx; e < 4 * T.n_bias; e += gridDim.x * blockDim.x) T.bias_g[e] = T.bias_g_cand[e], T.bias_a[e] = T.bias_a_cand[e];
    if (idx < 3) T.gravity[idx] = T.gravity_cand[idx];
  }
}

/// Batched trajectory sampling (state.evaluate(StateQuery{t, derivative}) loop of apps/hyperslam/main.cpp:72-79):
/// pose n x 7, velocity / acceleration n x 6 [angular (body) ; linear (world)], nullable.
template <int K>
__global__ void __launch_bounds__(kBlock) k_sample_trajectory(Tables T, int n, const double* stamps, double* pose, double* vel, double* acc) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* cps = smem;
  stage_cps(T.cp, cps, 8 * T.sp.n_cp);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double u;
  const int first = segment_of(stamps[i], T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dlam[K], ddlam[K];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dlam, ddlam, 2);
  SplineFull<K> S;
  spline_full<K, false>(cps + 8 * first, lam, dlam, ddlam, &S);
  double* o = pose + 7 * i;
  o[0] = S.q.x, o[1] = S.q.y, o[2] = S.q.z, o[3] = S.q.w, o[4] = S.p.x, o[5] = S.p.y, o[6] = S.p.z;
  if (vel) vel[6 * i] = S.w.x, vel[6 * i + 1] = S.w.y, vel[6 * i + 2] = S.w.z, vel[6 * i + 3] = S.v.x, vel[6 * i + 4] = S.v.y, vel[6 * i + 5] = S.v.z;
  if (acc) acc[6 * i] = S.al.x, acc[6 * i + 1] = S.al.y, acc[6 * i + 2] = S.al.z, acc[6 * i + 3] = S.a.x, acc[6 * i + 4] = S.a.y, acc[6 * i + 5] = S.a.z;
}

/// Pixel -> unit bearing in the sensor frame (radtan undistortion by fixed-point iteration; cam = [T_bs(7) | cx cy fx fy | k1 k2 p1 p2]).
HSD V3 pixel_to_bearing(const double* cam, double u, double v) {
  const double xd = (u - cam[7]) / cam[9], yd = (v - cam[8]) / cam[10];
  const double k1 = cam[11], k2 = cam[12], p1 = cam[13], p2 = cam[14];
  double x = xd, y = yd;
  for (int it = 0; it < 20; ++it) {
    const double r2 = x * x + y * y, rad = 1 + k1 * r2 + k2 * r2 * r2;
    const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x), dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
    x = (xd - dx) / rad, y = (yd - dy) / rad;
  }
  const double n = sqrt(x * x + y * y + 1);
  return V3{x / n, y / n, 1 / n};
}

/// AbstractOptimizer::process(VisualTracks) front half (abstract.cpp:197-223,250-255): one stereo track per lane.
template <int K>
__global__ void __launch_bounds__(kBlock) k_process_tracks(Tables T, double stamp, int n, const double* px0, const double* px1, double* b0o, double* b1o,
                                                           double* pwo) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* cps = smem;
  stage_cps(T.cp, cps, 8 * T.sp.n_cp);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* c0 = T.cam, *c1 = T.cam + 16;
  const V3 b0 = pixel_to_bearing(c0, px0[2 * i], px0[2 * i + 1]), b1 = pixel_to_bearing(c1, px1[2 * i], px1[2 * i + 1]);
  if (b0o) b0o[3 * i] = b0.x, b0o[3 * i + 1] = b0.y, b0o[3 * i + 2] = b0.z;
  if (b1o) b1o[3 * i] = b1.x, b1o[3 * i + 1] = b1.y, b1o[3 * i + 2] = b1.z;
  if (!pwo) return;
  // T_wb(stamp), T_w0 = T_wb o T_b0, T_01 = T_b0^-1 o T_b1
  double u;
  const int first = segment_of(stamp, T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dlam[K], ddlam[K];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dlam, ddlam, 0);
  Quat q_wb;
  V3 p_wb;
  spline_pose<K>(cps + 8 * first, lam, &q_wb, &p_wb);
  const Quat q_b0 = load_quat(c0), q_b1 = load_quat(c1);
  const V3 t_b0 = V3{c0[4], c0[5], c0[6]}, t_b1 = V3{c1[4], c1[5], c1[6]};
  const M3 R_wb = qmat(q_wb), R_b0 = qmat(q_b0), R_b1 = qmat(q_b1);
  const M3 R_01 = mul_tn(R_b0, R_b1);
  const V3 o = mul_t(R_b0, t_b1 - t_b0);  // origin of camera 1 in frame 0
  const V3 d1 = mul(R_01, b1);
  const double a = dot(b0, b0), b = dot(b0, d1), c = dot(d1, d1), e = dot(b0, o), f = dot(d1, o);
  const double den = a * c - b * b;
  const double s0 = den > 1e-12 ? (c * e - b * f) / den : 1.0, s1 = den > 1e-12 ? (b * e - a * f) / den : 1.0;
  const V3 p0 = 0.5 * (s0 * b0 + o + s1 * d1);  // midpoint of the two rays, frame 0
  const V3 pb = mul(R_b0, p0) + t_b0;
  const V3 pw = mul(R_wb, pb) + p_wb;
  pwo[3 * i] = pw.x, pwo[3 * i + 1] = pw.y, pwo[3 * i + 2] = pw.z;
}

/// Fresh trust-region state (LevenbergMarquardtStrategy: initial radius 1e4, decrease factor 2).
__global__ void k_reset_state(DevState* st, int max_iterations, double radius) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->radius = radius, st->decrease_factor = 2.0;
  st->cost = st->cand_cost = st->model_cost_change = 0.0;
  st->gmax_bits = 0ull, st->gmax_pose_bits = 0ull, st->gmax = 0.0, st->x_sqnorm = st->step_sqnorm = 0.0;
  st->g_dot_step_pose = st->d2_step2_pose = 0.0;
  st->iteration = 0, st->done = 0, st->termination = HS_NO_CONVERGENCE, st->accepted = 0, st->step_valid = 0;
  st->invalid_streak = 0, st->num_successful = 0, st->num_iterations = 0, st->scaling_ready = 0;
  st->max_iterations = max_iterations, st->chol_failed = 0;
}

/// Batched Manifold::Plus / PlusJacobian of the variable classes on the path (hs_manifold_plus*): the same device
/// functions k_backsub_retract and the local-coordinate Jacobians use. One element per lane. kind: HS_MANIFOLD_* of the C ABI.
__global__ void __launch_bounds__(kBlock) k_manifold_plus(int kind, int ambient, int tangent, int n, const double* __restrict__ x,
                                                          const double* __restrict__ d, double* __restrict__ out, double* __restrict__ jac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* xi = x + size_t(i) * ambient;
  if (out) {
    const double* di = d + size_t(i) * tangent;
    double* o = out + size_t(i) * ambient;
    switch (kind) {
      case 1:
        for (int c = 0; c < ambient; ++c) o[c] = xi[c] + di[c];
        break;
      case 2:
      case 3: {
        const Quat q = quat_plus(Quat{xi[0], xi[1], xi[2], xi[3]}, V3{di[0], di[1], di[2]});
        o[0] = q.x, o[1] = q.y, o[2] = q.z, o[3] = q.w;
        o[4] = xi[4] + di[3], o[5] = xi[5] + di[4], o[6] = xi[6] + di[5];
        if (kind == 2) o[7] = xi[7];
        break;
      }
      case 4: sphere_plus(xi, di, o); break;
      case 5:
        o[0] = xi[0] + di[0], o[1] = xi[1] + di[1], o[2] = xi[2] + di[2], o[3] = xi[3];
        break;
      default:
        for (int c = 0; c < ambient; ++c) o[c] = xi[c];
    }
  }
  if (jac && tangent > 0) {
    double* J = jac + size_t(i) * ambient * tangent;
    for (int e = 0; e < ambient * tangent; ++e) J[e] = 0.0;
    switch (kind) {
      case 1:
        for (int c = 0; c < ambient; ++c) J[c * tangent + c] = 1.0;
        break;
      case 2:
      case 3: {
        const Quat q = Quat{xi[0], xi[1], xi[2], xi[3]};
        // column c = d/d delta_c of [delta ; 1] (x) q at delta = 0 = (e_c, 0) (x) q
        const Quat c0 = qmul(Quat{1, 0, 0, 0}, q), c1 = qmul(Quat{0, 1, 0, 0}, q), c2 = qmul(Quat{0, 0, 1, 0}, q);
        J[0 * 6 + 0] = c0.x, J[1 * 6 + 0] = c0.y, J[2 * 6 + 0] = c0.z, J[3 * 6 + 0] = c0.w;
        J[0 * 6 + 1] = c1.x, J[1 * 6 + 1] = c1.y, J[2 * 6 + 1] = c1.z, J[3 * 6 + 1] = c1.w;
        J[0 * 6 + 2] = c2.x, J[1 * 6 + 2] = c2.y, J[2 * 6 + 2] = c2.z, J[3 * 6 + 2] = c2.w;
        J[4 * 6 + 3] = 1.0, J[5 * 6 + 4] = 1.0, J[6 * 6 + 5] = 1.0;
        break;
      }
      case 4: sphere_plus_jacobian(xi, J); break;
      case 5: J[0 * 3 + 0] = 1.0, J[1 * 3 + 1] = 1.0, J[2 * 3 + 2] = 1.0; break;
      default: break;
    }
  }
}

}  // namespace hs
