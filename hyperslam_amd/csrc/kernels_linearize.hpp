// kernels_linearize.hpp — linearisation and candidate-cost kernels (one residual block per lane) (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include "kernels_common.hpp"

namespace hs {

// ---------------------------------------------------------------------------------------------------------------------
// Linearisation
// ---------------------------------------------------------------------------------------------------------------------
/// out_rec/out_pos: where the record of residual q goes (solver: T.v_rec at T.v_pos[q]; debug export: table order).
/// Records are transposed through a per-wave LDS slab so that the scattered 448-byte (k = 4) records leave the CU as full
/// 16-byte-per-lane stores (7 cache lines per record instead of 56 partial-line writes).
template <int K>
constexpr int lin_block() { return K <= 4 ? 256 : 128; }  // 2 waves at k = 6: the record slab is 42 KB per wave

/// Speculative use (cp_src / lm_src = the candidate point, out_rec = nullptr): the records go to the visual record buffer that does NOT hold
/// the linearisation of the current point, the cost partials are the candidate's — if the step is accepted, the next iteration starts
/// from these records without linearising again (launch_update, decide_step).
HSD const double* current_visual_records(const Tables& T) { return T.st->rec_sel ? T.v_rec_alt : T.v_rec; }

template <int K>
__global__ void __launch_bounds__(lin_block<K>()) k_linearize_visual(Tables T, double* out_rec, const int* out_pos, int robustify,
                                                                     double* cost_part, double* cost_each, const double* cp_src = nullptr,
                                                                     const double* lm_src = nullptr) {
  HS_DYNAMIC_LDS(smem);
  if (T.st->done) return;
  if (!out_rec) {
    out_rec = T.st->rec_sel ? T.v_rec : T.v_rec_alt;
    if (blockIdx.x == 0 && threadIdx.x == 0) T.st->rec_pending = 1;
  }
  if (!cp_src) cp_src = T.cp;
  constexpr int REC = 8 + 12 * K, LREC = REC + 2, NCH = REC / 2;  // LDS record stride (16-byte aligned, bank-spread), 16-B chunks
  constexpr int NW = lin_block<K>() / 64;
  // control points: LDS copy when it fits next to the record slabs, otherwise straight from L2 (long windows)
  const bool cps_in_lds = size_t(8) * T.sp.n_cp * sizeof(double) <= 24 * 1024;
  const double* cps = cps_in_lds ? smem : cp_src;
  double* slab = smem + (cps_in_lds ? 8 * T.sp.n_cp : 0) + (threadIdx.x >> 6) * 64 * LREC;  // this wave's 64 records
  const bool lprof = prof_enabled(T.debug_flags, 32) && threadIdx.x == 0 && blockIdx.x < 256;
  long long* llog = reinterpret_cast<long long*>(T.xpart) + 32 * 1024 + 4 * blockIdx.x;
  if (lprof) llog[0] = wall_clock64();
  if (cps_in_lds) stage_cps(cp_src, smem, 8 * T.sp.n_cp);
  if (lprof) llog[1] = wall_clock64();
  __shared__ double red[NW];
  __shared__ int slots[NW * 64];
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0;
  int slot = -1;
  if (q < T.n_vis) {
    VisualOut<K> o;
    visual_linearize<K>(T, cps, q, robustify != 0, &o, lm_src);
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
  HS_DYNAMIC_LDS(smem);
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

/// Residuals per workgroup of the inertial kernels. The value-only kernels run one wave of 64 residuals per workgroup; the linearisation
/// spreads each residual over K lanes — lane (wave m, lane t) forms the Jacobian block of control point m of residual t — so that a window's
/// ~1e4 inertial residual blocks occupy K x 157 waves instead of 157 (one lane per residual held three derivative levels of the spline with
/// all their K Jacobian blocks: 512 VGPRs + scratch, 56 us at configs[2] on 15 % of the SIMDs).
constexpr int kInertialBlock = 64;

/// Inertial residual blocks (inertial.cpp:13-205): record = [r(6) | J_state(6 x 6K) | wg(KB) | wa(KB) | J_gravity(6 x 2)].
/// blockDim = 64 K: wave m handles control point m of the workgroup's 64 residuals (m is wave uniform: no divergence on it).
template <int K, int KB>
__global__ void __launch_bounds__(kInertialBlock * K) k_linearize_inertial(Tables T, double* out_rec, int robustify, double* cost_part, double* cost_each) {
  HS_DYNAMIC_LDS(smem);
  if (T.st->done) return;
  double* cps = smem;
  stage_cps(T.cp, cps, 8 * T.sp.n_cp);
  __shared__ double red[K];
  const int lane = threadIdx.x & 63, m = threadIdx.x >> 6;
  const int i = blockIdx.x * kInertialBlock + lane;
  double cost = 0.0;
  if (i < T.n_ine) {
    constexpr int REC = 18 + 36 * K + 2 * KB;
    cost = inertial_linearize_col<K, KB>(T, cps, T.bias_g, T.bias_a, T.gravity, i, m, robustify != 0, out_rec + size_t(i) * REC);
    if (cost_each && m == 0) cost_each[i] = cost;
  }
  const double s = block_sum(cost, red);  // (only wave 0 carries costs)
  if (threadIdx.x == 0 && cost_part) cost_part[blockIdx.x] = s;
}

template <int K, int KB>
__global__ void __launch_bounds__(kInertialBlock) k_cost_inertial(Tables T, const double* cp_src, const double* bg, const double* ba, const double* grav,
                                                         double* cost_part) {
  HS_DYNAMIC_LDS(smem);
  if (T.st->done) return;
  double* cps = smem;
  stage_cps(cp_src, cps, 8 * T.sp.n_cp);
  __shared__ double red[kBlock / 64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0;
  if (i < T.n_ine) {
    InertialOut<K, KB> o;
    o.Jp = nullptr;  // (value-only branch)
    inertial_evaluate<K, KB, false>(T, cps, bg, ba, grav, i, false, &o);
    cost = o.cost;
  }
  const double s = block_sum(cost, red);
  if (threadIdx.x == 0) cost_part[blockIdx.x] = s;
}

/// Cost at the candidate point (residual-only branch).
template <int K>
__global__ void __launch_bounds__(kBlock) k_cost_visual(Tables T, const double* cp_src, const double* lm_src, double* cost_part) {
  HS_DYNAMIC_LDS(smem);
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
  HS_DYNAMIC_LDS(smem);
  if (T.st->done) return;
  double* cps = smem;
  stage_cps(cp_src, cps, 8 * T.sp.n_cp);
  __shared__ double red[kBlock / 64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double cost = (i < T.n_pri) ? prior_cost<K>(T, cps, i) : 0.0;
  const double s = block_sum(cost, red);
  if (threadIdx.x == 0) cost_part[blockIdx.x] = s;
}

/// Cost at the candidate point of every factor type in ONE launch (windows with inertial or prior factors: three dependent launches of
/// ~8 us each otherwise, almost all of it launch + first-load latency). Workgroups [0, nb_vis) visual, [nb_vis, nb_vis + nb_pri) prior, the
/// rest inertial (kInertialBlock residuals per workgroup, first wave). Same partial sums, same slots of cost_part as the three kernels.
template <int K, int KB>
__global__ void __launch_bounds__(kBlock) k_cost_all(Tables T, const double* cp_src, const double* lm_src, const double* bg, const double* ba, const double* grav,
                                                     double* cost_part, int nb_vis, int nb_pri) {
  HS_DYNAMIC_LDS(smem);
  if (T.st->done) return;
  double* cps = smem;
  stage_cps(cp_src, cps, 8 * T.sp.n_cp);
  __shared__ double red[kBlock / 64];
  const int b = blockIdx.x;
  double cost = 0.0;
  if (b < nb_vis) {
    const int q = b * kBlock + threadIdx.x;
    if (q < T.n_vis) cost = visual_cost<K>(T, cps, lm_src, q);
  } else if (b < nb_vis + nb_pri) {
    const int i = (b - nb_vis) * kBlock + threadIdx.x;
    if (i < T.n_pri) cost = prior_cost<K>(T, cps, i);
  } else {
    const int i = (b - nb_vis - nb_pri) * kInertialBlock + threadIdx.x;
    if (threadIdx.x < kInertialBlock && i < T.n_ine) {
      InertialOut<K, KB> o;
      o.Jp = nullptr;  // (value-only branch)
      inertial_evaluate<K, KB, false>(T, cps, bg, ba, grav, i, false, &o);
      cost = o.cost;
    }
  }
  const double s = block_sum(cost, red);
  if (threadIdx.x == 0) cost_part[b] = s;
}

}  // namespace hs
