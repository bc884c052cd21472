// host_structure.hpp — host-side graph structure of the window (index arithmetic only, bit-exact parity target).
//
// Restates, for whole tables, what the reference does per residual block at add-time:
//   ExteroceptiveCost::update   /root/reference/internal/hyper/optimizers/ceres/costs/exteroceptive.cpp:25-99
//     (block list state || sensor || observation, indices, sizes, exclusive-prefix offsets, num_residuals)
//   segment lookup of the uniform basis (control points [i-(k-1)/2, i-(k-1)/2+k-1], /root/reference/internal/hyper/optimizers/abstract.cpp:89)
// and builds the sort orders the kernels rely on (landmark-major residuals, segment-major records, landmarks ordered by the
// first control point they touch).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/hyperslam_hip.h"
#include "device_math.hpp"

namespace hs {

inline double h_binom(int n, int r) {
  if (r < 0 || r > n) return 0.0;
  double v = 1.0;
  for (int i = 1; i <= r; ++i) v = v * (n - r + i) / i;
  return v;
}

/// Cumulative blending matrix of the uniform B-spline of order k.
inline hsd::BasisCoef make_basis_coef(int k) {
  hsd::BasisCoef b;
  for (double& v : b.c) v = 0.0;
  std::vector<double> M(size_t(k) * k, 0.0);
  double fact = 1.0;
  for (int i = 2; i <= k - 1; ++i) fact *= i;
  for (int s = 0; s < k; ++s)
    for (int n = 0; n < k; ++n) {
      double sum = 0.0;
      for (int l = s; l <= k - 1; ++l) sum += (((l - s) & 1) ? -1.0 : 1.0) * h_binom(k, l - s) * std::pow(double(k - 1 - l), double(k - 1 - n));
      M[size_t(s) * k + n] = h_binom(k - 1, n) / fact * sum;
    }
  for (int j = 0; j < k; ++j)
    for (int n = 0; n < k; ++n) {
      double sum = 0.0;
      for (int s = j; s < k; ++s) sum += M[size_t(s) * k + n];
      b.c[j * hsd::kMaxOrder + n] = sum;
    }
  return b;
}

inline int h_segment_first(double t, double t0, double dt, int k) {
  // same expression as the device (IEEE division) so that stamps on a knot land in the same segment
  const double x = (t - t0) / dt;
  return int(std::floor(x)) - (k - 1) / 2;
}

/// Block structure of one residual (ExteroceptiveCost::update).
struct BlockLayout {
  int num_blocks = 0, num_parameters = 0, num_residuals = 0;
  int indices[4] = {0, 0, 0, 0};
  int sizes[32], offsets[32];
};
inline BlockLayout make_block_layout(int type, int k, int kb) {
  BlockLayout L;
  int n = 0;
  for (int j = 0; j < k; ++j) L.sizes[n++] = 8;
  int n_static = 0, n_sensor = 0;
  if (type == HS_PIXEL || type == HS_BEARING) {
    L.sizes[n++] = 7, L.sizes[n++] = 4, L.sizes[n++] = 4;
    n_static = n_sensor = 3;
    L.sizes[n++] = 3;
    L.num_residuals = type == HS_PIXEL ? 2 : 1;
  } else if (type == HS_PRIOR) {
    L.sizes[n++] = 7;
    n_static = n_sensor = 1;
    L.num_residuals = 6;
  } else {
    L.sizes[n++] = 7, L.sizes[n++] = 6, L.sizes[n++] = 6, L.sizes[n++] = 9, L.sizes[n++] = 9;
    n_static = 5;
    for (int j = 0; j < 2 * kb; ++j) L.sizes[n++] = 4;
    n_sensor = 5 + 2 * kb;
    L.sizes[n++] = 3;
    L.num_residuals = 6;
  }
  L.num_blocks = n;
  L.indices[0] = 0, L.indices[1] = k, L.indices[2] = k + n_static, L.indices[3] = k + n_sensor;
  L.offsets[0] = 0;
  for (int i = 1; i < n; ++i) L.offsets[i] = L.offsets[i - 1] + L.sizes[i - 1];
  for (int i = 0; i < n; ++i) L.num_parameters += L.sizes[i];
  return L;
}

/// Sorted structure of the visual part of the window.
struct VisualStructure {
  // landmark-major residual list (index q)
  std::vector<int> table_type, table_idx;  // origin of residual q (HS_PIXEL / HS_BEARING, index in that table)
  std::vector<int> lm_dev;                 // device landmark id
  std::vector<int> first;                  // first control point
  std::vector<int> pos;                    // record slot (segment-major)
  std::vector<int> seg_ptr;                // n_seg + 1
  // landmarks
  std::vector<int> dev_of_table, table_of_dev;  // permutation (all landmarks; unobserved ones last)
  std::vector<int> lm_ptr, lm_cfirst, lm_ncp, lm_yoff, cf_ptr;
  int bw = 0;
  int y_total = 0;
  std::vector<int> scratch_first, scratch_cf, scratch_cl, scratch_cnt, scratch_start;  // work arrays of build_visual_structure (kept: no allocation per call)
};

struct VisualInput {
  int k, n_cp, n_lm;
  double t0, dt;
  int n_px, n_br;
  const double *px_stamp, *br_stamp;
  const int32_t *px_lm, *br_lm;
};

inline bool build_visual_structure(const VisualInput& in, VisualStructure* vs, std::string* err) {
  // Three passes over the residual tables (this runs on the host in front of every optimize() of a sliding window: ~2 ns per residual
  // and pass): (A) first control point, landmark ranges and the two histograms, (B) stable scatter into landmark-major order,
  // (C) stable scatter of the record slots into segment-major order. All orderings are stable counting sorts on small integer keys.
  const int n = in.n_px + in.n_br, n_seg = in.n_cp - in.k + 1;
  vs->scratch_first.resize(n);
  std::vector<int>& first = vs->scratch_first;
  std::vector<int>& cf = vs->scratch_cf;
  std::vector<int>& cl = vs->scratch_cl;
  std::vector<int>& cnt = vs->scratch_cnt;  // residuals per table landmark, then the write cursor of its run
  cf.assign(in.n_lm, 1 << 30), cl.assign(in.n_lm, -1), cnt.assign(in.n_lm, 0);
  vs->seg_ptr.assign(n_seg + 1, 0);
  auto pass_a = [&](const double* stamp, const int32_t* lm, int count, int offset) {
    for (int j = 0; j < count; ++j) {
      const int l = lm[j], f = h_segment_first(stamp[j], in.t0, in.dt, in.k);
      if (l < 0 || l >= in.n_lm) {
        *err = "visual residual references a landmark outside the landmark table";
        return false;
      }
      if (f < 0 || f >= n_seg) {
        *err = "visual residual stamp outside the valid range of the spline";
        return false;
      }
      first[offset + j] = f;
      cf[l] = std::min(cf[l], f), cl[l] = std::max(cl[l], f + in.k - 1);
      cnt[l]++;
      vs->seg_ptr[f + 1]++;
    }
    return true;
  };
  if (!pass_a(in.px_stamp, in.px_lm, in.n_px, 0) || !pass_a(in.br_stamp, in.br_lm, in.n_br, in.n_px)) return false;
  for (int s = 0; s < n_seg; ++s) vs->seg_ptr[s + 1] += vs->seg_ptr[s];
  // device order of the landmarks: observed ones by first control point (stable), unobserved last
  {
    std::vector<int>& start = vs->scratch_start;
    start.assign(in.n_cp + 2, 0);
    for (int t = 0; t < in.n_lm; ++t) start[std::min(cf[t], in.n_cp) + 1]++;
    for (int c = 0; c <= in.n_cp; ++c) start[c + 1] += start[c];
    vs->table_of_dev.resize(in.n_lm), vs->dev_of_table.resize(in.n_lm);
    for (int t = 0; t < in.n_lm; ++t) {
      const int d = start[std::min(cf[t], in.n_cp)]++;
      vs->table_of_dev[d] = t, vs->dev_of_table[t] = d;
    }
  }
  vs->lm_cfirst.resize(in.n_lm), vs->lm_ncp.resize(in.n_lm), vs->lm_yoff.resize(in.n_lm + 1), vs->lm_ptr.resize(in.n_lm + 1);
  vs->lm_yoff[0] = 0, vs->lm_ptr[0] = 0;
  vs->bw = in.k;
  for (int d = 0; d < in.n_lm; ++d) {
    const int t = vs->table_of_dev[d];
    if (cl[t] >= 0) {
      vs->lm_cfirst[d] = cf[t];
      vs->lm_ncp[d] = cl[t] - cf[t] + 1;
      vs->bw = std::max(vs->bw, vs->lm_ncp[d]);
    } else {
      vs->lm_cfirst[d] = in.n_cp;  // sorts after every control point
      vs->lm_ncp[d] = 0;
    }
    vs->lm_yoff[d + 1] = vs->lm_yoff[d] + 18 * vs->lm_ncp[d];
    vs->lm_ptr[d + 1] = vs->lm_ptr[d] + cnt[t];
    cnt[t] = vs->lm_ptr[d];  // from here on: where the next residual of table landmark t goes
  }
  vs->y_total = vs->lm_yoff[in.n_lm];
  vs->cf_ptr.assign(in.n_cp + 2, 0);
  for (int c = 0, d = 0; c <= in.n_cp + 1; ++c) {
    while (d < in.n_lm && vs->lm_cfirst[d] < c) ++d;
    vs->cf_ptr[c] = d;
  }
  // (B) landmark-major residual order (stable: table order within a landmark, pixel before bearing)
  vs->table_type.resize(n), vs->table_idx.resize(n), vs->lm_dev.resize(n), vs->first.resize(n), vs->pos.resize(n);
  auto pass_b = [&](const int32_t* lm, int count, int offset, int type) {
    for (int j = 0; j < count; ++j) {
      const int t = lm[j], q = cnt[t]++;
      vs->table_type[q] = type, vs->table_idx[q] = j, vs->lm_dev[q] = vs->dev_of_table[t], vs->first[q] = first[offset + j];
    }
  };
  pass_b(in.px_lm, in.n_px, 0, HS_PIXEL), pass_b(in.br_lm, in.n_br, in.n_px, HS_BEARING);
  // (C) segment-major record slots (stable over the landmark-major order)
  {
    std::vector<int>& cur = vs->scratch_start;
    cur.assign(vs->seg_ptr.begin(), vs->seg_ptr.end() - 1);
    for (int q = 0; q < n; ++q) vs->pos[q] = cur[vs->first[q]]++;
  }
  return true;
}

/// Work list of the fused build (kernels_build.hpp). Chunk = consecutive device landmarks of one landmark group (same first control point),
/// at most L landmarks and at most R residuals (one lane each). Returns false when a single landmark has more than R residuals (the caller
/// takes the record path then).
/// ch_ptr[w] .. ch_ptr[w + 1]: landmarks of chunk w;  gw_ptr[c] .. gw_ptr[c + 1]: chunks of group c;  gw_cf[w]: group of chunk w.
inline bool build_chunks(const VisualStructure& vs, int n_cp, int R, int L, std::vector<int>* ch_ptr, std::vector<int>* gw_ptr, std::vector<int>* gw_cf,
                         std::vector<int>* ch_desc = nullptr) {
  int n_obs = int(vs.lm_ptr.size()) - 1;
  while (n_obs > 0 && vs.lm_ptr[n_obs] == vs.lm_ptr[n_obs - 1]) --n_obs;  // unobserved landmarks are last in device order
  ch_ptr->clear(), gw_cf->clear();
  gw_ptr->assign(n_cp + 1, 0);
  for (int c = 0; c < n_cp; ++c) {
    const int d0 = std::min(vs.cf_ptr[c], n_obs), d1 = std::min(vs.cf_ptr[c + 1], n_obs);
    (*gw_ptr)[c] = int(gw_cf->size());
    if (d1 <= d0) continue;
    int d = d0;
    while (d < d1) {  // greedy fill: as many landmarks as the two limits admit (fewest chunks; an even split of a group costs chunks)
      if (vs.lm_ptr[d + 1] - vs.lm_ptr[d] > R) return false;
      ch_ptr->push_back(d), gw_cf->push_back(c);
      int cnt = 0, e = d;
      while (e < d1 && e - d < L && cnt + (vs.lm_ptr[e + 1] - vs.lm_ptr[e]) <= R) cnt += vs.lm_ptr[e + 1] - vs.lm_ptr[e], ++e;
      d = e;
    }
  }
  (*gw_ptr)[n_cp] = int(gw_cf->size());
  ch_ptr->push_back(n_obs);
  gw_cf->push_back(0);
  if (ch_desc) {  // [first landmark, landmarks, first control point, first residual, residuals, chunk id (= slot of its partial), 0, 0] per chunk
    const int n = int(ch_ptr->size()) - 1;
    ch_desc->assign(size_t(8) * std::max(n, 1), 0);
    for (int w = 0; w < n; ++w) {
      const int lo = (*ch_ptr)[w], hi = (*ch_ptr)[w + 1];
      int* d = ch_desc->data() + 8 * w;
      d[0] = lo, d[1] = hi - lo, d[2] = vs.lm_cfirst[lo], d[3] = vs.lm_ptr[lo], d[4] = vs.lm_ptr[hi] - vs.lm_ptr[lo], d[5] = w;
    }
  }
  return true;
}

/// Order in which the chunks are handed to the workgroups of k_build_visual / k_update_visual (descriptor w = the chunk workgroup w works on;
/// a chunk's partial stays in the slot of its id, so k_assemble is not concerned). The kernels keep two workgroups per CU, and the hardware
/// places workgroup w on CU w mod n_cu: with n_cu < n <= 2 n_cu chunks the workgroups n - n_cu .. n_cu - 1 have a CU to themselves (20 us
/// per chunk instead of ~29) and w shares its CU with w + n_cu (measured: tools/build_phase_timing.py). Of the two workgroups of a CU the
/// SECOND one ends ~6 us behind the first whatever its records are (round 6: two chunks take ~35 us of a CU's LDS + fp64 issue, and the
/// older workgroup gets the larger share early), and the slowest chunk is the kernel time: the heaviest chunks get the CUs of their own,
/// the rest are paired heaviest (first) with lightest (second). More than two rounds: heaviest first.
/// Weight of a chunk = most records in k consecutive segments (what the longest lane of its W phase walks; the J'J phase deals its streams
/// per chunk since round 6 and no longer depends on it for bands of up to 64 tiles).
inline void order_chunks_for_dispatch(const VisualStructure& vs, int k, int n_cu, std::vector<int>* ch_desc, int n) {
  if (n <= 1 || n_cu <= 0) return;
  std::vector<int> weight(n), order(n);
  std::vector<int> cnt;
  for (int c = 0; c < n; ++c) {
    const int* d = ch_desc->data() + 8 * c;
    cnt.assign(size_t(vs.bw) + k + 1, 0);
    for (int q = d[3]; q < d[3] + d[4]; ++q) {
      const int o = vs.first[q] - d[2];
      if (o >= 0 && o < vs.bw) cnt[o]++;
    }
    int run = 0, best = 0;
    for (int o = 0; o < vs.bw + k; ++o) {
      run += (o < vs.bw ? cnt[o] : 0) - (o >= k ? cnt[o - k] : 0);
      best = std::max(best, run);
    }
    weight[c] = best, order[c] = c;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weight[a] > weight[b]; });
  std::vector<int> slot_chunk(n, -1);
  if (n > n_cu && n <= 2 * n_cu) {
    const int n_lone = 2 * n_cu - n, n_pair = n - n_cu;
    for (int i = 0; i < n_lone; ++i) slot_chunk[n_pair + i] = order[i];                 // workgroups n - n_cu .. n_cu - 1: alone on their CU
    for (int i = 0; i < n_pair; ++i) slot_chunk[i] = order[n_lone + i], slot_chunk[n_cu + i] = order[n - 1 - i];  // heavy w, light w + n_cu
  } else {
    for (int i = 0; i < n; ++i) slot_chunk[i] = order[i];
  }
  std::vector<int> out(ch_desc->size(), 0);
  for (int w = 0; w < n; ++w) std::copy(ch_desc->begin() + 8 * slot_chunk[w], ch_desc->begin() + 8 * slot_chunk[w] + 8, out.begin() + 8 * w);
  ch_desc->swap(out);
}

/// Chunk geometry of the fused build: records (= lanes) per chunk R and landmarks per chunk L such that TWO workgroups share a CU
/// (lds_two bytes each); bands too wide for that get one workgroup per CU (lds_one); false: no geometry fits (record path).
template <class LdsBytes>
inline bool choose_build_geometry(int k, int R0, int L0, size_t lds_two, size_t lds_one, LdsBytes lds_bytes, int* R, int* L) {
  for (const size_t cap : {lds_two, lds_one}) {
    int r = R0, l = L0;
    while (lds_bytes(r, l) > cap && l > 4) --l;
    while (lds_bytes(r, l) > cap && r > 64) r -= 32;
    if (lds_bytes(r, l) <= cap) {
      *R = r, *L = l;
      return true;
    }
  }
  (void)k;
  return false;
}

}  // namespace hs
