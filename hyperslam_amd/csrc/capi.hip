// capi.hip — C ABI of libhyperslam_hip.so (include/hyperslam_hip.h): host-side table management, structure building
// and the launch sequence of the device-resident Levenberg-Marquardt loop.
//
// Replaces CeresOptimizer::{add(...), updateState, addLandmark, updateLandmarks, optimize}
// (/root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:189-382) behind flat tables. There is no CPU fallback:
// every evaluation entry point runs the gfx950 kernels of kernels.hpp and fails with HS_ERR_DEVICE if no GPU is usable.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "host_structure.hpp"
#include "kernels.hpp"

using namespace hs;

namespace {

/// Table uploads of one prepare() collected into ONE pinned staging arena: a sliding window re-uploads ~45 small tables at every
/// optimize(), and 45 hipMemcpyAsync calls from pageable memory cost more host time than the solve's launches. add() copies the
/// source into the arena and records (destination, offset, bytes); flush() sends the arena with one asynchronous copy and lets one
/// kernel scatter the segments to their destinations. The arena stays alive, so no host synchronisation is needed afterwards.
struct UploadBatch {
  struct Seg {
    unsigned long long dst, off, bytes;
  };
  std::vector<Seg> segs;
  char* host = nullptr;  // pinned
  char* dev = nullptr;
  size_t host_cap = 0, dev_cap = 0, used = 0;
  hipEvent_t sent = nullptr;  // the previous arena content has left the host
  bool in_flight = false;
  ~UploadBatch() {
    if (host) (void)hipHostFree(host);
    if (dev) (void)hipFree(dev);
    if (sent) (void)hipEventDestroy(sent);
  }
  hipError_t grow_host(size_t need) {
    if (need <= host_cap) return hipSuccess;
    const size_t want = std::max<size_t>(std::max<size_t>(need, size_t(1) << 20), 2 * host_cap);
    char* fresh = nullptr;
    const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&fresh), want, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (used) std::memcpy(fresh, host, used);
    if (host) (void)hipHostFree(host);
    host = fresh, host_cap = want;
    return hipSuccess;
  }
  /// Room for `bytes` more without moving the arena: pointers handed out by alloc() stay valid until then.
  hipError_t reserve_more(size_t bytes) { return grow_host(((used + 15) & ~size_t(15)) + bytes + 16); }
  /// add() without the copy: *out points at the segment's place in the arena, the caller fills it before flush().
  hipError_t alloc(void* dst, size_t bytes, void** out) { return add(dst, nullptr, bytes, out); }
  hipError_t add(void* dst, const void* src, size_t bytes, void** out = nullptr) {
    if (in_flight) {  // (only if two prepare() calls follow each other without a synchronising entry point in between)
      const hipError_t e = hipEventSynchronize(sent);
      if (e != hipSuccess) return e;
      in_flight = false;
    }
    const size_t off = (used + 15) & ~size_t(15);
    const hipError_t e = grow_host(off + bytes + 16);
    if (e != hipSuccess) return e;
    if (src) std::memcpy(host + off, src, bytes);
    if (out) *out = host + off;
    // one workgroup of the scatter kernel per 16 KB: a 0.8 MB residual table copied by a single workgroup took 46 us
    constexpr size_t kChunk = 16 * 1024;
    for (size_t o = 0; o < bytes; o += kChunk)
      segs.push_back(Seg{reinterpret_cast<unsigned long long>(dst) + o, off + o, std::min(kChunk, bytes - o)});
    used = off + bytes;
    return hipSuccess;
  }
  hipError_t flush(hipStream_t s);
};
thread_local UploadBatch* tl_upload_batch = nullptr;  // set by prepare() around its uploads

template <class T>
struct DBuf {
  T* p = nullptr;
  size_t cap = 0;
  ~DBuf() {
    if (p) (void)hipFree(p);
  }
  /// Capacity grows geometrically: a sliding window changes every table size by a little at every optimize(), and an exact-fit
  /// hipFree + hipMalloc per table and solve costs more than the solve itself (hipFree synchronises the device).
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    const size_t want = std::max<size_t>(std::max<size_t>(n, 256), 2 * cap);
    cap = 0;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  hipError_t upload(const std::vector<T>& h, hipStream_t s) {
    hipError_t e = reserve(h.size());
    if (e != hipSuccess || h.empty()) return e;
    if (tl_upload_batch) return tl_upload_batch->add(p, h.data(), h.size() * sizeof(T));
    return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
  }
};

__global__ void __launch_bounds__(256) k_scatter_uploads(const char* arena, const UploadBatch::Seg* segs) {
  const UploadBatch::Seg sg = segs[blockIdx.x];
  const char* src = arena + sg.off;
  char* dst = reinterpret_cast<char*>(sg.dst);
  const size_t n16 = sg.bytes / 16;  // destinations are hipMalloc'ed (256-byte aligned), arena offsets 16-byte aligned
  for (size_t i = threadIdx.x; i < n16; i += 256) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  for (size_t i = 16 * n16 + threadIdx.x; i < sg.bytes; i += 256) dst[i] = src[i];
}

hipError_t UploadBatch::flush(hipStream_t s) {
  if (segs.empty()) return hipSuccess;
  const size_t table = (used + 15) & ~size_t(15), total = table + segs.size() * sizeof(Seg);
  hipError_t e = grow_host(total);
  if (e != hipSuccess) return e;
  std::memcpy(host + table, segs.data(), segs.size() * sizeof(Seg));
  if (total > dev_cap) {
    if (dev) (void)hipFree(dev);
    dev = nullptr, dev_cap = 0;
    const size_t want = std::max<size_t>(2 * total, size_t(1) << 20);
    e = hipMalloc(reinterpret_cast<void**>(&dev), want);
    if (e != hipSuccess) return e;
    dev_cap = want;
  }
  e = hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  if (!sent) {
    e = hipEventCreateWithFlags(&sent, hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  e = hipEventRecord(sent, s);
  if (e != hipSuccess) return e;
  in_flight = true;
  k_scatter_uploads<<<int(segs.size()), 256, 0, s>>>(dev, reinterpret_cast<const Seg*>(dev + table));
  segs.clear(), used = 0;
  return hipGetLastError();
}

}  // namespace

struct hs_problem {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  bool dirty = true;  // tables changed since the last prepare()

  // host tables (caller's table order)
  int k = 0, n_cp = 0;
  double t0 = 0, dt = 0;
  std::vector<double> cp;
  std::vector<uint8_t> cp_const;
  int rot_const = 0, trans_const = 0;
  int n_cam = 0;
  std::vector<double> cam;  // n x 16
  int n_sensor = 0;
  std::vector<double> sensor;  // n x 8
  int n_lm = 0;
  std::vector<double> lm;
  std::vector<uint8_t> lm_const;
  std::vector<double> px_stamp, px_meas, br_stamp, br_meas, pr_stamp, pr_meas, in_stamp, in_meas;
  std::vector<int32_t> px_lm, px_cam, br_lm, br_cam, pr_sensor;
  bool has_imu = false;
  double imu_T_bs[7], imu_i_g[6], imu_i_a[6], imu_S_g[9], imu_X_a[9];
  int kb = 4, n_bias = 0;
  double bias_t0 = 0, bias_dt = 1;
  std::vector<double> bias_g, bias_a;
  int bias_const = 0;
  double gravity[3] = {0, 0, -9.80665};
  int gravity_const = 1;
  int inertial_mode = HS_INERTIAL_AS_REFERENCE;  // hs_set_inertial_jacobian
  hs_problem* scratch = nullptr;                 // one-residual handle of hs_cost_function_evaluate (created on first use)
  int frozen_prefix = 0;                         // leading constant control points: decoupled block rows of the reduced system
  bool stage_timing = false;                     // hs_set_stage_timing
  std::vector<double> weights[4];                // hs_set_weights: CostConfiguration::weights per factor type (empty: none)
  bool has_weights() const { return !weights[0].empty() || !weights[1].empty() || !weights[2].empty() || !weights[3].empty(); }
  UploadBatch batch;                             // table uploads of prepare()
  bool host_timing = false;                      // HS_HOST_TIMING=1: host wall-clock split of prepare() / hs_solve, printed by hs_destroy
  double host_ms[5] = {0, 0, 0, 0, 0};           // structure + table assembly, uploads, launches, wait for the device, (spare)
  int host_calls = 0;
  std::vector<double> host_log;                  // the same four numbers per call
  double host_prepare[2] = {0, 0};               // (prepare() of the running call)
  // Result read-back: a caller that fetches the state after every solve (the sliding-window driver: control points, landmarks, bias
  // points, gravity = four synchronous copies of ~30 us each) gets it copied into pinned host memory at the end of hs_solve, in the stream,
  // before the solve's own synchronisation; the getters then read host memory. Enabled by the first getter call that had to go to the device.
  double* h_result = nullptr;
  size_t h_result_cap = 0;
  bool want_results = false, results_cached = false;
  int zeroed_np = -1, zeroed_ncb = -1;           // layout / allocations for which the never-written parts of Sb2, Vb, yt were zeroed
  const void* zeroed_ptr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};

  // structure
  VisualStructure vs;
  std::vector<int> pr_order;  // segment-major order of prior residuals (device index -> table index)
  std::vector<int> pr_first, pr_seg_ptr;

  // device
  DBuf<double> d_cp, d_cp_cand, d_cam, d_sensor, d_lm, d_lm_cand;
  DBuf<uint8_t> d_cp_const, d_lm_const;
  DBuf<int> d_lm_ptr, d_lm_cfirst, d_lm_ncp, d_lm_yoff, d_cf_ptr;
  DBuf<double> d_lm_scale, d_lm_L, d_lm_yhat, d_lm_sb, d_lm_D2, d_lm_part, d_lm_gmax, d_gabs, d_Y;
  DBuf<double> d_v_stamp, d_v_meas, d_v_rec, d_v_rec_alt;
  DBuf<int> d_v_lm, d_v_info, d_v_first, d_v_pos, d_v_seg_ptr, d_v_dbgpos;
  DBuf<double> d_p_stamp, d_p_meas, d_p_rec;
  DBuf<double> d_i_stamp, d_i_meas, d_i_rec, d_bias_g, d_bias_a, d_bias_g_cand, d_bias_a_cand, d_gravity, d_gravity_cand;
  DBuf<int> d_i_first, d_i_first_bias, d_i_seg_ptr;
  DBuf<ImuParams> d_imu;
  std::vector<int> in_order, in_first, in_first_bias, in_seg_ptr, in_bias_ptr;
  int nb_ine = 0;
  DBuf<int> d_p_sensor, d_p_first, d_p_seg_ptr;
  DBuf<double> d_scale_p, d_Sb, d_Ub, d_Ubk, d_g_s, d_g_full, d_D2p, d_step_p, d_delta_p;
  DBuf<double> d_cost_part, d_cand_part, d_norm_part, d_dbg, d_dbg_cost;
  DBuf<DevState> d_state;
  DBuf<double> d_xbuf, d_xpart, d_segP, d_grpQ, d_gravity_part;
  DBuf<double> d_Vb, d_Vb2, d_yt, d_yt2;  // block-row-scaled factors diag(U_jj^-1) U and right-hand sides for the register sweep
  DBuf<double> d_Sb2, d_g2, d_Ub2, d_Ubk2, d_ybuf2, d_win, d_xsol;  // two-ended factorisation: reversed system, its factor, junction window
  DBuf<unsigned> d_join;
  DBuf<double> d_bf_handover;  // k_border_forward2: what the far end's sweep leaves on the middle rows, per column group
  unsigned join_epoch = 0;
  DBuf<int> d_gw_ptr, d_gw_cf, d_sw_ptr, d_sw_seg;
  int n_seg_wg = 0, n_group_wg = 0;
  // fused build of the visual factors (kernels_build.hpp)
  bool fused = false;
  int build_R = 0, build_L = 0;     // records per pass, landmarks per chunk
  size_t build_lds = 0;
  DBuf<int> d_ch_ptr, d_ch_desc;
  std::vector<int> h_ch_ptr, h_gw_ptr, h_gw_cf, h_ch_desc;
  DBuf<double> d_ybuf, d_scale_b, d_Spb, d_Sbb, d_gb_s, d_D2b, d_Zb, d_Cb, d_hb, d_xb, d_delta_b, d_bias_g_snap, d_bias_a_snap, d_gravity_snap;
  DBuf<int> d_i_bias_ptr, d_bfwd_start;
  int n_split = 1;
  int rank = 0, world = 1, min_bw = 0;
  DBuf<double> d_cp_snap, d_lm_snap;
  bool has_snapshot = false;
  DevState* h_state = nullptr;  // pinned
  std::vector<hipEvent_t> events;
  hipStream_t side = nullptr;           // second stream: the segment partials run next to the landmark pass (independent inputs)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_irec = nullptr;
  bool side_imu = false;  // this iteration's inertial linearisation + border gathers run on the side stream
  Tables T;
  int nb_vis = 0, nb_pri = 0, nb_cp = 0;
  int chol_lds_max = 64 * 1024;
  hs_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  void* rccl_comm = nullptr;  // ncclComm_t when hs_rccl_init was called
};

static const char* kWeightsMessage =
    "a weight matrix is set (hs_set_weights): weights are applied by hs_linearize / hs_cost_function_evaluate; the solver runs the "
    "reference's production configuration, weights = nullptr (optimizer.cpp:191,214,236,255)";

/// Measurement switches that select a kernel kept for A/B comparison only: compile-time false in the product library (HS_PROFILE_HOOKS = 0,
/// the alternatives are not compiled in), HS_DEBUG_FLAGS bits in profiling builds (tools/build_profiling_lib.sh).
#define HS_AB(flags, bit) (HS_PROFILE_HOOKS && ((flags) & (bit)))

#define HS_FAIL(code, msg) \
  do {                     \
    p->err = (msg);        \
    return (code);         \
  } while (0)
#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    const hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                      \
      p->err = std::string(#expr) + ": " + hipGetErrorString(e__);                                \
      return HS_ERR_DEVICE;                                                                       \
    }                                                                                             \
  } while (0)

namespace {

int mfma_window_tiles(int bw);

int prepare(hs_problem* p) {
  if (!p->dirty) return HS_OK;
  struct BatchScope {  // every DBuf::upload below goes through the staging arena; sent in one piece at the end
    UploadBatch* b;
    explicit BatchScope(UploadBatch* x) : b(x) { tl_upload_batch = b; }
    ~BatchScope() {
      tl_upload_batch = nullptr;
      b->segs.clear(), b->used = 0;  // (no-op after a flush; drops the pending segments of a failed prepare)
    }
  } batch_scope(&p->batch);
  const auto host_t0 = std::chrono::steady_clock::now();
  if (p->n_cp == 0) HS_FAIL(HS_ERR_STATE, "hs_set_spline has not been called");
  if (p->k != 4 && p->k != 6) HS_FAIL(HS_ERR_INVALID, "device kernels are instantiated for spline order 4 and 6");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  const int k = p->k, n_seg = p->n_cp - k + 1;
  const int n_px = int(p->px_stamp.size()), n_br = int(p->br_stamp.size());
  for (int i = 0; i < n_px; ++i)
    if (p->px_cam[i] < 0 || p->px_cam[i] >= p->n_cam) HS_FAIL(HS_ERR_INVALID, "pixel residual references a camera outside the camera table");
  for (int i = 0; i < n_br; ++i)
    if (p->br_cam[i] < 0 || p->br_cam[i] >= p->n_cam) HS_FAIL(HS_ERR_INVALID, "bearing residual references a camera outside the camera table");
  VisualInput in = {k, p->n_cp, p->n_lm, p->t0, p->dt, n_px, n_br, p->px_stamp.data(), p->br_stamp.data(), p->px_lm.data(), p->br_lm.data()};
  if (!build_visual_structure(in, &p->vs, &p->err)) return HS_ERR_INVALID;
  p->vs.bw = std::max(p->vs.bw, p->min_bw);
  const VisualStructure& vs = p->vs;
  if (6 * vs.bw > kBlock || (size_t(42) * (6 * vs.bw + 2) + size_t(6) * p->n_cp + 48) * 8 > size_t(p->chol_lds_max))
    HS_FAIL(HS_ERR_INVALID, "landmark tracks span too many control points for the LDS-resident banded factorisation");
  if (p->n_cp > 1024)  // 64 KiB of control points staged per workgroup; 96 KiB right-hand side + 49 KiB junction block in the backward sweep
    HS_FAIL(HS_ERR_INVALID, "window too long for the LDS-resident control-point table and backward sweep (more than 1024 control points)");
  const int n_vis = n_px + n_br;

  // ---- visual tables (landmark-major): written straight into the staging arena in the upload phase below ----
  // landmarks in device order
  std::vector<double> lm_dev(size_t(3) * p->n_lm);
  std::vector<uint8_t> lmc_dev(p->n_lm);
  for (int d = 0; d < p->n_lm; ++d) {
    const int t = vs.table_of_dev[d];
    for (int c = 0; c < 3; ++c) lm_dev[3 * d + c] = p->lm[3 * t + c];
    lmc_dev[d] = p->lm_const[t];
  }
  // ---- prior tables (segment-major) ----
  const int n_pri = int(p->pr_stamp.size());
  p->pr_order.resize(n_pri);
  std::vector<int> first_tab(n_pri);
  for (int i = 0; i < n_pri; ++i) {
    first_tab[i] = h_segment_first(p->pr_stamp[i], p->t0, p->dt, k);
    if (first_tab[i] < 0 || first_tab[i] >= n_seg) HS_FAIL(HS_ERR_INVALID, "prior residual stamp outside the valid range of the spline");
    if (p->pr_sensor[i] < 0 || p->pr_sensor[i] >= p->n_sensor) HS_FAIL(HS_ERR_INVALID, "prior residual references a sensor outside the sensor table");
    p->pr_order[i] = i;
  }
  std::stable_sort(p->pr_order.begin(), p->pr_order.end(), [&](int a, int b) { return first_tab[a] < first_tab[b]; });
  std::vector<double> p_stamp(n_pri), p_meas(size_t(7) * n_pri);
  std::vector<int> p_sensor(n_pri);
  p->pr_first.resize(n_pri);
  p->pr_seg_ptr.assign(n_seg + 1, 0);
  for (int d = 0; d < n_pri; ++d) {
    const int t = p->pr_order[d];
    p_stamp[d] = p->pr_stamp[t], p_sensor[d] = p->pr_sensor[t], p->pr_first[d] = first_tab[t];
    for (int c = 0; c < 7; ++c) p_meas[7 * d + c] = p->pr_meas[7 * t + c];
    p->pr_seg_ptr[first_tab[t] + 1]++;
  }
  for (int sgm = 0; sgm < n_seg; ++sgm) p->pr_seg_ptr[sgm + 1] += p->pr_seg_ptr[sgm];
  // ---- inertial tables (segment-major) ----
  const int n_ine = int(p->in_stamp.size());
  if (n_ine && !p->has_imu) HS_FAIL(HS_ERR_STATE, "inertial residuals need hs_set_imu");
  if (n_ine && p->kb != 4) HS_FAIL(HS_ERR_INVALID, "device kernels are instantiated for bias-spline order 4");
  std::vector<double> i_stamp(n_ine), i_meas(size_t(6) * n_ine);
  {
    std::vector<int> ft(n_ine), fbt(n_ine);
    p->in_order.resize(n_ine);
    for (int i = 0; i < n_ine; ++i) {
      ft[i] = h_segment_first(p->in_stamp[i], p->t0, p->dt, k);
      fbt[i] = h_segment_first(p->in_stamp[i], p->bias_t0, p->bias_dt, p->kb);
      if (ft[i] < 0 || ft[i] >= n_seg) HS_FAIL(HS_ERR_INVALID, "inertial residual stamp outside the valid range of the spline");
      if (fbt[i] < 0 || fbt[i] + p->kb > p->n_bias) HS_FAIL(HS_ERR_INVALID, "inertial residual stamp outside the valid range of the bias splines");
      p->in_order[i] = i;
    }
    // segment-major, and bias-segment-major inside a segment: both first indices are monotone in time, so first_bias is
    // non-decreasing over the whole table whatever the order of the caller's stamps (k_border_bb's i_bias_ptr ranges rely on it)
    std::stable_sort(p->in_order.begin(), p->in_order.end(), [&](int a, int b) { return ft[a] != ft[b] ? ft[a] < ft[b] : fbt[a] < fbt[b]; });
    p->in_first.resize(n_ine), p->in_first_bias.resize(n_ine);
    p->in_seg_ptr.assign(n_seg + 1, 0);
    for (int d = 0; d < n_ine; ++d) {
      const int t = p->in_order[d];
      i_stamp[d] = p->in_stamp[t], p->in_first[d] = ft[t], p->in_first_bias[d] = fbt[t];
      for (int c = 0; c < 6; ++c) i_meas[6 * d + c] = p->in_meas[6 * t + c];
      p->in_seg_ptr[ft[t] + 1]++;
    }
    for (int sgm = 0; sgm < n_seg; ++sgm) p->in_seg_ptr[sgm + 1] += p->in_seg_ptr[sgm];
    // first_bias is non-decreasing (see the sort above): i_bias_ptr[f] = first record with first_bias >= f
    const int nbias = p->has_imu ? p->n_bias : 0;
    p->in_bias_ptr.assign(nbias + 2, n_ine);
    for (int f = 0, d = 0; f <= nbias + 1; ++f) {
      while (d < n_ine && p->in_first_bias[d] < f) ++d;
      p->in_bias_ptr[f] = d;
    }
  }

  // ---- upload ----
  const auto host_t1 = std::chrono::steady_clock::now();
  HIP_TRY(p->d_cp.upload(p->cp, s));
  HIP_TRY(p->d_cp_cand.reserve(p->cp.size()));
  HIP_TRY(p->d_cp_const.upload(p->cp_const, s));
  p->frozen_prefix = 0;
  while (p->frozen_prefix < p->n_cp && p->cp_const[p->frozen_prefix]) ++p->frozen_prefix;
  HIP_TRY(p->d_cam.upload(p->cam, s));
  HIP_TRY(p->d_sensor.upload(p->sensor, s));
  HIP_TRY(p->d_lm.upload(lm_dev, s));
  HIP_TRY(p->d_lm_cand.reserve(lm_dev.size()));
  HIP_TRY(p->d_lm_const.upload(lmc_dev, s));
  HIP_TRY(p->d_lm_ptr.upload(vs.lm_ptr, s));
  HIP_TRY(p->d_lm_cfirst.upload(vs.lm_cfirst, s));
  HIP_TRY(p->d_lm_ncp.upload(vs.lm_ncp, s));
  HIP_TRY(p->d_lm_yoff.upload(vs.lm_yoff, s));
  HIP_TRY(p->d_cf_ptr.upload(vs.cf_ptr, s));
  const size_t nl = size_t(std::max(p->n_lm, 1));
  {
    std::vector<double> ones(3 * nl, 1.0);  // unobserved landmarks keep scale 1 (never visited by the landmark pass)
    HIP_TRY(p->d_lm_scale.upload(ones, s));  // (copied into the staging arena right here: the vector may go)
  }
  HIP_TRY(p->d_lm_L.reserve(6 * nl));
  HIP_TRY(p->d_lm_yhat.reserve(3 * nl));
  HIP_TRY(p->d_lm_sb.reserve(3 * nl));
  HIP_TRY(p->d_lm_D2.reserve(3 * nl));
  HIP_TRY(p->d_lm_part.reserve(4 * (nl + size_t(n_vis) / kBlock + 2) + 4));  // (one entry per four landmarks; fused path: per chunk, <= landmarks, padded to the grid)
  HIP_TRY(p->d_lm_gmax.reserve(nl));
  HIP_TRY(p->d_Y.reserve(size_t(vs.y_total) + 1));
  if (n_vis) {
    // [stamp | measurement (3: a pixel leaves the third entry zero) | camera | type << 16 | position in the caller's tables] per residual, gathered in
    // landmark-major order right where the staging copy will pick them up (as vectors first they cost an allocation, a zero fill and a copy of 1 MB per call)
    HIP_TRY(p->d_v_stamp.reserve(n_vis));
    HIP_TRY(p->d_v_meas.reserve(size_t(3) * n_vis));
    HIP_TRY(p->d_v_info.reserve(n_vis));
    HIP_TRY(p->d_v_dbgpos.reserve(n_vis));
    HIP_TRY(p->batch.reserve_more(size_t(n_vis) * (8 + 24 + 4 + 4) + 4 * 32));
    void *a0, *a1, *a2, *a3;
    HIP_TRY(p->batch.alloc(p->d_v_stamp.p, size_t(n_vis) * 8, &a0));
    HIP_TRY(p->batch.alloc(p->d_v_meas.p, size_t(n_vis) * 24, &a1));
    HIP_TRY(p->batch.alloc(p->d_v_info.p, size_t(n_vis) * 4, &a2));
    HIP_TRY(p->batch.alloc(p->d_v_dbgpos.p, size_t(n_vis) * 4, &a3));
    double *v_stamp = static_cast<double*>(a0), *v_meas = static_cast<double*>(a1);
    int *v_info = static_cast<int*>(a2), *v_dbgpos = static_cast<int*>(a3);
    for (int q = 0; q < n_vis; ++q) {
      const int ti = vs.table_idx[q];
      if (vs.table_type[q] == HS_PIXEL) {
        v_stamp[q] = p->px_stamp[ti];
        v_meas[3 * q] = p->px_meas[2 * ti], v_meas[3 * q + 1] = p->px_meas[2 * ti + 1], v_meas[3 * q + 2] = 0.0;
        v_info[q] = p->px_cam[ti];
        v_dbgpos[q] = ti;
      } else {
        v_stamp[q] = p->br_stamp[ti];
        for (int c = 0; c < 3; ++c) v_meas[3 * q + c] = p->br_meas[3 * ti + c];
        v_info[q] = p->br_cam[ti] | (1 << 16);
        v_dbgpos[q] = n_px + ti;
      }
    }
  }
  HIP_TRY(p->d_v_lm.upload(vs.lm_dev, s));
  HIP_TRY(p->d_v_first.upload(vs.first, s));
  HIP_TRY(p->d_v_pos.upload(vs.pos, s));
  HIP_TRY(p->d_v_seg_ptr.upload(vs.seg_ptr, s));
  // ---- fused build (kernels_build.hpp) or the record path (long feature tracks: more than 256 window tiles; A/B switch 2147483648... see below) ----
  {
    const int ntile_ = vs.bw * (vs.bw + 1) / 2, nband_ = k * vs.bw - k * (k - 1) / 2;
    const char* env = std::getenv("HS_BUILD_PATH");  // "records": the record path everywhere (measurement switch, like HS_DEBUG_FLAGS)
    p->fused = n_vis > 0 && ntile_ <= kBlock && nband_ <= kBlock && !(env && std::strcmp(env, "records") == 0);
  }
  if (p->fused) {
    // chunk geometry: R residuals (lanes) and L landmarks per chunk, sized for two workgroups per CU (every phase of the kernel is an LDS
    // gather: latency bound on a lone wave per SIMD). HS_BUILD_R / HS_BUILD_L: tuning overrides.
    int R0 = k == 4 ? 128 : 96, L0 = k == 4 ? 12 : 10;
    if (const char* e = std::getenv("HS_BUILD_R")) R0 = std::max(32, std::min(kBlock, std::atoi(e)));
    if (const char* e = std::getenv("HS_BUILD_L")) L0 = std::max(1, std::min(24, std::atoi(e)));  // (<= 24: 9 L + 8 lanes of phase 2a, L lanes of one wave in 4a)
    auto lds_bytes = [&](int r, int l) { return size_t(build_lds_layout(k, vs.bw, r, l).total_doubles) * 8; };
    const bool overridden = std::getenv("HS_BUILD_R") || std::getenv("HS_BUILD_L");  // (a tuning run asks for exactly this geometry, one workgroup per CU if need be)
    p->fused = choose_build_geometry(k, R0, L0, size_t(overridden ? 156 : 79) * 1024, size_t(156) * 1024, lds_bytes, &p->build_R, &p->build_L) &&
               build_chunks(vs, p->n_cp, p->build_R, p->build_L, &p->h_ch_ptr, &p->h_gw_ptr, &p->h_gw_cf, &p->h_ch_desc);
    p->build_lds = p->fused ? lds_bytes(p->build_R, p->build_L) : 0;
  }
  if (!p->fused) {
    HIP_TRY(p->d_v_rec.reserve(size_t(n_vis) * (8 + 12 * k) + 1));
    HIP_TRY(p->d_v_rec_alt.reserve(size_t(n_vis) * (8 + 12 * k) + 1));
  }
  HIP_TRY(p->d_p_stamp.upload(p_stamp, s));
  HIP_TRY(p->d_p_meas.upload(p_meas, s));
  HIP_TRY(p->d_p_sensor.upload(p_sensor, s));
  HIP_TRY(p->d_p_first.upload(p->pr_first, s));
  HIP_TRY(p->d_p_seg_ptr.upload(p->pr_seg_ptr, s));
  HIP_TRY(p->d_p_rec.reserve(size_t(n_pri) * (6 + 36 * k) + 1));
  HIP_TRY(p->d_i_stamp.upload(i_stamp, s));
  HIP_TRY(p->d_i_meas.upload(i_meas, s));
  HIP_TRY(p->d_i_first.upload(p->in_first, s));
  HIP_TRY(p->d_i_first_bias.upload(p->in_first_bias, s));
  HIP_TRY(p->d_i_seg_ptr.upload(p->in_seg_ptr, s));
  HIP_TRY(p->d_i_bias_ptr.upload(p->in_bias_ptr, s));
  {  // k_border_forward: column b of S_pb is zero above the first pose block row its bias point (or gravity) meets a residual in
    const int nbias = p->has_imu ? p->n_bias : 0, nbd_ = nbias ? 6 * nbias + 2 : 0;
    const int n_wg = (nbd_ + kBorderCols - 1) / kBorderCols;
    std::vector<int> start(std::max(n_wg, 1), 0);
    for (int w = 0; w < n_wg; ++w) {
      int first = p->n_cp;
      for (int c = w * kBorderCols; c < std::min(nbd_, (w + 1) * kBorderCols); ++c) {
        int rec = 0;  // gravity columns: the first inertial record
        if (c < 6 * nbias) {
          const int beta = (c < 3 * nbias ? c : c - 3 * nbias) / 3;
          rec = p->in_bias_ptr[std::max(beta - p->kb + 1, 0)];  // first record whose bias segment reaches bias point beta
        }
        if (rec < n_ine) first = std::min(first, p->in_first[rec]);  // (records are segment-major: the earliest control point)
      }
      start[w] = first;
    }
    HIP_TRY(p->d_bfwd_start.upload(start, s));
  }
  HIP_TRY(p->d_i_rec.reserve(size_t(n_ine) * (18 + 36 * k + 2 * p->kb) + 1));
  {
    std::vector<ImuParams> ip(1);
    std::memcpy(ip[0].T_bs, p->imu_T_bs, 56), std::memcpy(ip[0].i_g, p->imu_i_g, 48), std::memcpy(ip[0].i_a, p->imu_i_a, 48);
    std::memcpy(ip[0].S_g, p->imu_S_g, 72), std::memcpy(ip[0].X_a, p->imu_X_a, 72);
    HIP_TRY(p->d_imu.upload(ip, s));
    std::vector<double> grav(p->gravity, p->gravity + 3);
    HIP_TRY(p->d_gravity.upload(grav, s));
  }
  HIP_TRY(p->d_gravity_cand.reserve(3));
  HIP_TRY(p->d_bias_g.upload(p->bias_g, s));
  HIP_TRY(p->d_bias_a.upload(p->bias_a, s));
  HIP_TRY(p->d_bias_g_cand.reserve(p->bias_g.size() + 1));
  HIP_TRY(p->d_bias_a_cand.reserve(p->bias_a.size() + 1));
  p->nb_ine = (n_ine + kInertialBlock - 1) / kInertialBlock;
  const int np = 6 * p->n_cp, ncb = 6 * vs.bw;
  HIP_TRY(p->d_scale_p.reserve(np));
  HIP_TRY(p->d_Sb.reserve(size_t(np) * ncb));
  HIP_TRY(p->d_Ub.reserve(size_t(np) * ncb));
  HIP_TRY(p->d_Ubk.reserve(size_t(p->n_cp) * 24));
  HIP_TRY(p->d_g_s.reserve(np));
  HIP_TRY(p->d_g_full.reserve(np));
  HIP_TRY(p->d_D2p.reserve(np));
  HIP_TRY(p->d_gabs.reserve(np + (p->has_imu ? 6 * p->n_bias + 2 : 0) + 1));
  HIP_TRY(p->d_step_p.reserve(np));
  HIP_TRY(p->d_delta_p.reserve(np));
  const int vis_block = k == 4 ? lin_block<4>() : lin_block<6>();
  p->nb_vis = (n_vis + vis_block - 1) / vis_block;
  if (p->fused) {
    p->nb_vis = std::max((n_vis + kBlock - 1) / kBlock, int(p->h_ch_ptr.size()) - 1);  // one cost partial per chunk / per workgroup of k_cost_visual
    p->h_ch_desc.resize(size_t(8) * p->nb_vis, 0);                                        // (k_build_visual reads its descriptor before it knows whether it is a padding workgroup)
  }
  p->nb_pri = (n_pri + kBlock - 1) / kBlock;
  p->nb_cp = std::max((p->n_cp + kBlock - 1) / kBlock, 1);
  HIP_TRY(p->d_cost_part.reserve(p->nb_vis + p->nb_pri + (n_ine + kBlock - 1) / kBlock + 1));
  HIP_TRY(p->d_cand_part.reserve(p->nb_vis + p->nb_pri + (n_ine + kBlock - 1) / kBlock + 1));
  const int nb_norm = p->nb_cp;
  HIP_TRY(p->d_norm_part.reserve(2 * size_t(nb_norm)));
  const int nbd = p->has_imu ? 6 * p->n_bias + 2 : 0;
  if (nbd && size_t(np) * 8 * 8 > 150 * 1024) HS_FAIL(HS_ERR_INVALID, "window too long for the LDS-resident border forward sweep");
  const int x_count1 = np * (ncb + 3) + np * nbd + nbd * nbd + nbd + 1 + p->world;
  HIP_TRY(p->d_ybuf.reserve(np));
  HIP_TRY(p->d_scale_b.reserve(nbd + 1));
  HIP_TRY(p->d_Spb.reserve(size_t(np) * nbd + 1));
  HIP_TRY(p->d_Sbb.reserve(size_t(nbd) * nbd + 1));
  HIP_TRY(p->d_gb_s.reserve(nbd + 1));
  HIP_TRY(p->d_D2b.reserve(nbd + 1));
  HIP_TRY(p->d_Zb.reserve(size_t(np) * nbd + 1));
  HIP_TRY(p->d_Cb.reserve(size_t(nbd) * nbd + 1));
  HIP_TRY(p->d_hb.reserve(nbd + 1));
  HIP_TRY(p->d_xb.reserve(nbd + 1));
  HIP_TRY(p->d_delta_b.reserve(nbd + 1));
  HIP_TRY(p->d_xbuf.reserve(size_t(x_count1) + 8));
  HIP_TRY(p->d_gravity_part.reserve(size_t(5) * std::max(p->n_bias, 1)));
  {
    // Entries of the reversed copy past the end of the matrix are never written, the pads behind Vb / yt (operands of rows that do not
    // exist, k_band_backward_w) neither: they are zeroed once per allocation and layout — a sliding window keeps both from one
    // optimize() to the next, and five memsets per prepare() cost more host time than the structure tables.
    const size_t nv = size_t(np) * ncb, pad = 64;
    HIP_TRY(p->d_Sb2.reserve(nv));
    for (DBuf<double>* b : {&p->d_Vb, &p->d_Vb2}) HIP_TRY(b->reserve(nv + pad));
    for (DBuf<double>* b : {&p->d_yt, &p->d_yt2}) HIP_TRY(b->reserve(size_t(np) + pad));
    const void* now[5] = {p->d_Sb2.p, p->d_Vb.p, p->d_Vb2.p, p->d_yt.p, p->d_yt2.p};
    if (p->zeroed_np != np || p->zeroed_ncb != ncb || std::memcmp(now, p->zeroed_ptr, sizeof(now)) != 0) {
      HIP_TRY(hipMemsetAsync(p->d_Sb2.p, 0, p->d_Sb2.cap * sizeof(double), s));
      for (DBuf<double>* b : {&p->d_Vb, &p->d_Vb2}) HIP_TRY(hipMemsetAsync(b->p + nv, 0, pad * sizeof(double), s));
      for (DBuf<double>* b : {&p->d_yt, &p->d_yt2}) HIP_TRY(hipMemsetAsync(b->p + np, 0, pad * sizeof(double), s));
      p->zeroed_np = np, p->zeroed_ncb = ncb, std::memcpy(p->zeroed_ptr, now, sizeof(now));
    }
  }
  HIP_TRY(p->d_g2.reserve(np));
  HIP_TRY(p->d_Ub2.reserve(size_t(np) * ncb));
  HIP_TRY(p->d_Ubk2.reserve(size_t(p->n_cp) * 24));
  HIP_TRY(p->d_ybuf2.reserve(np));
  HIP_TRY(p->d_xsol.reserve(np));
  HIP_TRY(p->d_win.reserve(size_t(6) * vs.bw * (ncb + 1)));
  if (!p->d_join.p) {
    // [0] two-ended factor / sweep hand-over, [1] last-block ticket of the backward sweeps, [2] of k_border_bb, [4 ..] super-block inverses
    HIP_TRY(p->d_join.reserve(kBfFlagBase + 512));  // (+ one flag per column group of k_border_forward2)
    HIP_TRY(hipMemsetAsync(p->d_join.p, 0, (kBfFlagBase + 512) * sizeof(unsigned), s));
    p->join_epoch = 0;
  }
  // split the accumulation over enough workgroups to fill the chip (256 CUs x a few workgroups)
  p->n_split = std::max(1, std::min(16, 2048 / std::max(p->n_cp, 1)));
  HIP_TRY(p->d_xpart.reserve(size_t(x_count1) * p->n_split));
  {  // owner-computes reduced system: segment partials and landmark-group partials
    const int n_seg_ = p->n_cp - k + 1, nca = 6 * k, ntile = vs.bw * (vs.bw + 1) / 2;
    // k_seg_gram work list: ~96 visual-record equivalents per workgroup (one LDS stage)
    std::vector<int> sw_ptr(n_seg_ + 1, 0), sw_seg;
    for (int f = 0; f < n_seg_; ++f) {
      const int load = (p->fused ? 0 : vs.seg_ptr[f + 1] - vs.seg_ptr[f]) + 3 * (p->pr_seg_ptr[f + 1] - p->pr_seg_ptr[f]) +
                       (p->in_seg_ptr.empty() ? 0 : 3 * (p->in_seg_ptr[f + 1] - p->in_seg_ptr[f]));
      const int nw = p->fused ? (load + 95) / 96 : std::max(1, (load + 95) / 96);  // (fused build: segments without prior / inertial records have no workgroup)
      sw_ptr[f + 1] = sw_ptr[f] + nw;
      for (int w = 0; w < nw; ++w) sw_seg.push_back(f);
    }
    p->n_seg_wg = sw_ptr[n_seg_];
    sw_seg.push_back(0);
    HIP_TRY(p->d_sw_ptr.upload(sw_ptr, s));
    HIP_TRY(p->d_sw_seg.upload(sw_seg, s));
    // k_group_gram work list: <= 12 landmarks per workgroup (measured at configs[1]: 8 / 12 / 16 / 24 / 32 per workgroup give a
    // Schur stage of 71.2 / 65.8 / 66.5 / 68.1 / 70.9 us: fewer, larger partials for k_assemble against less parallelism)
    const int per_wg = 12;
    std::vector<int> gw_ptr(p->n_cp + 1, 0), gw_cf;
    if (p->fused) {  // the chunks of the fused build take the place of the k_group_gram workgroups
      HIP_TRY(p->d_gw_ptr.upload(p->h_gw_ptr, s));
      HIP_TRY(p->d_gw_cf.upload(p->h_gw_cf, s));
      HIP_TRY(p->d_ch_ptr.upload(p->h_ch_ptr, s));
      HIP_TRY(p->d_ch_desc.upload(p->h_ch_desc, s));
      p->n_group_wg = int(p->h_ch_ptr.size()) - 1;
    } else {
      for (int c = 0; c < p->n_cp; ++c) {
        const int cnt = vs.cf_ptr[c + 1] - vs.cf_ptr[c], nw = (cnt + per_wg - 1) / per_wg;
        gw_ptr[c + 1] = gw_ptr[c] + nw;
        for (int w = 0; w < nw; ++w) gw_cf.push_back(c);
      }
      p->n_group_wg = gw_ptr[p->n_cp];
      gw_cf.push_back(0);
      HIP_TRY(p->d_gw_ptr.upload(gw_ptr, s));
      HIP_TRY(p->d_gw_cf.upload(gw_cf, s));
    }
    HIP_TRY(p->d_segP.reserve(size_t(p->n_seg_wg) * (size_t(nca) * nca + nca) + 1));
    HIP_TRY(p->d_grpQ.reserve(size_t(p->n_group_wg) * (size_t(ntile) * 36 + (p->fused ? 3 : 1) * 6 * vs.bw) + 1));
  }
  HIP_TRY(p->d_state.reserve(1));

  Tables& T = p->T;
  std::memset(&T, 0, sizeof(T));
  T.sp = Spline{k, p->n_cp, p->t0, p->dt, 1.0 / p->dt, p->rot_const, p->trans_const};
  T.basis = make_basis_coef(k);
  T.cp = p->d_cp.p, T.cp_cand = p->d_cp_cand.p, T.cp_const = p->d_cp_const.p;
  T.cam = p->d_cam.p, T.n_cam = p->n_cam, T.sensor = p->d_sensor.p;
  T.n_lm = p->n_lm, T.lm = p->d_lm.p, T.lm_cand = p->d_lm_cand.p, T.lm_const = p->d_lm_const.p;
  T.lm_ptr = p->d_lm_ptr.p, T.lm_cfirst = p->d_lm_cfirst.p, T.lm_ncp = p->d_lm_ncp.p, T.lm_yoff = p->d_lm_yoff.p, T.cf_ptr = p->d_cf_ptr.p;
  T.lm_scale = p->d_lm_scale.p, T.lm_L = p->d_lm_L.p, T.lm_yhat = p->d_lm_yhat.p, T.lm_sb = p->d_lm_sb.p, T.lm_D2 = p->d_lm_D2.p;
  T.lm_part = p->d_lm_part.p, T.n_lm_part = p->fused ? p->nb_vis : (p->n_lm + kBlock / 64 - 1) / (kBlock / 64), T.lm_gmax = p->d_lm_gmax.p, T.Y = p->d_Y.p;
  {
    int n_obs = p->n_lm;
    while (n_obs > 0 && vs.lm_ptr[n_obs] == vs.lm_ptr[n_obs - 1]) --n_obs;
    T.n_obs_lm = n_obs;
  }
  T.n_vis = n_vis, T.v_stamp = p->d_v_stamp.p, T.v_meas = p->d_v_meas.p, T.v_lm = p->d_v_lm.p, T.v_info = p->d_v_info.p;
  T.v_first = p->d_v_first.p, T.v_pos = p->d_v_pos.p, T.v_rec = p->d_v_rec.p, T.v_rec_alt = p->d_v_rec_alt.p, T.v_seg_ptr = p->d_v_seg_ptr.p;
  T.n_pri = n_pri, T.p_stamp = p->d_p_stamp.p, T.p_meas = p->d_p_meas.p, T.p_sensor = p->d_p_sensor.p, T.p_first = p->d_p_first.p;
  T.p_rec = p->d_p_rec.p, T.p_seg_ptr = p->d_p_seg_ptr.p;
  T.n_ine = n_ine, T.i_stamp = p->d_i_stamp.p, T.i_meas = p->d_i_meas.p, T.i_first = p->d_i_first.p, T.i_first_bias = p->d_i_first_bias.p;
  T.i_rec = p->d_i_rec.p, T.i_seg_ptr = p->d_i_seg_ptr.p, T.imu = p->d_imu.p;
  T.bias_basis = make_basis_coef(p->kb), T.kb = p->kb, T.n_bias = p->has_imu ? p->n_bias : 0, T.bias_t0 = p->bias_t0, T.bias_dt = p->bias_dt;
  T.bias_g = p->d_bias_g.p, T.bias_a = p->d_bias_a.p, T.bias_g_cand = p->d_bias_g_cand.p, T.bias_a_cand = p->d_bias_a_cand.p;
  T.gravity = p->d_gravity.p, T.gravity_cand = p->d_gravity_cand.p, T.bias_const = p->bias_const, T.gravity_const = p->gravity_const;
  T.inertial_literal = p->inertial_mode == HS_INERTIAL_AS_REFERENCE;
  T.nb = p->has_imu ? 6 * p->n_bias + 2 : 0;
  T.n_seg = n_seg, T.bw = vs.bw, T.np = np;
  T.scale_p = p->d_scale_p.p, T.Sb = p->d_Sb.p, T.Ub = p->d_Ub.p, T.Ubk = p->d_Ubk.p, T.g_s = p->d_g_s.p, T.g_full = p->d_g_full.p, T.D2p = p->d_D2p.p, T.gabs = p->d_gabs.p;
  T.step_p = p->d_step_p.p, T.delta_p = p->d_delta_p.p;
  T.cost_part = p->d_cost_part.p, T.cand_part = p->d_cand_part.p, T.n_cost_part = p->nb_vis + p->nb_pri + p->nb_ine;
  T.norm_part = p->d_norm_part.p, T.n_norm_part = nb_norm;
  T.xbuf = p->d_xbuf.p;
  T.xpart = p->d_xpart.p, T.gravity_part = p->d_gravity_part.p, T.segP = p->d_segP.p, T.grpQ = p->d_grpQ.p, T.gw_ptr = p->d_gw_ptr.p, T.gw_cf = p->d_gw_cf.p, T.sw_ptr = p->d_sw_ptr.p, T.sw_seg = p->d_sw_seg.p;
  T.xo_g = np * ncb, T.xo_gs = T.xo_g + np, T.xo_dj = T.xo_gs + np, T.xo_pb = T.xo_dj + np, T.xo_bb = T.xo_pb + np * nbd;
  T.xo_gb = T.xo_bb + nbd * nbd, T.xo_cost = T.xo_gb + nbd, T.xo_gmax = T.xo_cost + 1;
  T.ybuf = p->d_ybuf.p, T.ybuf2 = nullptr, T.y_split = np;
  T.fj[0] = FactorJob{T.Sb, T.g_s, T.Ub, T.Ubk, T.ybuf, nullptr, np / 6, -1};
  T.fj[1] = FactorJob{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, -1};
  T.xsol = p->d_xsol.p, T.join_flag = p->d_join.p, T.join_epoch = 0;
  T.debug_flags = std::getenv("HS_DEBUG_FLAGS") ? std::atoi(std::getenv("HS_DEBUG_FLAGS")) : 0;
  {  // the reversed copy feeds the far end of a two-ended factorisation and, as the lower band, every MFMA factorisation
    const bool la_ok = la_compute_waves(vs.bw) > 0, two_ended = la_ok && np / 6 >= 4 * vs.bw;
    const bool need = two_ended || (HS_AB(T.debug_flags, 131072) && mfma_window_tiles(vs.bw) > 0);
    T.Sb2 = need ? p->d_Sb2.p : nullptr, T.g2 = need ? p->d_g2.p : nullptr;
  }
  T.scale_b = p->d_scale_b.p, T.Spb = p->d_Spb.p, T.Sbb = p->d_Sbb.p, T.gb_s = p->d_gb_s.p, T.D2b = p->d_D2b.p;
  T.Zb = p->d_Zb.p, T.Cb = p->d_Cb.p, T.hb = p->d_hb.p, T.xb = p->d_xb.p, T.delta_b = p->d_delta_b.p, T.i_bias_ptr = p->d_i_bias_ptr.p, T.bfwd_start = p->d_bfwd_start.p;
  T.x_count1 = x_count1, T.xo_dec = x_count1;
  T.fused = p->fused ? 1 : 0, T.n_chunk = p->fused ? p->n_group_wg : 0, T.ch_ptr = p->d_ch_ptr.p, T.ch_desc = p->d_ch_desc.p;
  T.rank = p->rank, T.world = p->world;
  // HS_DEBUG_FLAGS (measurement switches only, never needed for correct operation):
  //    1 skip the backward sweep          2 skip the rank-6 updates (timing of the panel chain alone; results are garbage)
  //    4 one-ended pre-look-ahead factorisation kernel                16 phase timestamps of the factorisation -> hs_debug_read
  //   32 per-workgroup timestamps of the linearise / gram kernels   1024 no side stream for the segment partials
  // 2048 one-ended factorisation (no second workgroup)              8192 generalised backward sweep on the one-ended factor
  // 131072 k_band_factor_mfma (trailing window in f64 MFMA tiles) instead of the VALU factorisation kernels
  // 65536 single-wave register backward sweep (k_band_backward_w) instead of the four-wave LDS sweeps
  // 262144 eliminate / sweep the decoupled block rows of leading constant control points like any other     524288 border Cholesky in LDS
  // 1048576 inertial branch on the main stream    2097152 banded kernels instead of k_dense_factor    4194304 k_landmark<K,4,1> instead of k_landmark_rows
  // 8388608 five finalisation launches for a bordered single shard    16777216 k_commit launch for small windows    33554432 one cost launch per factor type
  // 268435456 backward sweeps one block row per step    536870912 bordered systems one-ended    1073741824 no speculative linearisation at the candidate    67108864 k_commit in every iteration of a speculative solve
  T.st = p->d_state.p;
  HIP_TRY(p->batch.flush(s));  // (the staging arena outlives this call: no host synchronisation)
  p->dirty = false;
  if (p->host_timing) {
    const auto host_t2 = std::chrono::steady_clock::now();
    p->host_prepare[0] = std::chrono::duration<double, std::milli>(host_t1 - host_t0).count();
    p->host_prepare[1] = std::chrono::duration<double, std::milli>(host_t2 - host_t1).count();
    p->host_ms[0] += p->host_prepare[0], p->host_ms[1] += p->host_prepare[1];
  }
  return HS_OK;
}

int reset_state(hs_problem* p, int max_iterations, double radius, int spec = 0) {
  k_reset_state<<<1, 64, 0, p->stream>>>(p->d_state.p, max_iterations, radius, spec);
  HIP_TRY(hipGetLastError());
  return HS_OK;
}

size_t cp_lds_bytes(const hs_problem* p) { return size_t(8) * p->n_cp * sizeof(double); }
template <int K>
size_t lin_lds_bytes(const hs_problem* p) {  // control points + one record slab per wave
  return (cp_lds_bytes(p) <= 24 * 1024 ? cp_lds_bytes(p) : 0) + size_t(lin_block<K>()) * (8 + 12 * K + 2) * sizeof(double);
}

__global__ void k_noop() {}

/// The side stream of the inertial branch and its three events. Created by hs_create and used once there: a stream gets its hardware
/// queue at its first submission, which — together with the first allocations — made the first optimize() with an IMU 10 ms long.
static int ensure_side_stream(hs_problem* p) {
  if (!p->side) {
    HIP_TRY(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_irec, hipEventDisableTiming));
    k_noop<<<1, 64, 0, p->side>>>();
    HIP_TRY(hipEventRecord(p->ev_join, p->side));
    HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_join, 0));
    k_noop<<<1, 64, 0, p->stream>>>();
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  return HS_OK;
}

/// `inertial_on_side` (the solve loop of bordered systems): the inertial branch of an iteration — k_linearize_inertial, then the border
/// gathers k_border_pb / _bb in launch_build — only meets the visual branch (k_linearize_visual -> k_landmark -> Gram
/// kernels -> k_assemble) at the segment Gram kernel (reads the inertial records) and at k_reduce_partials, and each branch fills a
/// fraction of the chip: they run on two streams. configs[2]: 293 us of kernels back to back -> 175 us on the critical path.
/// Fused build (p->fused): the visual factors are linearised by k_build_visual inside launch_build — nothing to do for them here, unless only
/// the cost is wanted (`visual_cost_only`: hs_cost, hs_solve with zero iterations), which the value-only kernel delivers.
template <int K>
int launch_linearize(hs_problem* p, bool inertial_on_side = false, bool visual_cost_only = false) {
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  p->side_imu = inertial_on_side && T.n_ine > 0 && T.nb > 0 && !(T.debug_flags & 1048576);  // A/B switch 1048576: one stream
  if (p->side_imu) {
    const int rc = ensure_side_stream(p);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(p->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0));
  }
  if (T.n_vis && !p->fused) k_linearize_visual<K><<<p->nb_vis, lin_block<K>(), lin_lds_bytes<K>(p), s>>>(T, T.v_rec, T.v_pos, 1, T.cost_part, nullptr);
  if (T.n_vis && p->fused && visual_cost_only) k_cost_visual<K><<<p->nb_vis, kBlock, cp_lds_bytes(p), s>>>(T, T.cp, T.lm, T.cost_part);
  if (T.n_pri) k_linearize_prior<K><<<p->nb_pri, kBlock, cp_lds_bytes(p), s>>>(T, T.p_rec, T.cost_part + p->nb_vis, nullptr);
  if (T.n_ine)
    k_linearize_inertial<K, 4><<<p->nb_ine, kInertialBlock * K, cp_lds_bytes(p), p->side_imu ? p->side : s>>>(T, T.i_rec, 1, T.cost_part + p->nb_vis + p->nb_pri,
                                                                                                        nullptr);
  if (p->side_imu) HIP_TRY(hipEventRecord(p->ev_irec, p->side));
  HIP_TRY(hipGetLastError());
  return HS_OK;
}

// ---- RCCL, loaded on first use ---------------------------------------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
      api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
      api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
      api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
      api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
      api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
      api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
      if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) api.lib = nullptr;
    }
  }
  return api.lib ? &api : nullptr;
}

int exchange(hs_problem* p, double* buf, int64_t count) {
  if (p->rccl_comm) {  // one in-place sum all-reduce on the library's stream, no host involvement
    const ncclResult_t r = rccl_api()->AllReduce(buf, buf, size_t(count), ncclDouble, ncclSum, static_cast<ncclComm_t>(p->rccl_comm), p->stream);
    if (r != ncclSuccess) HS_FAIL(HS_ERR_DEVICE, std::string("ncclAllReduce failed: ") + (rccl_api()->GetErrorString ? rccl_api()->GetErrorString(r) : "?"));
    return HS_OK;
  }
  if (!p->allreduce) return HS_OK;
  if (p->allreduce(p->allreduce_user, buf, count, p->stream) != 0) HS_FAIL(HS_ERR_DEVICE, "all-reduce hook reported a failure");
  return HS_OK;
}

static int border_zero_wgs(const Tables& T) { return std::min(64, (T.nb * T.nb + T.nb + kPbThreads - 1) / kPbThreads); }

/// `after_build` (stage timing of a fused build): recorded behind k_build_visual — the launch that linearises the visual factors belongs to
/// the "linearise" stage of hs_summary, what follows it (segment Gram of the prior / inertial records, assembly, finalisation) to "schur".
template <int K>
int launch_build(hs_problem* p, hipEvent_t after_build = nullptr) {
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  // k_seg_gram only needs the records, k_landmark -> k_group_gram records and landmarks: the two gram kernels share one launch
  // (k_gram_pair). A/B switch 1024: the previous arrangement, k_seg_gram on a side stream next to k_landmark -> k_group_gram.
  // (for small grids only — configs[1]: ~940 workgroups, Schur stage 66 -> 62 us. The pair holds 80 KB of LDS per workgroup, two per
  //  CU, where k_group_gram alone fits three: at configs[3], ~3 750 workgroups, the two streams are faster, 0.165 vs 0.181 ms)
  const bool fused = p->fused;
  const bool pair = !fused && T.n_lm > 0 && p->n_group_wg > 0 && p->n_group_wg + p->n_seg_wg <= 2048 && !(T.debug_flags & 1024);
  const bool side_imu = p->side_imu;       // (set by launch_linearize: the side stream is busy with the inertial branch)
  const bool fork = !fused && T.n_lm > 0 && !pair && !side_imu;
  // Fused build: linearisation, landmark elimination and both Gram terms of the visual factors in one launch; what remains for the segment
  // Gram kernel are the prior / inertial records (none on visual-only windows: no launch)
  if (fused) k_build_visual<K><<<p->nb_vis, kBlock, p->build_lds, s>>>(T, p->build_R, p->build_L, 1);
  if (fused && after_build) HIP_TRY(hipEventRecord(after_build, s));
  if (fork) {
    const int rc = ensure_side_stream(p);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(p->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0));
  }
  hipStream_t sb = side_imu ? p->side : s;  // stream of the border gathers
  if (side_imu && T.nb) {  // behind k_linearize_inertial on the side stream, next to k_landmark / the Gram kernels
    k_border_pb<K><<<dim3(T.sp.n_cp + border_zero_wgs(T), p->n_split), kPbThreads, 0, sb>>>(T);  // (+ zero-fill of the border-border block)
    k_border_bb<K><<<T.n_bias, kBlock, 0, sb>>>(T);
    HIP_TRY(hipEventRecord(p->ev_join, p->side));
  }
  bool irec_ready = !side_imu;  // the segment Gram kernel reads the inertial records: wait for the side stream's linearisation once
  auto need_irec = [&]() -> hipError_t {
    if (irec_ready) return hipSuccess;
    irec_ready = true;
    return hipStreamWaitEvent(s, p->ev_irec, 0);
  };
  if (!pair && !(side_imu && T.n_lm) && p->n_seg_wg) {
    HIP_TRY(need_irec());
    k_seg_gram<K><<<p->n_seg_wg, kBlock, kSegStage * sizeof(double), fork ? p->side : s>>>(T);
  }
  if (fork) HIP_TRY(hipEventRecord(p->ev_join, p->side));
  if (T.n_lm && !fused) {
    const int grid = (T.n_lm + kBlock / 64 - 1) / (kBlock / 64);
    if (6 * T.bw <= 128)
      k_landmark<K, 2, 2><<<grid, kBlock, 0, s>>>(T);
    else if (T.debug_flags & 4194304)  // A/B switch 4194304: one wave per landmark with four passes
      k_landmark<K, 4, 1><<<grid, kBlock, 0, s>>>(T);
    else  // long feature tracks (6 * bw <= kBlock is checked in prepare()): one workgroup per landmark, one wave per 64 rows of W
      k_landmark_rows<K, 4><<<T.n_lm, kBlock, 0, s>>>(T);
  }
  if (T.n_lm && p->n_group_wg && !fused) {
    const int ntile = T.bw * (T.bw + 1) / 2;
    const int batch = std::max(2, std::min(kGroupBatch, int(48 * 1024 / (size_t(18) * T.bw * sizeof(double)))));
    const size_t lds = std::max((size_t(batch) * 18 * T.bw + 4 * batch) * sizeof(double), size_t(128) * 42 * sizeof(double));
    const dim3 grid(p->n_group_wg);
    if (pair) {
      HIP_TRY(need_irec());
      const size_t lds2 = std::max(lds, kSegStage * sizeof(double));
      const dim3 grid2(p->n_group_wg + p->n_seg_wg);
      if (ntile <= kBlock)
        k_gram_pair<K, 1><<<grid2, kBlock, lds2, s>>>(T, batch, p->n_group_wg);
      else if (ntile <= 2 * kBlock)
        k_gram_pair<K, 2><<<grid2, kBlock, lds2, s>>>(T, batch, p->n_group_wg);
      else
        k_gram_pair<K, 4><<<grid2, kBlock, lds2, s>>>(T, batch, p->n_group_wg);
    } else if (ntile <= kBlock)
      k_group_gram<1><<<grid, kBlock, lds, s>>>(T, batch);
    else if (ntile <= 2 * kBlock)
      k_group_gram<2><<<grid, kBlock, lds, s>>>(T, batch);
    else
      k_group_gram<4><<<grid, kBlock, lds, s>>>(T, batch);
  }
  if (side_imu && T.n_lm && !pair && p->n_seg_wg) {  // (large grids with an IMU: the segment Gram kernel after the landmark chain, same stream)
    HIP_TRY(need_irec());
    k_seg_gram<K><<<p->n_seg_wg, kBlock, kSegStage * sizeof(double), s>>>(T);
  }
  if (fork) HIP_TRY(hipStreamWaitEvent(s, p->ev_join, 0));
  k_assemble<K><<<dim3(T.sp.n_cp, 6), kAsmThreads, 0, s>>>(T);
  if (T.nb && !side_imu) {
    k_border_pb<K><<<dim3(T.sp.n_cp + border_zero_wgs(T), p->n_split), kPbThreads, 0, s>>>(T);
    k_border_bb<K><<<T.n_bias, kBlock, 0, s>>>(T);
  }
  if (side_imu) HIP_TRY(hipStreamWaitEvent(s, p->ev_join, 0));  // border gathers done
  // Nothing to exchange (single shard): packing + bookkeeping are an extra workgroup of k_finalize_reduced, the border blocks further
  // ones that sum the accumulation splits themselves — one launch where the exchanging path has five (~5 us each on the chain).
  // A/B switch 8388608: the five launches.
  const bool reduce_here = !p->allreduce && !p->rccl_comm && p->world == 1 && !(T.nb && (T.debug_flags & 8388608));
  const int nb_wg = T.nb ? std::min(256, ((T.np + T.nb) * T.nb + kBlock - 1) / kBlock) : 0;
  if (T.nb && !reduce_here)
    k_reduce_partials<<<std::min(1024, (T.xo_bb - T.xo_pb + kBlock - 1) / kBlock), kBlock, 0, s>>>(T, p->n_split, T.xo_pb);
  if (!reduce_here) k_pack_exchange<<<1, kBlock, 0, s>>>(T, 0);
  HIP_TRY(hipGetLastError());
  const int rc = exchange(p, T.xbuf, T.x_count1);  // one RCCL all-reduce of [S | g | diag | cost] per linearisation
  if (rc) return rc;
  k_finalize_reduced<<<T.sp.n_cp + (reduce_here ? 1 + nb_wg : 0), kBlock, 0, s>>>(T, p->n_split);  // + 1: packing / bookkeeping workgroup, + border
  if (T.nb && !reduce_here) k_finalize_border<<<nb_wg, kBlock, 0, s>>>(T);
  if (!reduce_here) k_cost_reduce<<<1, kBlock, 0, s>>>(T);
  HIP_TRY(hipGetLastError());
  return HS_OK;
}

/// Window size (in 16 x 16 tiles) of the MFMA factorisation for a band of bw blocks: 16 NT >= 6 bw + 12; 0: not supported.
int mfma_window_tiles(int bw) {
  for (int nt : {6, 9}) {  // (NT = 10 would cover bw <= 24: 256 VGPRs + scratch, and wrong results on gfx950 — not instantiated)
    if (6 * bw + 12 <= 16 * nt) return nt;
  }
  return 0;
}

#if HS_PROFILE_HOOKS
template <int NT, int NC>
hipError_t launch_mfma(const Tables& T, int grid, hipStream_t s) {
  static bool attr = false;
  const size_t lds = size_t(MfmaGeom<NT>::kTotal) * sizeof(double);
  if (!attr) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_mfma<NT, NC>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    attr = true;
  }
  k_band_factor_mfma<NT, NC><<<grid, 64 * (NC + 3), lds, s>>>(T);
  return hipGetLastError();
}

void launch_backward_w(const Tables& T, const BackJob& j0, const BackJob& j1, int m_mid, int grid, hipStream_t s) {
  const size_t lds = size_t(T.np) * sizeof(double);
  if (T.bw <= kBackBlocks)
    k_band_backward_w<1><<<grid, 64, lds, s>>>(T, j0, j1, m_mid);
  else if (T.bw <= 2 * kBackBlocks)
    k_band_backward_w<2><<<grid, 64, lds, s>>>(T, j0, j1, m_mid);
  else if (T.bw <= 4 * kBackBlocks)
    k_band_backward_w<4><<<grid, 64, lds, s>>>(T, j0, j1, m_mid);
  else
    k_band_backward_w<5><<<grid, 64, lds, s>>>(T, j0, j1, m_mid);
}

#endif  // HS_PROFILE_HOOKS

/// Dense Cholesky of the border Schur complement + solve for the border unknowns (one workgroup).
static hipError_t launch_border_solve(const Tables& T, hipStream_t s) {
  if (T.nb + 1 <= 128 && !(T.debug_flags & 524288)) {  // trailing matrix in registers (A/B switch 524288: the LDS version)
    const int R = std::max(4, (T.nb + 1 + 15) / 16), N = 16 * R;
    const size_t lds = (size_t(4) * N + size_t(T.nb) * (N + 1) + T.nb) * sizeof(double);
    switch (R) {
      case 4: k_border_solve_reg<4><<<1, kBlock, lds, s>>>(T); break;
      case 5: k_border_solve_reg<5><<<1, kBlock, lds, s>>>(T); break;
      case 6: k_border_solve_reg<6><<<1, kBlock, lds, s>>>(T); break;
      case 7: k_border_solve_reg<7><<<1, kBlock, lds, s>>>(T); break;
      default: k_border_solve_reg<8><<<1, kBlock, lds, s>>>(T); break;
    }
  } else {
    k_border_solve<<<1, kBlock, (size_t(T.nb + 1) * (T.nb + 1) + T.nb) * sizeof(double), s>>>(T);
  }
  return hipGetLastError();
}

int launch_factor(hs_problem* p) {
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  const int ncb = 6 * T.bw;
  const size_t chol_lds = (size_t(24) * (ncb + 2) + size_t(T.np)) * sizeof(double);
  const size_t la_lds = (size_t(42) * (ncb + 2) + size_t(T.np) + 48) * sizeof(double);
  const bool legacy = T.debug_flags & 4;  // A/B switch: pre-look-ahead kernel
  // Factoring from both ends at once (visual-only systems, look-ahead kernel, window long enough to pay for the junction)
  const int n_blk = T.np / 6, w_mid = T.bw - 1;
  const bool la_ok = !legacy && la_compute_waves(T.bw) > 0;
  const int la_ncw = la_compute_waves(T.bw);
  const int nt = HS_AB(T.debug_flags, 131072) ? mfma_window_tiles(T.bw) : 0;  // A/B switch 131072 (profiling builds): k_band_factor_mfma instead of the VALU kernels
  // (bordered systems — bias splines + gravity — too: the forward sweep of the border columns follows the two-ended elimination order,
  //  k_border_forward2; A/B switch 536870912: bordered systems one-ended)
  const bool two_ended = (la_ok || nt) && (T.nb == 0 || (!nt && !(T.debug_flags & 536870912) && (T.nb + kBorderCols - 1) / kBorderCols <= 512)) &&
                         n_blk >= 4 * T.bw && T.Sb2 && !(T.debug_flags & 2048);  // (512: flag words of k_border_forward2's column groups)
#if HS_PROFILE_HOOKS
  auto run_mfma = [&](const Tables& TT, int grid) -> hipError_t {
    switch (nt) {
      case 6: return launch_mfma<6, 3>(TT, grid, s);
      default: return launch_mfma<9, 3>(TT, grid, s);
    }
  };
#else
  auto run_mfma = [&](const Tables&, int) -> hipError_t { return hipErrorNotSupported; };  // (nt == 0: never reached)
#endif
  if (two_ended) {
    // The near end takes a few block rows more than the far end: the far end still has to hand its trailing window over (~7 us,
    // i.e. ~4 steps: window through HBM + agent-scope release) before the near end can pass the junction. With an even split
    // workgroup 0 waited 13 us there (tools/chol_phase_timing.py).
    const int m = std::min((n_blk - w_mid) / 2 + 3, n_blk - w_mid - w_mid), mB = n_blk - w_mid - m;  // (+2 / +3 / +4: 132.0 / 130.5 / 132.1 us)
    Tables T2 = T;
    T2.fj[0] = FactorJob{T.Sb, T.g_s, T.Ub, T.Ubk, T.ybuf, p->d_win.p, m + w_mid, m};
    T2.fj[1] = FactorJob{p->d_Sb2.p, p->d_g2.p, p->d_Ub2.p, p->d_Ubk2.p, p->d_ybuf2.p, p->d_win.p, mB, -1};
    T2.mj[0] = MfmaJob{p->d_Sb2.p, T.g_s, T.Ub, T.Ubk, T.ybuf, p->d_win.p, m + w_mid, m, m + w_mid, INT_MAX, 0, p->d_Vb.p + size_t(T.np) * (6 * T.bw)};
    T2.mj[1] = MfmaJob{T.Sb, p->d_g2.p, p->d_Ub2.p, p->d_Ubk2.p, p->d_ybuf2.p, p->d_win.p, mB, -1, mB + w_mid, mB, 1, p->d_Vb.p + size_t(T.np) * (6 * T.bw)};
    T2.join_epoch = ++p->join_epoch;
    if (nt)
      HIP_TRY(run_mfma(T2, 2));
    else
      if (la_ncw == 3)
        k_band_factor_la<1, 3><<<2, la_threads(3), la_lds, s>>>(T2);
      else
        k_band_factor_la<1, 4><<<2, la_threads(4), la_lds, s>>>(T2);
    if (T.nb) {  // bordered system: Z = U^-T S_pb in the two-ended elimination order, border Schur complement and solve, y' = y - Z x_b
      Tables Tb = T2;
      Tb.ybuf2 = p->d_ybuf2.p, Tb.y_split = 6 * (m + w_mid);
      Tb.join_epoch = ++p->join_epoch;
      const int n_groups = (T.nb + kBorderCols - 1) / kBorderCols;
      HIP_TRY(p->d_bf_handover.reserve(size_t(n_groups) * 6 * w_mid * kBorderCols + 1));
      const int fwd_threads = std::max(128, 64 * ((6 * w_mid + 63) / 64));  // one lane per pending row
      const int local_rows = (!p->allreduce && !p->rccl_comm && p->world == 1) ? 1 : 0;
      k_border_forward2<<<dim3(n_groups, 2), fwd_threads, size_t(T.np) * kBorderLd * sizeof(double), s>>>(
          Tb, BfJob{T.Ub, T.Ubk, m + w_mid, 0}, BfJob{p->d_Ub2.p, p->d_Ubk2.p, mB, 1}, m, 0, local_rows, p->d_bf_handover.p);
      const int n_tiles = (T.nb + kSchurTile - 1) / kSchurTile;
      k_border_schur<<<dim3(n_tiles, n_tiles), kBlock, 0, s>>>(Tb, 0, local_rows, m);  // (rows from the junction on are never skipped)
      HIP_TRY(launch_border_solve(Tb, s));
      k_border_apply<<<(T.np + kBlock / 64 - 1) / (kBlock / 64), kBlock, 0, s>>>(Tb);
    }
    Tables T3 = T2;
    T3.join_epoch = ++p->join_epoch;
    const BackJob j0{T.Ub, T.Ubk, T.ybuf, p->d_Vb.p, p->d_yt.p, m + w_mid, 0, 0};
    const BackJob j1{p->d_Ub2.p, p->d_Ubk2.p, p->d_ybuf2.p, p->d_Vb2.p, p->d_yt2.p, mB, w_mid, 1};
    // (the two older sweeps are kept as measurement switches for visual-only systems, the shape they were measured on; they do not write
    //  the border's step outputs)
    const bool sweep_w = HS_AB(T.debug_flags, 65536) && !T.nb, sweep_rows = HS_AB(T.debug_flags, 268435456) && !T.nb;
#if HS_PROFILE_HOOKS
    if (sweep_w) k_premultiply<<<m + w_mid + mB, 128, 0, s>>>(T3, j0, j1, m + w_mid);
#endif
    if (!sweep_w) {  // (A/B switch 65536: single-wave register sweep)
      const size_t g_lds = size_t(6 * (T.bw - 1)) * (6 * (T.bw - 1) | 1) * sizeof(double);  // given-column block of the far sweep
#if HS_PROFILE_HOOKS
      if (sweep_rows)  // A/B switch 268435456: one block row per step
        k_band_backward2<<<2, kCholThreads, 2 * size_t(T.np) * sizeof(double) + g_lds, s>>>(T3, j0, j1, m);
      else
#endif
        // super-blocks of four block rows; the inverses of the diagonal super-blocks come from extra workgroups of the launch
        k_band_backward_sb<<<2 + (m + w_mid + kSb - 1) / kSb + (mB + kSb - 1) / kSb, kCholThreads,
                             std::max((2 * size_t(T.np) + 32) * sizeof(double) + g_lds, size_t(3 * kSbN * (kSbN + 1)) * sizeof(double)), s>>>(T3, j0, j1, m, 2, 0);
    } else {
#if HS_PROFILE_HOOKS
      launch_backward_w(T3, j0, j1, m, 2, s);
#endif
    }
    (void)sweep_rows;
    HIP_TRY(hipGetLastError());
    return HS_OK;
  }
  // One-ended. Block rows of the leading constant control points are decoupled (k_factor_decoupled_rows): the dependency chain of the
  // factorisation starts behind them — the same kernels on the trailing sub-matrix (the band storage is row relative: pointer offsets).
  const int f0 = (T.debug_flags & 262144) ? 0 : std::min(p->frozen_prefix, n_blk - 1);  // A/B switch 262144: eliminate every block row
  Tables Tf = T;
  const int n_eff = n_blk - f0;
  const bool dense = !nt && !(T.debug_flags & 2097152) && T.bw > 14 && n_eff <= 2 * T.bw &&
                     dense_factor_fits(n_eff, std::min(T.bw, n_eff));  // A/B switch 2097152: banded kernels
  if (f0 > 0) {
    if (!dense) k_factor_decoupled_rows<<<f0, 64, 0, s>>>(T, f0);  // (the dense kernel writes them with extra workgroups of its own launch)
    Tf.Sb += size_t(6 * f0) * ncb, Tf.g_s += 6 * f0, Tf.Ub += size_t(6 * f0) * ncb, Tf.Ubk += size_t(24) * f0, Tf.ybuf += 6 * f0, Tf.np -= 6 * f0;
    Tf.fj[0] = FactorJob{Tf.Sb, Tf.g_s, Tf.Ub, Tf.Ubk, Tf.ybuf, nullptr, Tf.np / 6, -1};
  }
  // short systems with window-wide bands (the sliding-window replay): every band tile in a register for the whole factorisation
  if (dense) {
    k_dense_factor<<<1 + f0, kDenseThreads, (size_t(12) * (ncb + 8) + size_t(32) * n_eff) * sizeof(double), s>>>(Tf, f0);
  } else if (nt) {
    Tables T1 = Tf;
    // (the lower-band rows come from the reversed copy, whose rows are counted from the END of the matrix: no offset)
    T1.mj[0] = MfmaJob{p->d_Sb2.p, Tf.g_s, Tf.Ub, Tf.Ubk, Tf.ybuf, nullptr, n_blk - f0, -1, n_blk - f0, INT_MAX, 0, p->d_Vb.p + size_t(T.np) * (6 * T.bw)};
    T1.mj[1] = T1.mj[0];
    HIP_TRY(run_mfma(T1, 1));
  } else if (la_ok && la_ncw == 3)
    k_band_factor_la<1, 3><<<1, la_threads(3), la_lds, s>>>(Tf);
  else if (la_ok)
    k_band_factor_la<1, 4><<<1, la_threads(4), la_lds, s>>>(Tf);
  // (two tiles per lane need 168 accumulator registers: with six waves per workgroup the budget is 256 and the look-ahead
  //  kernel spills in its update loop - wider bands stay on the kernel below)
  else if (T.bw * T.bw <= kCholThreads)
    k_band_factor<1><<<1, kCholThreads + kCholIo, chol_lds, s>>>(Tf);
  else if (T.bw <= 21)  // two tiles per lane; the IO wave moves 12 x 64 entries per block row: 6 (6 bw + 1) <= 768 <=> bw <= 21
    k_band_factor<2><<<1, kCholThreads + kCholIo, chol_lds, s>>>(Tf);  // (bw = 22 dropped entries of every block row in round 1:
                                                                       //  found by the lock-step replay, tests/test_host_driver.py)
  else  // long feature tracks: trailing window in L2 instead of registers
    k_band_factor_wide<<<1, kWideThreads, size_t(12) * (ncb + 2) * sizeof(double), s>>>(Tf);
  if (T.nb) {  // bordered system (bias splines + gravity)
    const int fwd_threads = std::max(128, 64 * ((6 * (T.bw - 1) + 63) / 64));  // one lane per pending row
    // the first non-zero row of a border column follows from the inertial record table — of ALL shards: a shard of a distributed solve
    // only skips the rows of the constant control points (which every shard agrees on)
    const int local_rows = (!p->allreduce && !p->rccl_comm && p->world == 1) ? 1 : 0;
    k_border_forward<<<(T.nb + kBorderCols - 1) / kBorderCols, fwd_threads, size_t(T.np) * kBorderLd * sizeof(double), s>>>(T, f0, local_rows);
    const int nt = (T.nb + kSchurTile - 1) / kSchurTile;
    k_border_schur<<<dim3(nt, nt), kBlock, 0, s>>>(T, f0, local_rows, n_blk);
    HIP_TRY(launch_border_solve(T, s));
    k_border_apply<<<(T.np + kBlock / 64 - 1) / (kBlock / 64), kBlock, 0, s>>>(T);
  }
#if HS_PROFILE_HOOKS
  if ((T.debug_flags & 8192) && !T.nb) {  // A/B: the generalised sweep on the whole system
    const BackJob j0{T.Ub, T.Ubk, T.ybuf, nullptr, nullptr, T.np / 6, 0, 0};
    k_band_backward2<<<1, kCholThreads, 2 * size_t(T.np) * sizeof(double), s>>>(T, j0, j0, -1);
    k_step_outputs<<<1, kBlock, 0, s>>>(T);
  } else
#endif
  if (!HS_AB(T.debug_flags, 65536) || T.nb) {  // (A/B switch 65536: single-wave register sweep — visual-only systems, the shape it was measured on)
    if (6 * (T.bw - 1) <= 96 && !(T.debug_flags & 268435456)) {  // super-blocks of four block rows: one lane pair per pending row, 96 pairs
      Tables T3 = T;
      T3.join_epoch = ++p->join_epoch;
      const BackJob j0{T.Ub, T.Ubk, T.ybuf, p->d_Vb.p, p->d_yt.p, T.np / 6, 0, 0};
      k_band_backward_sb<<<1 + (T.np / 6 + kSb - 1) / kSb, kCholThreads,
                           std::max((2 * size_t(T.np) + 32) * sizeof(double), size_t(3 * kSbN * (kSbN + 1)) * sizeof(double)), s>>>(T3, j0, j0, -1, 1, f0);
    } else {  // wide bands (long feature tracks): one block row per step, one lane per pending row
      k_band_backward<<<1, kCholThreads, 2 * size_t(T.np) * sizeof(double), s>>>(T, f0);
    }
  } else {
#if HS_PROFILE_HOOKS
    const BackJob j0{T.Ub, T.Ubk, T.ybuf, p->d_Vb.p, p->d_yt.p, T.np / 6, 0, 0};
    k_premultiply<<<T.np / 6, 128, 0, s>>>(T, j0, j0, T.np / 6);
    launch_backward_w(T, j0, j0, -1, 1, s);
#endif
  }
  HIP_TRY(hipGetLastError());
  return HS_OK;
}

/// Speculative solves (visual-only windows, on one shard or on every shard of a distributed solve): the candidate is LINEARISED instead of only costed, unless this is the last
/// iteration of the solve: its records land in the record buffer that does not hold the current point and become the current
/// linearisation if the step is accepted (decide_step flips DevState::rec_sel), so that the next iteration starts at k_landmark — after an
/// accepted step and after a rejected one (the records of the unchanged current point are still there: today's path linearises again).
/// One linearise launch (16 us at configs[1]) replaces a cost launch (7.7 us) + a linearise launch per iteration.
static bool speculative_solve(const hs_problem* p) {
  const Tables& T = p->T;
  return !p->fused && T.n_vis > 0 && !T.n_pri && !T.n_ine && !T.nb && !(T.debug_flags & 1073741824);  // A/B switch (shards of a distributed solve too: the decision is replicated)
}
/// Fused build: a visual-only window keeps an accepted candidate in the candidate buffers (k_build_visual reads it from there, the next
/// k_backsub_retract copies it to x on its way): no k_commit launch per iteration. Other windows commit (their prior / inertial kernels read x).
static bool fused_visual_only(const hs_problem* p) {
  const Tables& T = p->T;
  return p->fused && !T.n_pri && !T.n_ine && !T.nb;
}

/// Small problems: the decision kernel copies the accepted candidate to x itself (single shard). A/B switch 16777216: always k_commit.
static bool commit_inline(const hs_problem* p) {
  const Tables& T = p->T;
  return !p->allreduce && !p->rccl_comm && 8 * T.sp.n_cp + 3 * T.n_lm + 8 * T.n_bias <= kCommitInline && !(T.debug_flags & 16777216);
}

template <int K>
int launch_update(hs_problem* p, bool linearize_candidate = false, bool deferred_commit = false, hipEvent_t* lin_events = nullptr) {
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  if (p->fused) {  // candidate point, landmark back-substitution and the visual candidate cost per chunk, one launch
    k_update_visual<K><<<p->nb_vis + T.n_norm_part, kBlock, size_t(update_lds_doubles(T.bw, p->build_R, p->build_L)) * 8, s>>>(T, p->build_R, p->build_L, p->nb_vis);
    if (T.n_pri) k_cost_prior<K><<<p->nb_pri, kBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.cand_part + p->nb_vis);
    if (T.n_ine)
      k_cost_inertial<K, 4><<<p->nb_ine, kInertialBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.bias_g_cand, T.bias_a_cand, T.gravity_cand,
                                                                          T.cand_part + p->nb_vis + p->nb_pri);
  } else
    k_backsub_retract<<<T.n_lm_part + T.n_norm_part, kBlock, 0, s>>>(T);
  if (p->fused) {
  } else if (linearize_candidate) {
    if (lin_events) HIP_TRY(hipEventRecord(lin_events[0], s));  // stage timing: this launch is booked under "linearise", not "update"
    k_linearize_visual<K><<<p->nb_vis, lin_block<K>(), lin_lds_bytes<K>(p), s>>>(T, nullptr, T.v_pos, 1, T.cand_part, nullptr, T.cp_cand, T.lm_cand);
    if (lin_events) HIP_TRY(hipEventRecord(lin_events[1], s));
  } else if ((T.n_ine || T.n_pri) && !(T.debug_flags & 33554432)) {  // one launch for all factor types (A/B switch 33554432: one per type)
    k_cost_all<K, 4><<<p->nb_vis + p->nb_pri + p->nb_ine, kBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.lm_cand, T.bias_g_cand, T.bias_a_cand, T.gravity_cand,
                                                                                       T.cand_part, p->nb_vis, p->nb_pri);
  } else {
    if (T.n_vis) k_cost_visual<K><<<p->nb_vis, kBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.lm_cand, T.cand_part);
    if (T.n_pri) k_cost_prior<K><<<p->nb_pri, kBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.cand_part + p->nb_vis);
    if (T.n_ine)
      k_cost_inertial<K, 4><<<p->nb_ine, kInertialBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.bias_g_cand, T.bias_a_cand, T.gravity_cand,
                                                                          T.cand_part + p->nb_vis + p->nb_pri);
  }
  const bool local_decision = !p->allreduce && !p->rccl_comm;  // single shard: decide in the packing kernel
  const bool inline_commit = commit_inline(p);
  const bool cps_here = p->fused && deferred_commit;  // fused path: the decision kernel commits the control points, the landmarks stay deferred
  k_pack_decision<<<1, kBlock, 0, s>>>(T, inline_commit ? 2 : local_decision ? (cps_here ? 3 : 1) : 0);
  HIP_TRY(hipGetLastError());
  const int rc = exchange(p, T.xbuf + T.xo_dec, 5);  // candidate cost + norms + landmark-side model-cost terms
  if (rc) return rc;
  if (!local_decision) k_decide<<<1, kBlock, 0, s>>>(T, cps_here ? 1 : 0);
  const int nb_commit = std::max((std::max(8 * T.sp.n_cp, 3 * T.n_lm) + kBlock - 1) / kBlock, 1);  // one element per lane
  // (deferred: speculative solves of larger problems — the next iteration's k_backsub_retract copies the accepted candidate to x on its way,
  //  hs_solve launches k_commit once behind the last iteration)
  if (!inline_commit && !deferred_commit) k_commit<<<nb_commit, kBlock, 0, s>>>(T);
  HIP_TRY(hipGetLastError());
  return HS_OK;
}

static void launch_commit(hs_problem* p) {
  const Tables& T = p->T;
  const int nb_commit = std::max((std::max(8 * T.sp.n_cp, 3 * T.n_lm) + kBlock - 1) / kBlock, 1);
  k_commit<<<nb_commit, kBlock, 0, p->stream>>>(T);
}

/// First use of a kernel costs ~0.35 ms of host time (the runtime builds its kernel object lazily); a solve touches ~25 different kernels,
/// which showed up as a 9 ms hs_solve on the first optimize() of a process (HS_HOST_TIMING=2: "launches" of call 0). hs_create resolves
/// the kernels of the solve path up front, once per process and device; what remains on the first call is the allocation of the tables.
template <int K>
static void warm_kernels_of_order() {
  hipFuncAttributes fa;
  const void* kernels[] = {
      reinterpret_cast<const void*>(&k_build_visual<K>), reinterpret_cast<const void*>(&k_update_visual<K>), reinterpret_cast<const void*>(&k_linearize_visual<K>), reinterpret_cast<const void*>(&k_linearize_prior<K>),
      reinterpret_cast<const void*>(&k_linearize_inertial<K, 4>), reinterpret_cast<const void*>(&k_landmark<K, 2, 2>),
      reinterpret_cast<const void*>(&k_landmark<K, 4, 1>), reinterpret_cast<const void*>(&k_landmark_rows<K, 4>),
      reinterpret_cast<const void*>(&k_gram_pair<K, 1>), reinterpret_cast<const void*>(&k_gram_pair<K, 2>), reinterpret_cast<const void*>(&k_gram_pair<K, 4>),
      reinterpret_cast<const void*>(&k_seg_gram<K>), reinterpret_cast<const void*>(&k_assemble<K>), reinterpret_cast<const void*>(&k_border_pb<K>),
      reinterpret_cast<const void*>(&k_border_bb<K>), reinterpret_cast<const void*>(&k_cost_visual<K>), reinterpret_cast<const void*>(&k_cost_prior<K>),
      reinterpret_cast<const void*>(&k_cost_inertial<K, 4>), reinterpret_cast<const void*>(&k_cost_all<K, 4>),
      reinterpret_cast<const void*>(&k_process_tracks<K>), reinterpret_cast<const void*>(&k_sample_trajectory<K>)};
  for (const void* k : kernels) (void)hipFuncGetAttributes(&fa, k);
}
static void warm_kernels(int device) {
  static std::mutex mu;
  static std::vector<int> done;
  std::lock_guard<std::mutex> lock(mu);
  if (std::find(done.begin(), done.end(), device) != done.end()) return;
  done.push_back(device);
  hipFuncAttributes fa;
  const void* kernels[] = {
      reinterpret_cast<const void*>(&k_group_gram<1>), reinterpret_cast<const void*>(&k_group_gram<2>), reinterpret_cast<const void*>(&k_group_gram<4>),
      reinterpret_cast<const void*>(&k_pack_exchange), reinterpret_cast<const void*>(&k_cost_reduce), reinterpret_cast<const void*>(&k_finalize_reduced),
      reinterpret_cast<const void*>(&k_finalize_border), reinterpret_cast<const void*>(&k_reduce_partials), reinterpret_cast<const void*>(&k_factor_decoupled_rows),
      reinterpret_cast<const void*>(&k_dense_factor), reinterpret_cast<const void*>(&k_band_factor_wide), reinterpret_cast<const void*>(&k_band_factor<1>),
      reinterpret_cast<const void*>(&k_band_factor<2>), reinterpret_cast<const void*>(&k_band_factor_la<1, 3>), reinterpret_cast<const void*>(&k_band_factor_la<1, 4>),
      reinterpret_cast<const void*>(&k_band_backward), reinterpret_cast<const void*>(&k_band_backward_sb), reinterpret_cast<const void*>(&k_border_forward),
      reinterpret_cast<const void*>(&k_border_forward2),
      reinterpret_cast<const void*>(&k_border_schur), reinterpret_cast<const void*>(&k_border_solve), reinterpret_cast<const void*>(&k_border_solve_reg<4>),
      reinterpret_cast<const void*>(&k_border_solve_reg<5>), reinterpret_cast<const void*>(&k_border_solve_reg<6>), reinterpret_cast<const void*>(&k_border_solve_reg<7>),
      reinterpret_cast<const void*>(&k_border_solve_reg<8>), reinterpret_cast<const void*>(&k_border_apply), reinterpret_cast<const void*>(&k_backsub_retract),
      reinterpret_cast<const void*>(&k_pack_decision), reinterpret_cast<const void*>(&k_decide), reinterpret_cast<const void*>(&k_commit),
      reinterpret_cast<const void*>(&k_reset_state), reinterpret_cast<const void*>(&k_scatter_uploads)};
  for (const void* k : kernels) (void)hipFuncGetAttributes(&fa, k);
  warm_kernels_of_order<4>();
  warm_kernels_of_order<6>();
}

int set_func_attributes(hs_problem* p) {
  // opt in to > 64 KiB dynamic LDS for the factorisation
  hipFuncAttributes fa;
  HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_band_factor<2>)));
  p->chol_lds_max = 160 * 1024 - int(fa.sharedSizeBytes);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor<1>), hipFuncAttributeMaxDynamicSharedMemorySize, p->chol_lds_max));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor<2>), hipFuncAttributeMaxDynamicSharedMemorySize, p->chol_lds_max));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_la<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, p->chol_lds_max));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_la<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, p->chol_lds_max));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seg_gram<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seg_gram<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_solve_reg<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_solve_reg<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_solve_reg<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_solve_reg<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_pair<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_pair<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_pair<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_pair<6, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_pair<6, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_pair<6, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_visual<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_visual<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_visual<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_visual<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linearize_visual<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linearize_visual<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_forward), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_forward2), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
#if HS_PROFILE_HOOKS
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_backward2), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
#endif
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_backward_sb), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_backward), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_solve), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  return HS_OK;
}

}  // namespace

extern "C" {

int hs_version(void) { return 1; }
const char* hs_arch(void) { return "gfx950"; }

int hs_create(int device, void* stream, hs_problem** out) {
  if (!out) return HS_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return HS_ERR_DEVICE;
  hs_problem* p = new hs_problem();
  p->device = device;
  if (const char* e = std::getenv("HS_REFERENCE_LITERAL")) p->inertial_mode = std::atoi(e) ? HS_INERTIAL_AS_REFERENCE : HS_INERTIAL_EXACT;
  if (const char* e = std::getenv("HS_STAGE_TIMING")) p->stage_timing = std::atoi(e) != 0;
  if (const char* e = std::getenv("HS_HOST_TIMING")) p->host_timing = std::atoi(e) != 0;
  if (hipSetDevice(device) != hipSuccess) {
    delete p;
    return HS_ERR_DEVICE;
  }
  if (stream) {
    p->stream = static_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) {
      delete p;
      return HS_ERR_DEVICE;
    }
    p->own_stream = true;
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&p->h_state), sizeof(DevState), hipHostMallocDefault) != hipSuccess) {
    delete p;
    return HS_ERR_DEVICE;
  }
  warm_kernels(device);
  if (set_func_attributes(p) != HS_OK || ensure_side_stream(p) != HS_OK) {
    const std::string e = p->err;
    std::fprintf(stderr, "hyperslam_hip: %s\n", e.c_str());
    delete p;
    return HS_ERR_DEVICE;
  }
  *out = p;
  return HS_OK;
}

int hs_destroy(hs_problem* p) {
  if (!p) return HS_OK;
  if (p->host_timing && p->host_calls) {
    std::fprintf(stderr, "hs host timing over %d hs_solve calls [ms per call]: structure + tables %.4f, uploads %.4f, launches %.4f, wait %.4f\n", p->host_calls,
                 p->host_ms[0] / p->host_calls, p->host_ms[1] / p->host_calls, p->host_ms[2] / p->host_calls, p->host_ms[3] / p->host_calls);
    const size_t n = p->host_log.size() / 4, h = n / 2;  // second half of the calls: buffers have reached their size, nothing is allocated
    double m[4] = {0, 0, 0, 0};
    for (size_t i = h; i < n; ++i)
      for (int c = 0; c < 4; ++c) m[c] += p->host_log[4 * i + c] / double(n - h);
    std::fprintf(stderr, "hs host timing, second half of the calls: structure + tables %.4f, uploads %.4f, launches %.4f, wait %.4f\n", m[0], m[1], m[2], m[3]);
    if (const char* e = std::getenv("HS_HOST_TIMING"); e && std::atoi(e) >= 2)  // HS_HOST_TIMING=2: every call (where a one-off cost sits)
      for (size_t i = 0; i < n; ++i)
        std::fprintf(stderr, "hs host timing call %zu: structure + tables %.4f, uploads %.4f, launches %.4f, wait %.4f\n", i, p->host_log[4 * i],
                     p->host_log[4 * i + 1], p->host_log[4 * i + 2], p->host_log[4 * i + 3]);
  }
  if (p->scratch) hs_destroy(p->scratch), p->scratch = nullptr;
  (void)hipSetDevice(p->device);
  (void)hipStreamSynchronize(p->stream);
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_join) (void)hipEventDestroy(p->ev_join);
  if (p->ev_irec) (void)hipEventDestroy(p->ev_irec);
  if (p->side) (void)hipStreamDestroy(p->side);
  if (p->h_state) (void)hipHostFree(p->h_state);
  if (p->h_result) (void)hipHostFree(p->h_result);
  if (p->rccl_comm && rccl_api()) (void)rccl_api()->CommDestroy(static_cast<ncclComm_t>(p->rccl_comm));
  if (p->own_stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return HS_OK;
}

const char* hs_last_error(const hs_problem* p) { return p ? p->err.c_str() : "null handle"; }

int hs_set_spline(hs_problem* p, int order, double t0, double dt, int n_cp, const double* cp, const uint8_t* cp_constant, int rc, int tc) {
  if (!p) return HS_ERR_INVALID;
  if (order < 2 || order > hsd::kMaxOrder) HS_FAIL(HS_ERR_INVALID, "spline order out of range");
  if (n_cp < order || !(dt > 0) || !cp) HS_FAIL(HS_ERR_INVALID, "need n_cp >= order, dt > 0 and a control-point table");
  // The basis is uniform: control point j is taken to sit at t0 + j dt, whatever its row says. A table with a hole (upstream prunes state
  // elements one by one, ceres/optimizer.cpp:330-341) or non-uniform knots would silently re-index every later control point: refused.
  // (Stamps accumulated as t += dt differ from t0 + j dt in the last bits only: 1e-9 dt is far above that and far below a knot.)
  for (int j = 0; j < n_cp; ++j)
    if (!(std::fabs(cp[8 * j + 7] - (t0 + j * dt)) <= 1e-9 * dt))
      HS_FAIL(HS_ERR_INVALID, "control-point stamps are not t0 + j dt (row " + std::to_string(j) + "): the spline basis is uniform, a table with a hole or non-uniform knots is refused");
  p->k = order, p->t0 = t0, p->dt = dt, p->n_cp = n_cp;
  p->cp.assign(cp, cp + size_t(8) * n_cp);
  p->cp_const.assign(n_cp, 0);
  if (cp_constant) p->cp_const.assign(cp_constant, cp_constant + n_cp);
  p->rot_const = rc != 0, p->trans_const = tc != 0;
  p->dirty = true;
  return HS_OK;
}

int hs_set_cameras(hs_problem* p, int n, const double* T_bs, const double* intr, const double* dist) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || n > 0xffff || (n && (!T_bs || !intr || !dist))) HS_FAIL(HS_ERR_INVALID, "bad camera table");
  p->n_cam = n;
  p->cam.assign(size_t(kCamStride) * n, 0.0);
  for (int i = 0; i < n; ++i) {
    double* c = &p->cam[size_t(kCamStride) * i];
    std::memcpy(c, T_bs + 7 * i, 56), std::memcpy(c + 7, intr + 4 * i, 32), std::memcpy(c + 11, dist + 4 * i, 32);
  }
  p->dirty = true;
  return HS_OK;
}

int hs_set_sensors(hs_problem* p, int n, const double* T_bs) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && !T_bs)) HS_FAIL(HS_ERR_INVALID, "bad sensor table");
  p->n_sensor = n;
  p->sensor.assign(size_t(8) * n, 0.0);
  for (int i = 0; i < n; ++i) std::memcpy(&p->sensor[size_t(8) * i], T_bs + 7 * i, 56);
  p->dirty = true;
  return HS_OK;
}

int hs_set_landmarks(hs_problem* p, int n, const double* xyz, const uint8_t* constant) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && !xyz)) HS_FAIL(HS_ERR_INVALID, "bad landmark table");
  p->n_lm = n;
  p->lm.assign(xyz, xyz + size_t(3) * n);
  p->lm_const.assign(n, 0);
  if (constant) p->lm_const.assign(constant, constant + n);
  p->dirty = true;
  return HS_OK;
}

int hs_set_imu(hs_problem* p, const double* T_bs, const double* i_g, const double* i_a, const double* S_g, const double* X_a, int bias_order,
               double bias_t0, double bias_dt, int n_bias, const double* bias_g, const double* bias_a, int bias_constant) {
  if (!p) return HS_ERR_INVALID;
  if (!T_bs || !i_g || !i_a || !S_g || !X_a || !bias_g || !bias_a) HS_FAIL(HS_ERR_INVALID, "null IMU table");
  if (bias_order < 2 || bias_order > hsd::kMaxOrder || n_bias < bias_order || !(bias_dt > 0)) HS_FAIL(HS_ERR_INVALID, "bad bias spline");
  p->has_imu = true;
  std::memcpy(p->imu_T_bs, T_bs, 56), std::memcpy(p->imu_i_g, i_g, 48), std::memcpy(p->imu_i_a, i_a, 48);
  std::memcpy(p->imu_S_g, S_g, 72), std::memcpy(p->imu_X_a, X_a, 72);
  p->kb = bias_order, p->bias_t0 = bias_t0, p->bias_dt = bias_dt, p->n_bias = n_bias;
  p->bias_g.assign(bias_g, bias_g + size_t(4) * n_bias), p->bias_a.assign(bias_a, bias_a + size_t(4) * n_bias);
  p->bias_const = bias_constant != 0;
  p->dirty = true;
  return HS_OK;
}

int hs_set_inertial_jacobian(hs_problem* p, int mode) {
  if (!p) return HS_ERR_INVALID;
  if (mode != HS_INERTIAL_AS_REFERENCE && mode != HS_INERTIAL_EXACT) HS_FAIL(HS_ERR_INVALID, "unknown inertial Jacobian mode");
  if (mode != p->inertial_mode) p->inertial_mode = mode, p->dirty = true;
  return HS_OK;
}

int hs_set_gravity(hs_problem* p, const double* g, int constant) {
  if (!p || !g) return HS_ERR_INVALID;
  std::memcpy(p->gravity, g, 24);
  p->gravity_const = constant != 0;
  p->dirty = true;
  return HS_OK;
}

int hs_set_pixel_residuals(hs_problem* p, int n, const double* st, const double* px, const int32_t* lm, const int32_t* cam) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !px || !lm || !cam))) HS_FAIL(HS_ERR_INVALID, "bad pixel residual table");
  p->px_stamp.assign(st, st + n), p->px_meas.assign(px, px + size_t(2) * n), p->px_lm.assign(lm, lm + n), p->px_cam.assign(cam, cam + n);
  p->dirty = true;
  return HS_OK;
}
int hs_set_bearing_residuals(hs_problem* p, int n, const double* st, const double* b, const int32_t* lm, const int32_t* cam) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !b || !lm || !cam))) HS_FAIL(HS_ERR_INVALID, "bad bearing residual table");
  p->br_stamp.assign(st, st + n), p->br_meas.assign(b, b + size_t(3) * n), p->br_lm.assign(lm, lm + n), p->br_cam.assign(cam, cam + n);
  p->dirty = true;
  return HS_OK;
}
int hs_set_prior_residuals(hs_problem* p, int n, const double* st, const double* poses, const int32_t* sensor) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !poses || !sensor))) HS_FAIL(HS_ERR_INVALID, "bad prior residual table");
  p->pr_stamp.assign(st, st + n), p->pr_meas.assign(poses, poses + size_t(7) * n), p->pr_sensor.assign(sensor, sensor + n);
  p->dirty = true;
  return HS_OK;
}
int hs_set_inertial_residuals(hs_problem* p, int n, const double* st, const double* m) {
  if (!p) return HS_ERR_INVALID;
  if (n < 0 || (n && (!st || !m))) HS_FAIL(HS_ERR_INVALID, "bad inertial residual table");
  p->in_stamp.assign(st, st + n), p->in_meas.assign(m, m + size_t(6) * n);
  p->dirty = true;
  return HS_OK;
}

int hs_num_residuals(hs_problem* p, int type) {
  if (!p) return -1;
  switch (type) {
    case HS_PIXEL: return int(p->px_stamp.size());
    case HS_BEARING: return int(p->br_stamp.size());
    case HS_PRIOR: return int(p->pr_stamp.size());
    case HS_INERTIAL: return int(p->in_stamp.size());
  }
  return -1;
}
int hs_dim_pose(hs_problem* p) { return p ? 6 * p->n_cp + (p->has_imu ? 6 * p->n_bias + 2 : 0) : -1; }

int hs_residual_layout(hs_problem* p, int type, int idx, int32_t* num_blocks, int32_t* indices, int32_t* sizes, int32_t* offsets, int32_t* block_ids,
                       int32_t* num_parameters, int32_t* num_residuals) {
  if (!p) return HS_ERR_INVALID;
  if (type < 0 || type > 3 || idx < 0 || idx >= hs_num_residuals(p, type)) HS_FAIL(HS_ERR_INVALID, "residual index out of range");
  if (p->n_cp == 0) HS_FAIL(HS_ERR_STATE, "hs_set_spline has not been called");
  const BlockLayout L = make_block_layout(type, p->k, p->kb);
  *num_blocks = L.num_blocks, *num_parameters = L.num_parameters, *num_residuals = L.num_residuals;
  for (int i = 0; i < 4; ++i) indices[i] = L.indices[i];
  for (int i = 0; i < L.num_blocks; ++i) sizes[i] = L.sizes[i], offsets[i] = L.offsets[i];
  double st = 0;
  switch (type) {
    case HS_PIXEL: st = p->px_stamp[idx]; break;
    case HS_BEARING: st = p->br_stamp[idx]; break;
    case HS_PRIOR: st = p->pr_stamp[idx]; break;
    case HS_INERTIAL: st = p->in_stamp[idx]; break;
  }
  const int first = h_segment_first(st, p->t0, p->dt, p->k);
  int b = 0;
  for (int j = 0; j < p->k; ++j) block_ids[b++] = first + j;
  if (type == HS_PIXEL || type == HS_BEARING) {
    const int cam = type == HS_PIXEL ? p->px_cam[idx] : p->br_cam[idx];
    block_ids[b++] = cam, block_ids[b++] = cam, block_ids[b++] = cam;
    block_ids[b++] = type == HS_PIXEL ? p->px_lm[idx] : p->br_lm[idx];
  } else if (type == HS_PRIOR) {
    block_ids[b++] = p->pr_sensor[idx];
  } else {
    for (int j = 0; j < 5; ++j) block_ids[b++] = 0;
    const int fb = h_segment_first(st, p->bias_t0, p->bias_dt, p->kb);
    for (int j = 0; j < p->kb; ++j) block_ids[b++] = fb + j;
    for (int j = 0; j < p->kb; ++j) block_ids[b++] = fb + j;
    block_ids[b++] = 0;
  }
  return HS_OK;
}

/// Optional sensor-block outputs of hs_linearize (kernels_sensor.hpp): a pass of its own, the solver's kernels do not carry these columns.
static int linearize_sensor_blocks(hs_problem* p, int type, int robustify, const hs_linearization* out) {
  if (!out->J_extrinsics && !out->J_intrinsics && !out->J_distortion && !out->J_gyro_intrinsics && !out->J_acc_intrinsics && !out->J_gyro_sensitivity &&
      !out->J_acc_offsets)
    return HS_OK;
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  const int k = p->k;
  const bool visual = type == HS_PIXEL || type == HS_BEARING;
  const int n = visual ? T.n_vis : (type == HS_PRIOR ? T.n_pri : T.n_ine);
  const int REC = visual ? kSensorRecVisual : (type == HS_PRIOR ? kSensorRecPrior : kSensorRecInertial);
  if (n == 0) return HS_OK;
  HIP_TRY(p->d_dbg.reserve(size_t(n) * REC));
  const int nb = (n + kBlock - 1) / kBlock;
  if (visual) {
    if (k == 4)
      k_sensor_visual<4><<<nb, kBlock, 0, s>>>(T, p->d_dbg.p, p->d_v_dbgpos.p, robustify);
    else
      k_sensor_visual<6><<<nb, kBlock, 0, s>>>(T, p->d_dbg.p, p->d_v_dbgpos.p, robustify);
  } else if (type == HS_PRIOR) {
    if (k == 4)
      k_sensor_prior<4><<<nb, kBlock, 0, s>>>(T, p->d_dbg.p);
    else
      k_sensor_prior<6><<<nb, kBlock, 0, s>>>(T, p->d_dbg.p);
  } else {
    if (k == 4)
      k_sensor_inertial<4, 4><<<nb, kBlock, 0, s>>>(T, p->d_dbg.p, robustify);
    else
      k_sensor_inertial<6, 4><<<nb, kBlock, 0, s>>>(T, p->d_dbg.p, robustify);
  }
  HIP_TRY(hipGetLastError());
  std::vector<double> rec(size_t(n) * REC);
  HIP_TRY(hipMemcpyAsync(rec.data(), p->d_dbg.p, rec.size() * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (visual) {
    const int n_px = int(p->px_stamp.size()), n_br = int(p->br_stamp.size());
    const int base = type == HS_PIXEL ? 0 : n_px, cnt = type == HS_PIXEL ? n_px : n_br, nres = type == HS_PIXEL ? 2 : 1;
    for (int i = 0; i < cnt; ++i) {
      const double* r = &rec[size_t(base + i) * REC];
      if (out->J_extrinsics) std::memcpy(out->J_extrinsics + size_t(i) * nres * 6, r, sizeof(double) * nres * 6);
      if (type == HS_PIXEL) {
        if (out->J_intrinsics) std::memcpy(out->J_intrinsics + size_t(i) * 8, r + 12, 64);
        if (out->J_distortion) std::memcpy(out->J_distortion + size_t(i) * 8, r + 20, 64);
      }
    }
  } else if (type == HS_PRIOR) {
    for (int d = 0; d < n; ++d)
      if (out->J_extrinsics) std::memcpy(out->J_extrinsics + size_t(p->pr_order[d]) * 36, &rec[size_t(d) * REC], 36 * 8);
  } else {
    for (int d = 0; d < n; ++d) {
      const size_t i = size_t(p->in_order[d]);
      const double* r = &rec[size_t(d) * REC];
      if (out->J_extrinsics) std::memcpy(out->J_extrinsics + i * 36, r, 36 * 8);
      if (out->J_gyro_intrinsics) std::memcpy(out->J_gyro_intrinsics + i * 36, r + 36, 36 * 8);
      if (out->J_acc_intrinsics) std::memcpy(out->J_acc_intrinsics + i * 36, r + 72, 36 * 8);
      if (out->J_gyro_sensitivity) std::memcpy(out->J_gyro_sensitivity + i * 54, r + 108, 54 * 8);
      if (out->J_acc_offsets) std::memcpy(out->J_acc_offsets + i * 54, r + 162, 54 * 8);
    }
  }
  return HS_OK;
}

static int linearize_impl(hs_problem* p, int type, int robustify, const hs_linearization* out);

/// hs_linearize with CostConfiguration::weights (exteroceptive.cpp:129-147): output = W * distance, J_w = W * J_m * J_e, then Ceres' loss
/// corrector on the weighted residual. The device produces the unweighted, uncorrected rows (linearize_impl, robustify = 0); W (n_res x
/// n_res) and the corrector are applied per residual block while the rows are handed out.
int hs_linearize(hs_problem* p, int type, int robustify, const hs_linearization* out) {
  if (!p || !out) return HS_ERR_INVALID;
  if (type < HS_PIXEL || type > HS_INERTIAL) HS_FAIL(HS_ERR_INVALID, "unknown factor type");
  const std::vector<double>& W = p->weights[type];
  if (W.empty()) return linearize_impl(p, type, robustify, out);
  const int n = hs_num_residuals(p, type), nr = type == HS_PIXEL ? 2 : (type == HS_BEARING ? 1 : 6);
  hs_linearization o = *out;
  std::vector<double> r_tmp, c_tmp;
  if (!o.r) r_tmp.resize(size_t(n) * nr), o.r = r_tmp.data();
  if (!o.cost) c_tmp.resize(n), o.cost = c_tmp.data();
  const int rc = linearize_impl(p, type, 0, &o);
  if (rc) return rc;
  struct Block {
    double* J;
    int cols;
  };
  const Block blocks[] = {{o.J_state, 6 * p->k},       {o.J_landmark, 3},        {o.J_bias_g, 3 * p->kb},      {o.J_bias_a, 3 * p->kb},
                          {o.J_gravity, 2},            {o.J_extrinsics, 6},      {o.J_intrinsics, 4},          {o.J_distortion, 4},
                          {o.J_gyro_intrinsics, 6},    {o.J_acc_intrinsics, 6},  {o.J_gyro_sensitivity, 9},    {o.J_acc_offsets, 9}};
  const bool visual = type == HS_PIXEL || type == HS_BEARING;
  std::vector<double> tmp;
  for (int i = 0; i < n; ++i) {
    double rw[6], s2 = 0.0;
    for (int a = 0; a < nr; ++a) {
      rw[a] = 0.0;
      for (int c = 0; c < nr; ++c) rw[a] += W[a * nr + c] * o.r[size_t(i) * nr + c];
      s2 += rw[a] * rw[a];
    }
    double rho = s2, sr = 1.0;  // loss of the factor type on the weighted residual (optimizer.cpp:204,226,250,267)
    if (visual) {
      const double a = type == HS_PIXEL ? kHuberPixel : kHuberBearing;
      if (s2 > a * a) rho = 2.0 * a * std::sqrt(s2) - a * a, sr = std::sqrt(a / std::sqrt(s2));
    } else if (type == HS_INERTIAL) {
      rho = kScaleInertial * s2, sr = std::sqrt(kScaleInertial);
    }
    if (!robustify) sr = 1.0;
    o.cost[i] = 0.5 * rho;
    for (int a = 0; a < nr; ++a) o.r[size_t(i) * nr + a] = sr * rw[a];
    for (const Block& b : blocks) {
      if (!b.J) continue;
      if ((b.J == o.J_landmark || b.J == o.J_intrinsics || b.J == o.J_distortion) && !visual) continue;
      if ((b.J == o.J_intrinsics || b.J == o.J_distortion) && type != HS_PIXEL) continue;
      if ((b.J == o.J_bias_g || b.J == o.J_bias_a || b.J == o.J_gravity || b.J == o.J_gyro_intrinsics || b.J == o.J_acc_intrinsics ||
           b.J == o.J_gyro_sensitivity || b.J == o.J_acc_offsets) && type != HS_INERTIAL)
        continue;
      double* J = b.J + size_t(i) * nr * b.cols;
      tmp.assign(J, J + size_t(nr) * b.cols);
      for (int a = 0; a < nr; ++a)
        for (int c = 0; c < b.cols; ++c) {
          double v = 0.0;
          for (int m = 0; m < nr; ++m) v += W[a * nr + m] * tmp[size_t(m) * b.cols + c];
          J[size_t(a) * b.cols + c] = sr * v;
        }
    }
  }
  return HS_OK;
}

static int linearize_impl(hs_problem* p, int type, int robustify, const hs_linearization* out) {
  if (!p || !out) return HS_ERR_INVALID;
  int rc = prepare(p);
  if (rc) return rc;
  rc = reset_state(p, 0, 1e4);
  if (rc) return rc;
  if (type < HS_PIXEL || type > HS_INERTIAL) HS_FAIL(HS_ERR_INVALID, "unknown factor type");
  rc = linearize_sensor_blocks(p, type, robustify, out);
  if (rc) return rc;
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  const int k = p->k;
  if (type == HS_PIXEL || type == HS_BEARING) {
    const int n_px = int(p->px_stamp.size()), n_br = int(p->br_stamp.size());
    const int n = T.n_vis, REC = 8 + 12 * k;
    if (n == 0) return HS_OK;
    HIP_TRY(p->d_dbg.reserve(size_t(n) * REC));
    HIP_TRY(p->d_dbg_cost.reserve(n));
    if (k == 4)
      k_linearize_visual<4><<<(n + lin_block<4>() - 1) / lin_block<4>(), lin_block<4>(), lin_lds_bytes<4>(p), s>>>(T, p->d_dbg.p, p->d_v_dbgpos.p, robustify, nullptr, p->d_dbg_cost.p);
    else
      k_linearize_visual<6><<<(n + lin_block<6>() - 1) / lin_block<6>(), lin_block<6>(), lin_lds_bytes<6>(p), s>>>(T, p->d_dbg.p, p->d_v_dbgpos.p, robustify, nullptr, p->d_dbg_cost.p);
    HIP_TRY(hipGetLastError());
    std::vector<double> rec(size_t(n) * REC), cost(n);
    HIP_TRY(hipMemcpyAsync(rec.data(), p->d_dbg.p, rec.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(cost.data(), p->d_dbg_cost.p, cost.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int base = type == HS_PIXEL ? 0 : n_px, cnt = type == HS_PIXEL ? n_px : n_br, nres = type == HS_PIXEL ? 2 : 1;
    for (int i = 0; i < cnt; ++i) {
      const double* r = &rec[size_t(base + i) * REC];
      for (int rr = 0; rr < nres; ++rr) {
        if (out->r) out->r[size_t(i) * nres + rr] = r[rr];
        if (out->J_landmark)
          for (int c = 0; c < 3; ++c) out->J_landmark[(size_t(i) * nres + rr) * 3 + c] = r[2 + 3 * rr + c];
        if (out->J_state)
          for (int c = 0; c < 6 * k; ++c) out->J_state[(size_t(i) * nres + rr) * 6 * k + c] = r[8 + rr * 6 * k + c];
      }
      if (out->cost) out->cost[i] = cost[base + i];
      if (out->first_cp) out->first_cp[i] = h_segment_first(type == HS_PIXEL ? p->px_stamp[i] : p->br_stamp[i], p->t0, p->dt, k);
    }
    return HS_OK;
  }
  if (type == HS_PRIOR) {
    const int n = T.n_pri, REC = 6 + 36 * k;
    if (n == 0) return HS_OK;
    HIP_TRY(p->d_dbg.reserve(size_t(n) * REC));
    HIP_TRY(p->d_dbg_cost.reserve(n));
    if (k == 4)
      k_linearize_prior<4><<<p->nb_pri, kBlock, cp_lds_bytes(p), s>>>(T, p->d_dbg.p, nullptr, p->d_dbg_cost.p);
    else
      k_linearize_prior<6><<<p->nb_pri, kBlock, cp_lds_bytes(p), s>>>(T, p->d_dbg.p, nullptr, p->d_dbg_cost.p);
    HIP_TRY(hipGetLastError());
    std::vector<double> rec(size_t(n) * REC), cost(n);
    HIP_TRY(hipMemcpyAsync(rec.data(), p->d_dbg.p, rec.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(cost.data(), p->d_dbg_cost.p, cost.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int d = 0; d < n; ++d) {
      const int i = p->pr_order[d];
      const double* r = &rec[size_t(d) * REC];
      if (out->r) std::memcpy(out->r + size_t(i) * 6, r, 48);
      if (out->J_state) std::memcpy(out->J_state + size_t(i) * 36 * k, r + 6, sizeof(double) * 36 * k);
      if (out->cost) out->cost[i] = cost[d];
      if (out->first_cp) out->first_cp[i] = p->pr_first[d];
    }
    return HS_OK;
  }
  {
    const int n = T.n_ine, kb = p->kb, REC = 18 + 36 * k + 2 * kb;
    if (n == 0) return HS_OK;
    HIP_TRY(p->d_dbg.reserve(size_t(n) * REC));
    HIP_TRY(p->d_dbg_cost.reserve(n));
    if (k == 4)
      k_linearize_inertial<4, 4><<<p->nb_ine, kInertialBlock * 4, cp_lds_bytes(p), s>>>(T, p->d_dbg.p, robustify, nullptr, p->d_dbg_cost.p);
    else
      k_linearize_inertial<6, 4><<<p->nb_ine, kInertialBlock * 6, cp_lds_bytes(p), s>>>(T, p->d_dbg.p, robustify, nullptr, p->d_dbg_cost.p);
    HIP_TRY(hipGetLastError());
    std::vector<double> rec(size_t(n) * REC), cost(n);
    HIP_TRY(hipMemcpyAsync(rec.data(), p->d_dbg.p, rec.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(cost.data(), p->d_dbg_cost.p, cost.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int d = 0; d < n; ++d) {
      const int i = p->in_order[d];
      const double* r = &rec[size_t(d) * REC];
      if (out->r) std::memcpy(out->r + size_t(i) * 6, r, 48);
      if (out->J_state) std::memcpy(out->J_state + size_t(i) * 36 * k, r + 6, sizeof(double) * 36 * k);
      const double* wg = r + 6 + 36 * k;
      const double* wa = wg + kb;
      const double* jg = wa + kb;
      if (out->J_bias_g || out->J_bias_a)
        for (int row = 0; row < 6; ++row)
          for (int j = 0; j < kb; ++j)
            for (int c = 0; c < 3; ++c) {
              if (out->J_bias_g) out->J_bias_g[(size_t(i) * 6 + row) * 3 * kb + 3 * j + c] = (row == c) ? wg[j] : 0.0;
              if (out->J_bias_a) out->J_bias_a[(size_t(i) * 6 + row) * 3 * kb + 3 * j + c] = (row == 3 + c) ? wa[j] : 0.0;
            }
      if (out->J_gravity) std::memcpy(out->J_gravity + size_t(i) * 12, jg, 96);
      if (out->cost) out->cost[i] = cost[d];
      if (out->first_cp) out->first_cp[i] = p->in_first[d];
      if (out->first_bias) out->first_bias[i] = p->in_first_bias[d];
    }
    return HS_OK;
  }
}

int hs_cost(hs_problem* p, double* cost) {
  if (!p || !cost) return HS_ERR_INVALID;
  if (p->has_weights()) HS_FAIL(HS_ERR_INVALID, kWeightsMessage);
  int rc = prepare(p);
  if (rc) return rc;
  rc = reset_state(p, 0, 1e4);
  if (rc) return rc;
  rc = p->k == 4 ? launch_linearize<4>(p, false, true) : launch_linearize<6>(p, false, true);
  if (rc) return rc;
  k_pack_exchange<<<1, kBlock, 0, p->stream>>>(p->T, 0);
  rc = exchange(p, p->T.xbuf + p->T.xo_cost, 1);
  if (rc) return rc;
  k_cost_reduce<<<1, kBlock, 0, p->stream>>>(p->T);
  HIP_TRY(hipGetLastError());
  DevState st;
  HIP_TRY(hipMemcpyAsync(&st, p->d_state.p, sizeof(st), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  *cost = st.cost;
  return HS_OK;
}

int hs_reduced_system(hs_problem* p, double radius, double* S, double* g) {
  if (!p || !S || !g) return HS_ERR_INVALID;
  if (p->has_weights()) HS_FAIL(HS_ERR_INVALID, kWeightsMessage);
  int rc = prepare(p);
  if (rc) return rc;
  rc = reset_state(p, 1, radius);
  if (rc) return rc;
  rc = p->k == 4 ? launch_linearize<4>(p) : launch_linearize<6>(p);
  if (rc) return rc;
  rc = p->k == 4 ? launch_build<4>(p) : launch_build<6>(p);
  if (rc) return rc;
  const int np = p->T.np, ncb = 6 * p->T.bw;
  std::vector<double> Sb(size_t(np) * ncb), gs(np);
  HIP_TRY(hipMemcpyAsync(Sb.data(), p->d_Sb.p, Sb.size() * 8, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipMemcpyAsync(gs.data(), p->d_g_s.p, gs.size() * 8, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  const int dim = hs_dim_pose(p);
  std::memset(S, 0, sizeof(double) * size_t(dim) * dim);
  for (int rho = 0; rho < np; ++rho) {
    const int c0 = 6 * (rho / 6);
    for (int c = 0; c < ncb && c0 + c < np; ++c) {
      const double v = Sb[size_t(rho) * ncb + c];
      if (c0 + c >= rho) S[size_t(rho) * dim + c0 + c] = v, S[size_t(c0 + c) * dim + rho] = v;
    }
    g[rho] = gs[rho];
  }
  if (p->T.nb) {
    const int nb = p->T.nb;
    std::vector<double> Spb(size_t(np) * nb), Sbb(size_t(nb) * nb), gb(nb);
    HIP_TRY(hipMemcpyAsync(Spb.data(), p->d_Spb.p, Spb.size() * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(Sbb.data(), p->d_Sbb.p, Sbb.size() * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(gb.data(), p->d_gb_s.p, gb.size() * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int rho = 0; rho < np; ++rho)
      for (int b = 0; b < nb; ++b) S[size_t(rho) * dim + np + b] = S[size_t(np + b) * dim + rho] = Spb[size_t(rho) * nb + b];
    for (int a = 0; a < nb; ++a) {
      for (int b = 0; b < nb; ++b) S[size_t(np + a) * dim + np + b] = Sbb[size_t(a) * nb + b];
      g[np + a] = gb[a];
    }
  }
  return HS_OK;
}

int hs_set_weights(hs_problem* p, int type, const double* weights) {
  if (!p) return HS_ERR_INVALID;
  if (type < HS_PIXEL || type > HS_INERTIAL) HS_FAIL(HS_ERR_INVALID, "unknown factor type");
  const int nr = type == HS_PIXEL ? 2 : (type == HS_BEARING ? 1 : 6);
  if (weights)
    p->weights[type].assign(weights, weights + nr * nr);
  else
    p->weights[type].clear();
  return HS_OK;
}

int hs_set_stage_timing(hs_problem* p, int enabled) {
  if (!p) return HS_ERR_INVALID;
  p->stage_timing = enabled != 0;
  return HS_OK;
}

int hs_solve(hs_problem* p, int max_iterations, hs_summary* summary, hs_iteration* iterations) {
  if (!p || !summary) return HS_ERR_INVALID;
  if (p->has_weights()) HS_FAIL(HS_ERR_INVALID, kWeightsMessage);
  if (max_iterations < 0 || max_iterations > kMaxIterations) HS_FAIL(HS_ERR_INVALID, "max_iterations out of range");
  p->results_cached = false;
  int rc = prepare(p);
  if (rc) return rc;
  const bool spec = speculative_solve(p);
  const bool deferred = (spec || fused_visual_only(p)) && !commit_inline(p) && !(p->T.debug_flags & 67108864);  // A/B switch 67108864: k_commit in every iteration
  rc = reset_state(p, max_iterations, 1e4, spec ? (deferred ? 2 : 1) : (deferred ? 4 : 0));
  if (rc) return rc;
  hipStream_t s = p->stream;
  const auto host_t3 = std::chrono::steady_clock::now();
  // stage timing (optional, hs_set_stage_timing): 4 stages per iteration bracketed by HIP events on the launch stream; every event is a
  // barrier packet (~5.7 us of idle device each, rocprofv3 kernel trace), so by default only the two ends of the solve are stamped
  const bool stages = p->stage_timing;
  const size_t n_ev = size_t(6) * max_iterations + 1;  // (+ two per iteration around the candidate's linearisation of a speculative solve)
  const size_t ev_cand = size_t(4) * max_iterations + 1;
  while (p->events.size() < n_ev) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    p->events.push_back(e);
  }
  std::vector<hipEvent_t>& ev = p->events;
  HIP_TRY(hipEventRecord(ev[0], s));
  for (int it = 0; it < max_iterations; ++it) {
    if (it == 0 || !spec) {  // (speculative solves: the linearisation of the current point came with the previous iteration's candidate)
      rc = p->k == 4 ? launch_linearize<4>(p, true) : launch_linearize<6>(p, true);
      if (rc) return rc;
    }
    if (stages && !p->fused) HIP_TRY(hipEventRecord(ev[4 * it + 1], s));
    hipEvent_t after_build = stages && p->fused ? ev[4 * it + 1] : nullptr;  // fused build: the linearise stage ends behind k_build_visual
    rc = p->k == 4 ? launch_build<4>(p, after_build) : launch_build<6>(p, after_build);
    if (rc) return rc;
    if (stages) HIP_TRY(hipEventRecord(ev[4 * it + 2], s));
    rc = launch_factor(p);
    if (rc) return rc;
    if (stages) HIP_TRY(hipEventRecord(ev[4 * it + 3], s));
    const bool lin_cand = spec && it + 1 < max_iterations;
    hipEvent_t* lin_ev = stages && lin_cand ? &ev[ev_cand + 2 * it] : nullptr;
    rc = p->k == 4 ? launch_update<4>(p, lin_cand, deferred, lin_ev) : launch_update<6>(p, lin_cand, deferred, lin_ev);
    if (rc) return rc;
    if (deferred && it + 1 == max_iterations) launch_commit(p);  // the last accepted candidate (also when a convergence test ended the solve early)
    if (stages || it + 1 == max_iterations) HIP_TRY(hipEventRecord(ev[4 * it + 4], s));
  }
  if (max_iterations == 0) {
    rc = p->k == 4 ? launch_linearize<4>(p, false, true) : launch_linearize<6>(p, false, true);
    if (rc) return rc;
    k_pack_exchange<<<1, kBlock, 0, s>>>(p->T, 0);
    rc = exchange(p, p->T.xbuf + p->T.xo_cost, 1);
    if (rc) return rc;
    k_cost_reduce<<<1, kBlock, 0, s>>>(p->T);
  }
  DevState& st = *p->h_state;
  const auto host_t4 = std::chrono::steady_clock::now();
  if (p->want_results) {  // [cp | lm (device order) | bias_g | bias_a | gravity] -> pinned host memory, behind the last kernel
    const size_t n_cp8 = p->cp.size(), n_lm3 = p->lm.size(), n_b = p->has_imu ? p->bias_g.size() : 0, total = n_cp8 + n_lm3 + 2 * n_b + 3;
    if (total > p->h_result_cap) {
      if (p->h_result) (void)hipHostFree(p->h_result);
      p->h_result = nullptr, p->h_result_cap = 0;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->h_result), 2 * total * sizeof(double), hipHostMallocDefault));
      p->h_result_cap = 2 * total;
    }
    double* h = p->h_result;
    HIP_TRY(hipMemcpyAsync(h, p->d_cp.p, n_cp8 * 8, hipMemcpyDeviceToHost, s));
    if (n_lm3) HIP_TRY(hipMemcpyAsync(h + n_cp8, p->d_lm.p, n_lm3 * 8, hipMemcpyDeviceToHost, s));
    if (n_b) {
      HIP_TRY(hipMemcpyAsync(h + n_cp8 + n_lm3, p->d_bias_g.p, n_b * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(h + n_cp8 + n_lm3 + n_b, p->d_bias_a.p, n_b * 8, hipMemcpyDeviceToHost, s));
    }
    if (p->has_imu) HIP_TRY(hipMemcpyAsync(h + n_cp8 + n_lm3 + 2 * n_b, p->d_gravity.p, 24, hipMemcpyDeviceToHost, s));  // also with an empty bias table
  }
  HIP_TRY(hipMemcpyAsync(&st, p->d_state.p, sizeof(st), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  p->results_cached = p->want_results;  // only once every copy above has completed: a failed copy or synchronisation leaves the getters on the device path
  if (p->host_timing) {
    const auto host_t5 = std::chrono::steady_clock::now();
    const double tl = std::chrono::duration<double, std::milli>(host_t4 - host_t3).count(), tw = std::chrono::duration<double, std::milli>(host_t5 - host_t4).count();
    p->host_ms[2] += tl, p->host_ms[3] += tw;
    p->host_log.insert(p->host_log.end(), {p->host_prepare[0], p->host_prepare[1], tl, tw});
    p->host_prepare[0] = p->host_prepare[1] = 0;
    ++p->host_calls;
  }
  std::memset(summary, 0, sizeof(*summary));
  summary->initial_cost = st.records[0].cost;
  summary->final_cost = st.cost;
  summary->num_iterations = st.num_iterations;
  summary->num_successful_steps = st.num_successful;
  summary->termination = st.termination;
  summary->num_residual_blocks = p->T.n_vis + p->T.n_pri + p->T.n_ine;
  if (!stages) summary->linearize_ms = summary->schur_ms = summary->solve_ms = summary->update_ms = -1.0;  // not measured (hs_set_stage_timing)
  for (int it = 0; stages && it < max_iterations; ++it) {
    float t[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) (void)hipEventElapsedTime(&t[k], ev[4 * it + k], ev[4 * it + k + 1]);
    summary->linearize_ms += t[0], summary->schur_ms += t[1], summary->solve_ms += t[2], summary->update_ms += t[3];
    if (spec && it + 1 < max_iterations) {  // the linearisation of this iteration's candidate = the next iteration's linearisation when accepted
      float tc = 0;
      (void)hipEventElapsedTime(&tc, ev[ev_cand + 2 * it], ev[ev_cand + 2 * it + 1]);
      summary->linearize_ms += tc, summary->update_ms -= tc;
    }
  }
  if (max_iterations > 0) {
    float t = 0;
    (void)hipEventElapsedTime(&t, ev[0], ev[4 * max_iterations]);
    summary->total_ms = t;
  }
  if (iterations) {
    std::memset(iterations, 0, sizeof(hs_iteration) * (size_t(max_iterations) + 1));
    const int n = std::min(st.num_iterations, max_iterations);
    for (int i = 0; i <= n; ++i) iterations[i] = st.records[i];
  }
  if (st.chol_failed && st.termination == HS_FAILURE)
    p->err = st.chol_failed == 2 ? "two-ended solve: the partner workgroup did not arrive within 2 s" : "reduced system not positive definite";
  return HS_OK;
}

#if HS_PROFILE_HOOKS
/// Profiling builds only (hipcc -DHS_PROFILE_HOOKS=1, tools/build_profiling_lib.sh): the phase timestamps the kernels wrote under
/// HS_DEBUG_FLAGS 16 / 32. Not part of the C ABI: the product library neither contains the hooks nor exports this function.
extern "C" int hs_debug_read(hs_problem* p, double* dst, int n) {
  if (!p || !dst) return HS_ERR_INVALID;
  HIP_TRY(hipMemcpyAsync(dst, p->d_xpart.p, size_t(n) * 8, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return HS_OK;
}
#endif

int hs_snapshot(hs_problem* p) {
  if (!p) return HS_ERR_INVALID;
  int rc = prepare(p);
  if (rc) return rc;
  HIP_TRY(p->d_cp_snap.reserve(p->cp.size()));
  HIP_TRY(p->d_lm_snap.reserve(p->lm.size() + 1));
  HIP_TRY(hipMemcpyAsync(p->d_cp_snap.p, p->d_cp.p, p->cp.size() * 8, hipMemcpyDeviceToDevice, p->stream));
  if (p->n_lm) HIP_TRY(hipMemcpyAsync(p->d_lm_snap.p, p->d_lm.p, p->lm.size() * 8, hipMemcpyDeviceToDevice, p->stream));
  if (p->has_imu) {
    HIP_TRY(p->d_bias_g_snap.reserve(p->bias_g.size() + 1));
    HIP_TRY(p->d_bias_a_snap.reserve(p->bias_a.size() + 1));
    HIP_TRY(p->d_gravity_snap.reserve(3));
    HIP_TRY(hipMemcpyAsync(p->d_bias_g_snap.p, p->d_bias_g.p, p->bias_g.size() * 8, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_bias_a_snap.p, p->d_bias_a.p, p->bias_a.size() * 8, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_gravity_snap.p, p->d_gravity.p, 24, hipMemcpyDeviceToDevice, p->stream));
  }
  p->has_snapshot = true;
  return HS_OK;
}

int hs_restore(hs_problem* p) {
  if (!p) return HS_ERR_INVALID;
  if (!p->has_snapshot || p->dirty) HS_FAIL(HS_ERR_STATE, "hs_restore without a valid hs_snapshot");
  p->results_cached = false;
  HIP_TRY(hipMemcpyAsync(p->d_cp.p, p->d_cp_snap.p, p->cp.size() * 8, hipMemcpyDeviceToDevice, p->stream));
  if (p->n_lm) HIP_TRY(hipMemcpyAsync(p->d_lm.p, p->d_lm_snap.p, p->lm.size() * 8, hipMemcpyDeviceToDevice, p->stream));
  if (p->has_imu) {
    HIP_TRY(hipMemcpyAsync(p->d_bias_g.p, p->d_bias_g_snap.p, p->bias_g.size() * 8, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_bias_a.p, p->d_bias_a_snap.p, p->bias_a.size() * 8, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_gravity.p, p->d_gravity_snap.p, 24, hipMemcpyDeviceToDevice, p->stream));
  }
  return HS_OK;
}

int hs_rccl_unique_id(char id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!id) return HS_ERR_INVALID;
  RcclApi* api = rccl_api();
  if (!api) return HS_ERR_DEVICE;
  ncclUniqueId u;
  if (api->GetUniqueId(&u) != ncclSuccess) return HS_ERR_DEVICE;
  std::memcpy(id, &u, 128);
  return HS_OK;
}

int hs_rccl_init(hs_problem* p, const char id[128], int rank, int world) {
  if (!p || !id || world < 1 || rank < 0 || rank >= world) return HS_ERR_INVALID;
  RcclApi* api = rccl_api();
  if (!api) HS_FAIL(HS_ERR_DEVICE, "librccl.so could not be loaded");
  HIP_TRY(hipSetDevice(p->device));
  if (p->rccl_comm) {  // a second initialisation replaces the communicator instead of leaking it
    (void)api->CommDestroy(static_cast<ncclComm_t>(p->rccl_comm));
    p->rccl_comm = nullptr;
  }
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  ncclComm_t comm = nullptr;
  const ncclResult_t r = api->CommInitRank(&comm, world, u, rank);
  if (r != ncclSuccess) HS_FAIL(HS_ERR_DEVICE, std::string("ncclCommInitRank failed: ") + (api->GetErrorString ? api->GetErrorString(r) : "?"));
  p->rccl_comm = comm;
  return HS_OK;
}

int hs_rccl_shutdown(hs_problem* p) {
  if (!p) return HS_ERR_INVALID;
  if (p->rccl_comm && rccl_api()) {
    (void)hipStreamSynchronize(p->stream);
    (void)rccl_api()->CommDestroy(static_cast<ncclComm_t>(p->rccl_comm));
  }
  p->rccl_comm = nullptr;
  return HS_OK;
}

int hs_exchange_info(hs_problem* p, int32_t* rccl_ranks, int64_t* doubles_per_linearisation, int64_t* doubles_per_decision) {
  if (!p) return HS_ERR_INVALID;
  if (rccl_ranks) {
    int n = 0;
    if (p->rccl_comm && rccl_api() && rccl_api()->CommCount) {
      const ncclResult_t r = rccl_api()->CommCount(static_cast<ncclComm_t>(p->rccl_comm), &n);
      if (r != ncclSuccess) HS_FAIL(HS_ERR_DEVICE, "ncclCommCount failed");
    }
    *rccl_ranks = n;
  }
  if (doubles_per_linearisation) *doubles_per_linearisation = p->dirty ? 0 : p->T.x_count1;
  if (doubles_per_decision) *doubles_per_decision = 5;
  return HS_OK;
}

int hs_set_allreduce(hs_problem* p, hs_allreduce_fn fn, void* user) {
  if (!p) return HS_ERR_INVALID;
  p->allreduce = fn, p->allreduce_user = user;
  return HS_OK;
}

int hs_set_shard(hs_problem* p, int rank, int world, int min_band_blocks) {
  if (!p) return HS_ERR_INVALID;
  if (world < 1 || rank < 0 || rank >= world || min_band_blocks < 0) HS_FAIL(HS_ERR_INVALID, "bad shard description");
  p->rank = rank, p->world = world, p->min_bw = min_band_blocks;
  p->dirty = true;
  return HS_OK;
}

int hs_band_blocks(hs_problem* p) {
  if (!p) return -1;
  if (prepare(p) != HS_OK) return -1;
  return p->T.bw;
}

int hs_get_control_points(hs_problem* p, double* cp) {
  if (!p || !cp) return HS_ERR_INVALID;
  if (p->dirty) {
    std::memcpy(cp, p->cp.data(), p->cp.size() * 8);
    return HS_OK;
  }
  if (p->results_cached) {
    std::memcpy(cp, p->h_result, p->cp.size() * 8);
  } else {
    p->want_results = true;
    HIP_TRY(hipMemcpyAsync(cp, p->d_cp.p, size_t(8) * p->n_cp * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  std::memcpy(p->cp.data(), cp, p->cp.size() * 8);
  return HS_OK;
}
int hs_get_landmarks(hs_problem* p, double* xyz) {
  if (!p || !xyz) return HS_ERR_INVALID;
  if (p->dirty || p->n_lm == 0) {
    std::memcpy(xyz, p->lm.data(), p->lm.size() * 8);
    return HS_OK;
  }
  std::vector<double> fetched;
  const double* dev = p->h_result + p->cp.size();
  if (!p->results_cached) {
    p->want_results = true;
    fetched.resize(size_t(3) * p->n_lm);
    HIP_TRY(hipMemcpyAsync(fetched.data(), p->d_lm.p, fetched.size() * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    dev = fetched.data();
  }
  for (int d = 0; d < p->n_lm; ++d) {
    const int t = p->vs.table_of_dev[d];
    for (int c = 0; c < 3; ++c) xyz[3 * t + c] = p->lm[3 * t + c] = dev[3 * d + c];
  }
  return HS_OK;
}
int hs_get_bias(hs_problem* p, double* bg, double* ba) {
  if (!p || !bg || !ba) return HS_ERR_INVALID;
  if (!p->dirty && p->has_imu && !p->bias_g.empty()) {
    if (p->results_cached) {
      const double* h = p->h_result + p->cp.size() + p->lm.size();
      std::memcpy(p->bias_g.data(), h, p->bias_g.size() * 8), std::memcpy(p->bias_a.data(), h + p->bias_g.size(), p->bias_a.size() * 8);
    } else {
      p->want_results = true;
      HIP_TRY(hipMemcpyAsync(p->bias_g.data(), p->d_bias_g.p, p->bias_g.size() * 8, hipMemcpyDeviceToHost, p->stream));
      HIP_TRY(hipMemcpyAsync(p->bias_a.data(), p->d_bias_a.p, p->bias_a.size() * 8, hipMemcpyDeviceToHost, p->stream));
      HIP_TRY(hipStreamSynchronize(p->stream));
    }
  }
  std::memcpy(bg, p->bias_g.data(), p->bias_g.size() * 8), std::memcpy(ba, p->bias_a.data(), p->bias_a.size() * 8);
  return HS_OK;
}
int hs_get_gravity(hs_problem* p, double* g) {
  if (!p || !g) return HS_ERR_INVALID;
  if (!p->dirty && p->has_imu) {
    if (p->results_cached) {
      std::memcpy(p->gravity, p->h_result + p->cp.size() + p->lm.size() + 2 * p->bias_g.size(), 24);
    } else {
      p->want_results = true;
      HIP_TRY(hipMemcpyAsync(p->gravity, p->d_gravity.p, 24, hipMemcpyDeviceToHost, p->stream));
      HIP_TRY(hipStreamSynchronize(p->stream));
    }
  }
  std::memcpy(g, p->gravity, 24);
  return HS_OK;
}

// Left inverse of EigenQuaternionManifold's PlusJacobian (orthonormal columns e_i (x) q): J_ambient = J_local * P^T.
static void quat_plus_jacobian_T(const double* q, double* PT /* 3 x 4 */) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  // column i of P = (e_i, 0) (x) q  (Hamilton, xyzw)
  const double P[4][3] = {{w, z, -y}, {-z, w, x}, {y, -x, w}, {-x, -y, -z}};
  for (int i = 0; i < 3; ++i)
    for (int r = 0; r < 4; ++r) PT[4 * i + r] = P[r][i];
}

int hs_cost_function_evaluate(hs_problem* p, int type, int idx, const double* const* parameters, double* residuals, double** jacobians) {
  if (!p || !parameters || !residuals) return HS_ERR_INVALID;
  if (type < 0 || type > 3 || idx < 0 || idx >= hs_num_residuals(p, type)) HS_FAIL(HS_ERR_INVALID, "residual index out of range");
  const int k = p->k, kb = p->kb;
  const BlockLayout L = make_block_layout(type, k, kb);
  // one-residual window at the given parameter values, on a scratch handle that lives as long as p (same device, same stream)
  if (!p->scratch) {
    const int rc0 = hs_create(p->device, p->stream, &p->scratch);
    if (rc0) HS_FAIL(rc0, "hs_cost_function_evaluate: scratch handle");
  }
  hs_problem* q = p->scratch;
  q->inertial_mode = p->inertial_mode;
  for (int t = 0; t < 4; ++t) q->weights[t] = p->weights[t];
  // tables of the previous call (possibly another factor type)
  q->px_stamp.clear(), q->px_meas.clear(), q->px_lm.clear(), q->px_cam.clear(), q->br_stamp.clear(), q->br_meas.clear(), q->br_lm.clear(), q->br_cam.clear();
  q->pr_stamp.clear(), q->pr_meas.clear(), q->pr_sensor.clear(), q->in_stamp.clear(), q->in_meas.clear();
  q->has_imu = false, q->n_bias = 0, q->bias_g.clear(), q->bias_a.clear(), q->n_lm = 0, q->lm.clear(), q->lm_const.clear();
  q->dirty = true;
  int rc = HS_OK;
  std::vector<double> cps(size_t(8) * k);
  for (int j = 0; j < k; ++j) std::memcpy(&cps[8 * j], parameters[j], 64);
  const double t0 = cps[7], dt = cps[15] - cps[7];
  rc = hs_set_spline(q, k, t0, dt, k, cps.data(), nullptr, 0, 0);
  int32_t zero = 0;
  double stamp = 0;
  int nres = L.num_residuals;
  if (!rc && (type == HS_PIXEL || type == HS_BEARING)) {
    rc = hs_set_cameras(q, 1, parameters[k], parameters[k + 1], parameters[k + 2]);
    if (!rc) rc = hs_set_landmarks(q, 1, parameters[k + 3], nullptr);
    stamp = type == HS_PIXEL ? p->px_stamp[idx] : p->br_stamp[idx];
    if (!rc) rc = type == HS_PIXEL ? hs_set_pixel_residuals(q, 1, &stamp, &p->px_meas[2 * idx], &zero, &zero)
                                   : hs_set_bearing_residuals(q, 1, &stamp, &p->br_meas[3 * idx], &zero, &zero);
  } else if (!rc && type == HS_PRIOR) {
    rc = hs_set_sensors(q, 1, parameters[k]);
    stamp = p->pr_stamp[idx];
    if (!rc) rc = hs_set_prior_residuals(q, 1, &stamp, &p->pr_meas[7 * idx], &zero);
  } else if (!rc) {
    std::vector<double> bg(size_t(4) * kb), ba(size_t(4) * kb);
    for (int j = 0; j < kb; ++j) std::memcpy(&bg[4 * j], parameters[k + 5 + j], 32), std::memcpy(&ba[4 * j], parameters[k + 5 + kb + j], 32);
    rc = hs_set_imu(q, parameters[k], parameters[k + 1], parameters[k + 2], parameters[k + 3], parameters[k + 4], kb, bg[3], bg[7] - bg[3], kb,
                    bg.data(), ba.data(), 0);
    if (!rc) rc = hs_set_gravity(q, parameters[k + 5 + 2 * kb], 0);
    stamp = p->in_stamp[idx];
    if (!rc) rc = hs_set_inertial_residuals(q, 1, &stamp, &p->in_meas[6 * idx]);
  }
  if (rc) HS_FAIL(rc, std::string("hs_cost_function_evaluate: ") + hs_last_error(q));
  std::vector<double> r(6), Js(size_t(6) * 6 * k), Jl(18), Jbg(size_t(18) * kb), Jba(size_t(18) * kb), Jg(12);
  double Jext[36], Jintr[8], Jdist[8], Jig[36], Jia[36], JSg[54], JXa[54];
  hs_linearization lin;
  std::memset(&lin, 0, sizeof(lin));
  lin.r = r.data(), lin.J_state = Js.data(), lin.J_landmark = Jl.data(), lin.J_bias_g = Jbg.data(), lin.J_bias_a = Jba.data(), lin.J_gravity = Jg.data();
  if (jacobians) {  // sensor blocks (static_sensor_idx ..): only when asked for — a pass of their own
    const int s0 = L.indices[1];
    if (jacobians[s0]) lin.J_extrinsics = Jext;
    if (type == HS_PIXEL) {
      if (jacobians[s0 + 1]) lin.J_intrinsics = Jintr;
      if (jacobians[s0 + 2]) lin.J_distortion = Jdist;
    } else if (type == HS_INERTIAL) {
      if (jacobians[s0 + 1]) lin.J_gyro_intrinsics = Jig;
      if (jacobians[s0 + 2]) lin.J_acc_intrinsics = Jia;
      if (jacobians[s0 + 3]) lin.J_gyro_sensitivity = JSg;
      if (jacobians[s0 + 4]) lin.J_acc_offsets = JXa;
    }
  }
  rc = hs_linearize(q, type, /*robustify=*/0, &lin);
  if (rc) HS_FAIL(rc, std::string("hs_cost_function_evaluate: ") + hs_last_error(q));
  for (int i = 0; i < nres; ++i) residuals[i] = r[i];
  if (!jacobians) return HS_OK;
  for (int j = 0; j < k; ++j) {  // Stamped<SE3>: [q(4) p(3) t(1)], local [rot(3) trans(3)]
    if (!jacobians[j]) continue;
    double PT[12];
    quat_plus_jacobian_T(parameters[j], PT);
    for (int row = 0; row < nres; ++row) {
      const double* jl = &Js[(size_t(row) * k + j) * 6];
      double* out = jacobians[j] + size_t(row) * 8;
      for (int c = 0; c < 4; ++c) out[c] = jl[0] * PT[c] + jl[1] * PT[4 + c] + jl[2] * PT[8 + c];
      out[4] = jl[3], out[5] = jl[4], out[6] = jl[5], out[7] = 0.0;
    }
  }
  if (double* out = jacobians[k]) {  // sensor extrinsics SE3 [q(4) p(3)]: ambient = local * P^T as for the control points
    double PT[12];
    quat_plus_jacobian_T(parameters[k], PT);
    for (int row = 0; row < nres; ++row) {
      const double* jl = &Jext[6 * row];
      for (int c = 0; c < 4; ++c) out[7 * row + c] = jl[0] * PT[c] + jl[1] * PT[4 + c] + jl[2] * PT[8 + c];
      out[7 * row + 4] = jl[3], out[7 * row + 5] = jl[4], out[7 * row + 6] = jl[5];
    }
  }
  if (type == HS_PIXEL || type == HS_BEARING) {
    // intrinsics / distortion: Euclidean blocks; the bearing evaluator leaves them zero (bearing.cpp:58-77)
    if (jacobians[k + 1])
      for (int e = 0; e < nres * 4; ++e) jacobians[k + 1][e] = type == HS_PIXEL ? Jintr[e] : 0.0;
    if (jacobians[k + 2])
      for (int e = 0; e < nres * 4; ++e) jacobians[k + 2][e] = type == HS_PIXEL ? Jdist[e] : 0.0;
    if (jacobians[k + 3])
      for (int e = 0; e < nres * 3; ++e) jacobians[k + 3][e] = Jl[e];
  } else if (type == HS_INERTIAL) {
    if (jacobians[k + 1]) std::memcpy(jacobians[k + 1], Jig, sizeof(Jig));
    if (jacobians[k + 2]) std::memcpy(jacobians[k + 2], Jia, sizeof(Jia));
    if (jacobians[k + 3]) std::memcpy(jacobians[k + 3], JSg, sizeof(JSg));
    if (jacobians[k + 4]) std::memcpy(jacobians[k + 4], JXa, sizeof(JXa));
    for (int j = 0; j < kb; ++j)
      for (int part = 0; part < 2; ++part) {
        double* out = jacobians[k + 5 + part * kb + j];
        if (!out) continue;
        const std::vector<double>& J = part ? Jba : Jbg;
        for (int row = 0; row < 6; ++row) {
          for (int c = 0; c < 3; ++c) out[row * 4 + c] = J[(size_t(row) * kb + j) * 3 + c];
          out[row * 4 + 3] = 0.0;
        }
      }
    if (double* out = jacobians[k + 5 + 2 * kb]) {  // gravity: left inverse of the SphereManifold PlusJacobian = P^T / |x|^2
      const double* x = parameters[k + 5 + 2 * kb];
      // Householder basis as in Ceres (same code path as the device: recompute on the host)
      double v[3] = {x[0], x[1], 1.0}, beta = 0.0;
      const double sigma = x[0] * x[0] + x[1] * x[1];
      if (sigma <= 2.220446049250313e-16) {
        if (x[2] < 0) beta = 2.0;
      } else {
        const double mu = std::sqrt(x[2] * x[2] + sigma);
        const double vp = (x[2] <= 0.0) ? (x[2] - mu) : (-sigma / (x[2] + mu));
        beta = 2.0 * vp * vp / (sigma + vp * vp);
        v[0] /= vp, v[1] /= vp;
      }
      const double nx2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], nx = std::sqrt(nx2);
      double P[6];
      for (int i = 0; i < 2; ++i)
        for (int rr = 0; rr < 3; ++rr) P[rr * 2 + i] = nx * ((rr == i ? 1.0 : 0.0) - beta * v[rr] * v[i]);
      for (int row = 0; row < 6; ++row)
        for (int c = 0; c < 3; ++c) out[row * 3 + c] = (Jg[row * 2] * P[c * 2] + Jg[row * 2 + 1] * P[c * 2 + 1]) / nx2;
    }
  }
  return HS_OK;
}

/// Evaluation-only device tables (spline + cameras) for the entry points that do not touch residuals: the residual tables of the
/// last solve may refer to control points that the sliding window has already dropped, so prepare() is not run here.
static int eval_tables(hs_problem* p, Tables* T, DBuf<double>* d_cp, DBuf<double>* d_cam) {
  if (p->n_cp < p->k || p->cp.empty()) HS_FAIL(HS_ERR_STATE, "hs_set_spline has not been called");
  hipStream_t s = p->stream;
  HIP_TRY(d_cp->upload(p->cp, s));
  if (!p->cam.empty()) HIP_TRY(d_cam->upload(p->cam, s));
  std::memset(T, 0, sizeof(*T));
  T->sp = Spline{p->k, p->n_cp, p->t0, p->dt, 1.0 / p->dt, p->rot_const, p->trans_const};
  T->basis = make_basis_coef(p->k);
  T->cp = d_cp->p, T->cam = d_cam->p;
  return HS_OK;
}

int hs_process_tracks(hs_problem* p, double stamp, int n, const double* pixels0, const double* pixels1, double* bearings0, double* bearings1,
                      double* positions_w) {
  if (!p || n < 0 || (n && (!pixels0 || !pixels1))) return HS_ERR_INVALID;
  if (p->cam.size() < 32) HS_FAIL(HS_ERR_STATE, "hs_process_tracks needs a stereo pair (two cameras)");
  if (n == 0) return HS_OK;
  Tables T;
  DBuf<double> d_cp, d_cam;
  int rc = eval_tables(p, &T, &d_cp, &d_cam);
  if (rc) return rc;
  const int k = p->k, n_seg = p->n_cp - k + 1;
  const int f = h_segment_first(stamp, p->t0, p->dt, k);
  if (positions_w && (f < 0 || f >= n_seg)) HS_FAIL(HS_ERR_INVALID, "stamp outside the valid range of the spline");
  hipStream_t s = p->stream;
  DBuf<double> d_p0, d_p1, d_b0, d_b1, d_pw;
  HIP_TRY(d_p0.upload(std::vector<double>(pixels0, pixels0 + 2 * size_t(n)), s));
  HIP_TRY(d_p1.upload(std::vector<double>(pixels1, pixels1 + 2 * size_t(n)), s));
  if (bearings0) HIP_TRY(d_b0.reserve(size_t(3) * n));
  if (bearings1) HIP_TRY(d_b1.reserve(size_t(3) * n));
  if (positions_w) HIP_TRY(d_pw.reserve(size_t(3) * n));
  const int nb = (n + kBlock - 1) / kBlock;
  if (k == 4)
    k_process_tracks<4><<<nb, kBlock, cp_lds_bytes(p), s>>>(T, stamp, n, d_p0.p, d_p1.p, bearings0 ? d_b0.p : nullptr, bearings1 ? d_b1.p : nullptr,
                                                           positions_w ? d_pw.p : nullptr);
  else
    k_process_tracks<6><<<nb, kBlock, cp_lds_bytes(p), s>>>(T, stamp, n, d_p0.p, d_p1.p, bearings0 ? d_b0.p : nullptr, bearings1 ? d_b1.p : nullptr,
                                                           positions_w ? d_pw.p : nullptr);
  HIP_TRY(hipGetLastError());
  if (bearings0) HIP_TRY(hipMemcpyAsync(bearings0, d_b0.p, size_t(3) * n * 8, hipMemcpyDeviceToHost, s));
  if (bearings1) HIP_TRY(hipMemcpyAsync(bearings1, d_b1.p, size_t(3) * n * 8, hipMemcpyDeviceToHost, s));
  if (positions_w) HIP_TRY(hipMemcpyAsync(positions_w, d_pw.p, size_t(3) * n * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return HS_OK;
}

int hs_manifold_tangent_size(int kind, int ambient) {
  switch (kind) {
    case HS_MANIFOLD_CONSTANT: return (ambient >= 1 && ambient <= 9) ? 0 : -1;
    case HS_MANIFOLD_EUCLIDEAN: return (ambient >= 1 && ambient <= 9) ? ambient : -1;
    case HS_MANIFOLD_CONTROL_POINT: return ambient == 8 ? 6 : -1;
    case HS_MANIFOLD_SE3: return ambient == 7 ? 6 : -1;
    case HS_MANIFOLD_SPHERE3: return ambient == 3 ? 2 : -1;
    case HS_MANIFOLD_BIAS_POINT: return ambient == 4 ? 3 : -1;
    default: return -1;
  }
}

static int manifold_launch(hs_problem* p, int kind, int ambient, int n, const double* x, const double* delta, double* out, double* jac) {
  if (!p || n < 0 || (n && !x)) return HS_ERR_INVALID;
  const int tangent = hs_manifold_tangent_size(kind, ambient);
  if (tangent < 0) HS_FAIL(HS_ERR_INVALID, "unknown manifold kind / ambient size");
  if (out && tangent > 0 && !delta) HS_FAIL(HS_ERR_INVALID, "delta is null");
  if (n == 0) return HS_OK;
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  DBuf<double> d_x, d_d, d_o, d_j;
  HIP_TRY(d_x.upload(std::vector<double>(x, x + size_t(n) * ambient), s));
  if (out && tangent > 0) HIP_TRY(d_d.upload(std::vector<double>(delta, delta + size_t(n) * tangent), s));
  if (out) HIP_TRY(d_o.reserve(size_t(n) * ambient));
  if (jac && tangent > 0) HIP_TRY(d_j.reserve(size_t(n) * ambient * tangent));
  k_manifold_plus<<<(n + kBlock - 1) / kBlock, kBlock, 0, s>>>(kind, ambient, tangent, n, d_x.p, d_d.p, out ? d_o.p : nullptr,
                                                             (jac && tangent > 0) ? d_j.p : nullptr);
  HIP_TRY(hipGetLastError());
  if (out) HIP_TRY(hipMemcpyAsync(out, d_o.p, size_t(n) * ambient * 8, hipMemcpyDeviceToHost, s));
  if (jac && tangent > 0) HIP_TRY(hipMemcpyAsync(jac, d_j.p, size_t(n) * ambient * tangent * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return HS_OK;
}

int hs_manifold_plus(hs_problem* p, int kind, int ambient, int n, const double* x, const double* delta, double* x_plus_delta) {
  if (n > 0 && !x_plus_delta) return HS_ERR_INVALID;
  return manifold_launch(p, kind, ambient, n, x, delta, x_plus_delta, nullptr);
}

int hs_manifold_plus_jacobian(hs_problem* p, int kind, int ambient, int n, const double* x, double* jacobian) {
  if (n > 0 && !jacobian) return HS_ERR_INVALID;
  return manifold_launch(p, kind, ambient, n, x, nullptr, nullptr, jacobian);
}

static int manifold_minus_launch(hs_problem* p, int kind, int ambient, int n, const double* y, const double* x, double* out, double* jac) {
  if (!p || n < 0 || (n && !x)) return HS_ERR_INVALID;
  const int tangent = hs_manifold_tangent_size(kind, ambient);
  if (tangent < 0) HS_FAIL(HS_ERR_INVALID, "unknown manifold kind / ambient size");
  if (out && !y) HS_FAIL(HS_ERR_INVALID, "y is null");
  if (n == 0 || tangent == 0) return HS_OK;
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  DBuf<double> d_x, d_y, d_o, d_j;
  HIP_TRY(d_x.upload(std::vector<double>(x, x + size_t(n) * ambient), s));
  if (out) HIP_TRY(d_y.upload(std::vector<double>(y, y + size_t(n) * ambient), s));
  if (out) HIP_TRY(d_o.reserve(size_t(n) * tangent));
  if (jac) HIP_TRY(d_j.reserve(size_t(n) * ambient * tangent));
  k_manifold_minus<<<(n + kBlock - 1) / kBlock, kBlock, 0, s>>>(kind, ambient, tangent, n, out ? d_y.p : nullptr, d_x.p, out ? d_o.p : nullptr, jac ? d_j.p : nullptr);
  HIP_TRY(hipGetLastError());
  if (out) HIP_TRY(hipMemcpyAsync(out, d_o.p, size_t(n) * tangent * 8, hipMemcpyDeviceToHost, s));
  if (jac) HIP_TRY(hipMemcpyAsync(jac, d_j.p, size_t(n) * ambient * tangent * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return HS_OK;
}

int hs_manifold_minus(hs_problem* p, int kind, int ambient, int n, const double* y, const double* x, double* y_minus_x) {
  if (n > 0 && hs_manifold_tangent_size(kind, ambient) > 0 && !y_minus_x) return HS_ERR_INVALID;
  return manifold_minus_launch(p, kind, ambient, n, y, x, y_minus_x, nullptr);
}

int hs_manifold_minus_jacobian(hs_problem* p, int kind, int ambient, int n, const double* x, double* jacobian) {
  if (n > 0 && hs_manifold_tangent_size(kind, ambient) > 0 && !jacobian) return HS_ERR_INVALID;
  return manifold_minus_launch(p, kind, ambient, n, nullptr, x, nullptr, jacobian);
}

int hs_sample_trajectory(hs_problem* p, int n, const double* stamps, double* pose, double* velocity, double* acceleration) {
  if (!p || n < 0 || (n && (!stamps || !pose))) return HS_ERR_INVALID;
  if (n == 0) return HS_OK;
  Tables T;
  DBuf<double> d_cp, d_cam;
  int rc = eval_tables(p, &T, &d_cp, &d_cam);
  if (rc) return rc;
  const int k = p->k, n_seg = p->n_cp - k + 1;
  for (int i = 0; i < n; ++i) {
    const int f = h_segment_first(stamps[i], p->t0, p->dt, k);
    if (f < 0 || f >= n_seg) HS_FAIL(HS_ERR_INVALID, "stamp outside the valid range of the spline");
  }
  hipStream_t s = p->stream;
  DBuf<double> d_st, d_pose, d_vel, d_acc;
  std::vector<double> st(stamps, stamps + n);
  HIP_TRY(d_st.upload(st, s));
  HIP_TRY(d_pose.reserve(size_t(7) * n));
  if (velocity) HIP_TRY(d_vel.reserve(size_t(6) * n));
  if (acceleration) HIP_TRY(d_acc.reserve(size_t(6) * n));
  const int nb = (n + kBlock - 1) / kBlock;
  if (k == 4)
    k_sample_trajectory<4><<<nb, kBlock, cp_lds_bytes(p), s>>>(T, n, d_st.p, d_pose.p, velocity ? d_vel.p : nullptr, acceleration ? d_acc.p : nullptr);
  else
    k_sample_trajectory<6><<<nb, kBlock, cp_lds_bytes(p), s>>>(T, n, d_st.p, d_pose.p, velocity ? d_vel.p : nullptr, acceleration ? d_acc.p : nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(pose, d_pose.p, size_t(7) * n * 8, hipMemcpyDeviceToHost, s));
  if (velocity) HIP_TRY(hipMemcpyAsync(velocity, d_vel.p, size_t(6) * n * 8, hipMemcpyDeviceToHost, s));
  if (acceleration) HIP_TRY(hipMemcpyAsync(acceleration, d_acc.p, size_t(6) * n * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return HS_OK;
}

}  // extern "C"
