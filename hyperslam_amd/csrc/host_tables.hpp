// host_tables.hpp — the window on the host and in HBM: hs_problem (caller's tables, device buffers, sort orders), the batched table upload
// and prepare(), which turns the caller's tables into the sorted device tables of problem.hpp (part of capi.hip: included once, by it).
#pragma once
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

// HS_GUARD=1 (a debugging mode of the library, read once per process): every device table is allocated at exactly the size asked for, followed
// by kGuardBytes of a known pattern, and hs_solve / hs_cost / hs_reduced_system / hs_linearize check the patterns of every table of the process
// before they return — a kernel that writes past the end of its table fails the call with the table's size instead of corrupting a neighbour.
// (Reads past the end are not caught.) tests/test_gpu_edge_cases.py::test_guarded_tables and tools/fuzz_parity.py run under it.
constexpr size_t kGuardBytes = 4096;
struct GuardedBuffer {
  void* base = nullptr;   // device allocation
  size_t bytes = 0;       // size the owner asked for; the pattern follows
};
struct GuardRegistry {
  std::mutex mu;
  std::vector<GuardedBuffer*> all;
  static GuardRegistry& get() {
    static GuardRegistry r;
    return r;
  }
  /// HS_GUARD=1 in the environment, or hs_set_guard (a process-wide switch: tables allocated while it is on carry the pattern and are checked,
  /// the others are left alone — a test session turns it on before its first handle and has every table of every test guarded).
  static int& flag() {
    static int v = [] {
      const char* e = std::getenv("HS_GUARD");
      return (e && std::atoi(e) != 0) ? 1 : 0;
    }();
    return v;
  }
  static bool on() { return flag() != 0; }
  /// 0: every pattern intact; otherwise the size in bytes of a table whose pattern was overwritten (first found), offset of the first bad byte in *at.
  size_t check(size_t* at) {
    std::lock_guard<std::mutex> lock(mu);
    std::vector<unsigned char> h(kGuardBytes);
    (void)hipDeviceSynchronize();
    for (GuardedBuffer* b : all) {
      if (!b->base) continue;
      if (hipMemcpy(h.data(), static_cast<char*>(b->base) + b->bytes, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) continue;
      for (size_t i = 0; i < kGuardBytes; ++i)
        if (h[i] != 0xA5) {
          *at = i;
          return b->bytes ? b->bytes : 1;
        }
    }
    return 0;
  }
};

template <class T>
struct DBuf {
  T* p = nullptr;
  size_t cap = 0;
  GuardedBuffer guard;
  DBuf() {  // (registered whatever the mode: the switch may be turned on later, and a table without a pattern is skipped by the check)
    std::lock_guard<std::mutex> lock(GuardRegistry::get().mu);
    GuardRegistry::get().all.push_back(&guard);
  }
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  ~DBuf() {
    {
      std::lock_guard<std::mutex> lock(GuardRegistry::get().mu);
      auto& v = GuardRegistry::get().all;
      v.erase(std::remove(v.begin(), v.end(), &guard), v.end());
    }
    if (p) (void)hipFree(p);
  }
  /// Capacity grows geometrically: a sliding window changes every table size by a little at every optimize(), and an exact-fit
  /// hipFree + hipMalloc per table and solve costs more than the solve itself (hipFree synchronises the device).
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr, guard.base = nullptr;
    if (GuardRegistry::on()) {  // exact fit + pattern (see GuardRegistry)
      cap = 0;
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T) + kGuardBytes);
      if (e != hipSuccess) return e;
      e = hipMemset(reinterpret_cast<char*>(p) + n * sizeof(T), 0xA5, kGuardBytes);
      if (e != hipSuccess) return e;
      e = hipDeviceSynchronize();
      cap = n, guard.base = p, guard.bytes = n * sizeof(T);
      return e;
    }
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
  int n_cu = 256;  // compute units of the device (hs_create): the fused build orders its chunks by how the hardware places workgroups on them
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  bool dirty = true;  // tables changed since the last prepare()
  // What changed (prepare() redoes only that). Structure sections: the sorted visual / prior / inertial tables; kTail = sizes + Tables only.
  // Values: tables whose content changed while every size, index and sort order stayed (a sliding window re-sends the control points before
  // every solve; the delta interface — hs_append_* / hs_retire_* / hs_stage — keeps the rest resident between solves).
  enum : unsigned { kVis = 1, kPri = 2, kIne = 4, kTail = 8, kStructure = 15, vCp = 16, vCam = 32, vSensor = 64, vLm = 128, vImu = 256, vGravity = 512, vBias = 1024, kValues = 2032, kAll = 2047 };
  unsigned changed = kAll;
  void touch(unsigned what) { changed |= what, dirty = true; }
  bool device_ahead = false;  // hs_solve moved the point on the device and the host copies have not been refreshed since (pull_state)

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
  size_t vb_len = 0;                             // doubles of d_Vb / d_Vb2 in front of their zero pad (prepare())
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
  DBuf<double> d_xbuf, d_xpart, d_segP, d_grpQ, d_Qw, d_Yt, d_gravity_part;
  int yt_stride = 0;
  bool wide_q = false;  // fused build on window-wide bands: landmark term once per window (k_landmark_gram_wide; HS_WIDE_Q=0: per chunk, round 5's arrangement)
  DBuf<double> d_Vb, d_Vb2, d_yt, d_yt2;  // block-row-scaled factors diag(U_jj^-1) U and right-hand sides for the register sweep
  DBuf<double> d_Sb2, d_g2, d_Ub2, d_Ubk2, d_ybuf2, d_win, d_xsol;  // two-ended factorisation: reversed system, its factor, junction window
  DBuf<unsigned> d_join;
  DBuf<double> d_dense_ut;     // k_dense_solve_mx: the factor by columns (256 x 256), read back by its sweep
  DBuf<double> d_bf_handover;  // k_border_forward2: what the far end's sweep leaves on the middle rows, per column group
  unsigned join_epoch = 0;
  DBuf<int> d_gw_ptr, d_gw_cf, d_sw_ptr, d_sw_seg;
  int n_seg_wg = 0, n_group_wg = 0;
  // fused build of the visual factors (kernels_build.hpp)
  bool fused = false;
  int dense_border_nb = -1;         // border size the padding of the dense copy was written for (k_dense_border_init, launch_factor); -1: not yet
  bool bookkeep = false;            // this linearisation's rows were finalised by k_assemble: the factorisation does the iteration bookkeeping (launch_build)
  int build_R = 0, build_L = 0;     // records per pass, landmarks per chunk
  size_t build_lds = 0;
  DBuf<int> d_ch_ptr, d_ch_desc;
  DBuf<double> d_ch_gmax;
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

/// Spline orders the device kernels are instantiated for: 4, 5 and 6. Runs the statement(s) with the compile-time constant K = k.
#define HS_ORDER_SWITCH(k, ...)    \
  do {                             \
    if ((k) == 4) {                \
      constexpr int K = 4;         \
      __VA_ARGS__;                 \
    } else if ((k) == 5) {         \
      constexpr int K = 5;         \
      __VA_ARGS__;                 \
    } else {                       \
      constexpr int K = 6;         \
      __VA_ARGS__;                 \
    }                              \
  } while (0)

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

/// Value tables whose content changed (hs_problem::changed): control points + constancy mask, cameras, sensors, landmarks (device order), IMU
/// parameters, gravity, bias points. Sizes, indices and sort orders are those of the last structural prepare().
static int upload_values(hs_problem* p, unsigned what) {
  hipStream_t s = p->stream;
  if (what & hs_problem::vCp) {
    HIP_TRY(p->d_cp.upload(p->cp, s));
    HIP_TRY(p->d_cp_cand.reserve(p->cp.size()));
    HIP_TRY(p->d_cp_const.upload(p->cp_const, s));
    p->frozen_prefix = 0;
    while (p->frozen_prefix < p->n_cp && p->cp_const[p->frozen_prefix]) ++p->frozen_prefix;
  }
  if (what & hs_problem::vCam) HIP_TRY(p->d_cam.upload(p->cam, s));
  if (what & hs_problem::vSensor) HIP_TRY(p->d_sensor.upload(p->sensor, s));
  if (what & hs_problem::vLm) {  // landmarks in device order
    const VisualStructure& vs = p->vs;
    HIP_TRY(p->d_lm.reserve(size_t(3) * p->n_lm));
    HIP_TRY(p->d_lm_const.reserve(p->n_lm));
    if (p->n_lm) {
      void *a0, *a1;
      HIP_TRY(p->batch.reserve_more(size_t(p->n_lm) * 25 + 64));
      HIP_TRY(p->batch.alloc(p->d_lm.p, size_t(p->n_lm) * 24, &a0));
      HIP_TRY(p->batch.alloc(p->d_lm_const.p, size_t(p->n_lm), &a1));
      double* lm_dev = static_cast<double*>(a0);
      uint8_t* lmc_dev = static_cast<uint8_t*>(a1);
      for (int d = 0; d < p->n_lm; ++d) {
        const int t = vs.table_of_dev[d];
        for (int c = 0; c < 3; ++c) lm_dev[3 * d + c] = p->lm[3 * t + c];
        lmc_dev[d] = p->lm_const[t];
      }
    }
    HIP_TRY(p->d_lm_cand.reserve(size_t(3) * p->n_lm));
  }
  if (what & hs_problem::vImu) {
    std::vector<ImuParams> ip(1);
    std::memcpy(ip[0].T_bs, p->imu_T_bs, 56), std::memcpy(ip[0].i_g, p->imu_i_g, 48), std::memcpy(ip[0].i_a, p->imu_i_a, 48);
    std::memcpy(ip[0].S_g, p->imu_S_g, 72), std::memcpy(ip[0].X_a, p->imu_X_a, 72);
    HIP_TRY(p->d_imu.upload(ip, s));
  }
  if (what & hs_problem::vGravity) {
    std::vector<double> grav(p->gravity, p->gravity + 3);
    HIP_TRY(p->d_gravity.upload(grav, s));
    HIP_TRY(p->d_gravity_cand.reserve(3));
  }
  if (what & hs_problem::vBias) {
    HIP_TRY(p->d_bias_g.upload(p->bias_g, s));
    HIP_TRY(p->d_bias_a.upload(p->bias_a, s));
    HIP_TRY(p->d_bias_g_cand.reserve(p->bias_g.size() + 1));
    HIP_TRY(p->d_bias_a_cand.reserve(p->bias_a.size() + 1));
  }
  return HS_OK;
}

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
  if (p->k < 4 || p->k > 6) HS_FAIL(HS_ERR_INVALID, "device kernels are instantiated for spline orders 4, 5 and 6");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  // A failed prepare() leaves `changed` as it was: the next one redoes the same sections (the tables of a refused window stay refused).
  const unsigned what = p->changed;
  if (!(what & hs_problem::kStructure)) {  // values only: every size, index and sort order of the resident tables stands
    const int rc = upload_values(p, what);
    if (rc) return rc;
    HIP_TRY(p->batch.flush(s));
    p->dirty = false, p->changed = 0;
    if (p->host_timing) {
      const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
      p->host_prepare[1] += t, p->host_ms[1] += t;
    }
    return HS_OK;
  }
  const int k = p->k, n_seg = p->n_cp - k + 1;
  const int n_px = int(p->px_stamp.size()), n_br = int(p->br_stamp.size());
  const int n_vis = n_px + n_br;
  const int n_pri = int(p->pr_stamp.size());
  const int n_ine = int(p->in_stamp.size());
  if (what & hs_problem::kVis) {
    for (int i = 0; i < n_px; ++i)
      if (p->px_cam[i] < 0 || p->px_cam[i] >= p->n_cam) HS_FAIL(HS_ERR_INVALID, "pixel residual references a camera outside the camera table");
    for (int i = 0; i < n_br; ++i)
      if (p->br_cam[i] < 0 || p->br_cam[i] >= p->n_cam) HS_FAIL(HS_ERR_INVALID, "bearing residual references a camera outside the camera table");
    VisualInput in = {k, p->n_cp, p->n_lm, p->t0, p->dt, n_px, n_br, p->px_stamp.data(), p->br_stamp.data(), p->px_lm.data(), p->br_lm.data()};
    if (!build_visual_structure(in, &p->vs, &p->err)) return HS_ERR_INVALID;
    p->vs.bw = std::max(p->vs.bw, p->min_bw);
  }
  const VisualStructure& vs = p->vs;
  if (6 * vs.bw > kBlock || (size_t(42) * (6 * vs.bw + 2) + size_t(6) * p->n_cp + 48) * 8 > size_t(p->chol_lds_max))
    HS_FAIL(HS_ERR_INVALID, "landmark tracks span too many control points for the LDS-resident banded factorisation");
  if (p->n_cp > 1024)  // 64 KiB of control points staged per workgroup; 96 KiB right-hand side + 49 KiB junction block in the backward sweep
    HS_FAIL(HS_ERR_INVALID, "window too long for the LDS-resident control-point table and backward sweep (more than 1024 control points)");

  // ---- prior tables (segment-major) ----
  std::vector<double> p_stamp, p_meas;
  std::vector<int> p_sensor;
  if (what & hs_problem::kPri) {
    p->pr_order.resize(n_pri);
    std::vector<int> first_tab(n_pri);
    for (int i = 0; i < n_pri; ++i) {
      first_tab[i] = h_segment_first(p->pr_stamp[i], p->t0, p->dt, k);
      if (first_tab[i] < 0 || first_tab[i] >= n_seg) HS_FAIL(HS_ERR_INVALID, "prior residual stamp outside the valid range of the spline");
      if (p->pr_sensor[i] < 0 || p->pr_sensor[i] >= p->n_sensor) HS_FAIL(HS_ERR_INVALID, "prior residual references a sensor outside the sensor table");
      p->pr_order[i] = i;
    }
    std::stable_sort(p->pr_order.begin(), p->pr_order.end(), [&](int a, int b) { return first_tab[a] < first_tab[b]; });
    p_stamp.resize(n_pri), p_meas.resize(size_t(7) * n_pri), p_sensor.resize(n_pri);
    p->pr_first.resize(n_pri);
    p->pr_seg_ptr.assign(n_seg + 1, 0);
    for (int d = 0; d < n_pri; ++d) {
      const int t = p->pr_order[d];
      p_stamp[d] = p->pr_stamp[t], p_sensor[d] = p->pr_sensor[t], p->pr_first[d] = first_tab[t];
      for (int c = 0; c < 7; ++c) p_meas[7 * d + c] = p->pr_meas[7 * t + c];
      p->pr_seg_ptr[first_tab[t] + 1]++;
    }
    for (int sgm = 0; sgm < n_seg; ++sgm) p->pr_seg_ptr[sgm + 1] += p->pr_seg_ptr[sgm];
  }
  // ---- inertial tables (segment-major) ----
  if (n_ine && !p->has_imu) HS_FAIL(HS_ERR_STATE, "inertial residuals need hs_set_imu");
  if (n_ine && p->kb != 4) HS_FAIL(HS_ERR_INVALID, "device kernels are instantiated for bias-spline order 4");
  std::vector<double> i_stamp, i_meas;
  if (what & hs_problem::kIne) {
    i_stamp.resize(n_ine), i_meas.resize(size_t(6) * n_ine);
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
  {  // (a structural change of the visual tables re-orders the landmarks on the device: their values go with it)
    const int rc = upload_values(p, (what & hs_problem::kValues) | ((what & hs_problem::kVis) ? unsigned(hs_problem::vLm) : 0u));
    if (rc) return rc;
  }
  const size_t nl = size_t(std::max(p->n_lm, 1));
  if (what & hs_problem::kVis) {
    HIP_TRY(p->d_lm_ptr.upload(vs.lm_ptr, s));
    HIP_TRY(p->d_lm_cfirst.upload(vs.lm_cfirst, s));
    HIP_TRY(p->d_lm_ncp.upload(vs.lm_ncp, s));
    HIP_TRY(p->d_lm_yoff.upload(vs.lm_yoff, s));
    HIP_TRY(p->d_cf_ptr.upload(vs.cf_ptr, s));
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
    // ---- fused build (kernels_build.hpp) or the record path (a lane group per J'J band tile: k bw - k (k - 1) / 2 <= 256, i.e. bands of up to 65 / 53 /
    //      45 control points at orders 4 / 5 / 6; a landmark with more residuals than a chunk has lanes; no chunk geometry that fits the LDS) ----
    //      Window-wide bands (the steady state of a sliding window: tracks as long as the window, bw(bw + 1) / 2 > 256 window tiles) take the
    //      tiles of the landmark term in passes (kernels_build.hpp phase 5). HS_BUILD_PATH=records: the record path everywhere; =narrow: the
    //      fused build for at most 256 window tiles only (round 4's rule) — measurement switches, like HS_DEBUG_FLAGS.
    {
      const int ntile_ = vs.bw * (vs.bw + 1) / 2, nband_ = k * vs.bw - k * (k - 1) / 2;
      const char* env = std::getenv("HS_BUILD_PATH");
      const bool narrow_only = env && std::strcmp(env, "narrow") == 0;
      p->fused = n_vis > 0 && nband_ <= kBlock && vs.bw <= 64 && (ntile_ <= kBlock || !narrow_only) && !(env && std::strcmp(env, "records") == 0);
    }
    if (p->fused) {
      // chunk geometry: R residuals (lanes) and L landmarks per chunk, sized for two workgroups per CU (every phase of the kernel is an LDS
      // gather: latency bound on a lone wave per SIMD). HS_BUILD_R / HS_BUILD_L: tuning overrides.
      int R0 = k == 4 ? 128 : k == 5 ? 112 : 96, L0 = k == 4 ? 12 : k == 5 ? 11 : 10;
      if (const char* e = std::getenv("HS_BUILD_R")) R0 = std::max(32, std::min(kBlock, std::atoi(e)));
      if (const char* e = std::getenv("HS_BUILD_L")) L0 = std::max(1, std::min(24, std::atoi(e)));  // (<= 24: 9 L + 8 lanes of phase 2a, L lanes of one wave in 4a)
      auto lds_bytes = [&](int r, int l) { return size_t(build_lds_layout(k, vs.bw, r, l).total_doubles) * 8; };
      const bool overridden = std::getenv("HS_BUILD_R") || std::getenv("HS_BUILD_L");  // (a tuning run asks for exactly this geometry, one workgroup per CU if need be)
      p->fused = choose_build_geometry(k, R0, L0, size_t(overridden ? 156 : 79) * 1024, size_t(156) * 1024, lds_bytes, &p->build_R, &p->build_L) &&
                 build_chunks(vs, p->n_cp, p->build_R, p->build_L, &p->h_ch_ptr, &p->h_gw_ptr, &p->h_gw_cf, &p->h_ch_desc);
      p->build_lds = p->fused ? lds_bytes(p->build_R, p->build_L) : 0;
      if (p->fused && !std::getenv("HS_BUILD_ORDER"))  // (HS_BUILD_ORDER=table: chunks in table order, measurement switch)
        order_chunks_for_dispatch(vs, k, p->n_cu, &p->h_ch_desc, int(p->h_ch_ptr.size()) - 1);
    }
    if (!p->fused) {
      HIP_TRY(p->d_v_rec.reserve(size_t(n_vis) * (8 + 12 * k) + 1));
      HIP_TRY(p->d_v_rec_alt.reserve(size_t(n_vis) * (8 + 12 * k) + 1));
    }
  }
  if (what & hs_problem::kPri) {
    HIP_TRY(p->d_p_stamp.upload(p_stamp, s));
    HIP_TRY(p->d_p_meas.upload(p_meas, s));
    HIP_TRY(p->d_p_sensor.upload(p_sensor, s));
    HIP_TRY(p->d_p_first.upload(p->pr_first, s));
    HIP_TRY(p->d_p_seg_ptr.upload(p->pr_seg_ptr, s));
    HIP_TRY(p->d_p_rec.reserve(size_t(n_pri) * (6 + 36 * k) + 1));
  }
  if (what & hs_problem::kIne) {
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
  }
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
  int vis_block = 0;
  HS_ORDER_SWITCH(k, vis_block = lin_block<K>());
  p->nb_vis = (n_vis + vis_block - 1) / vis_block;
  if (p->fused) {
    p->nb_vis = std::max((n_vis + kBlock - 1) / kBlock, int(p->h_ch_ptr.size()) - 1);  // one cost partial per chunk / per workgroup of k_cost_visual
    p->h_ch_desc.resize(size_t(8) * p->nb_vis, 0);                                        // (k_build_visual reads its descriptor before it knows whether it is a padding workgroup)
  }
  p->nb_pri = (n_pri + kBlock - 1) / kBlock;
  p->nb_cp = std::max((p->n_cp + kBlock - 1) / kBlock, 1);
  // (one slot per workgroup of the kernels that write them: the inertial ones take kInertialBlock = 64 residuals each — sized by kBlock = 256 the
  //  table was four times too short for them, an out-of-bounds write of a few hundred bytes that the allocator's granularity hid until a
  //  randomised sweep of large windows hit a page boundary, tools/fuzz_parity.py large)
  HIP_TRY(p->d_cost_part.reserve(p->nb_vis + p->nb_pri + p->nb_ine + 1));
  HIP_TRY(p->d_ch_gmax.reserve(size_t(p->nb_vis) + 1));
  HIP_TRY(p->d_cand_part.reserve(p->nb_vis + p->nb_pri + p->nb_ine + 1));
  const int nb_norm = p->nb_cp;
  HIP_TRY(p->d_norm_part.reserve(2 * size_t(nb_norm)));
  const int nbd = p->has_imu ? 6 * p->n_bias + 2 : 0;
  if (nbd && size_t(np) * 8 * 8 > 150 * 1024) HS_FAIL(HS_ERR_INVALID, "window too long for the LDS-resident border forward sweep");
  // (k_border_solve: the border Schur complement, augmented, in LDS — (nb + 1)^2 + nb doubles within the 150 KB the kernel may ask for: nb <= 137,
  //  i.e. 22 bias control points; the launch used to fail inside hs_solve with "invalid argument")
  if (nbd + 1 > 128 && (size_t(nbd + 1) * (nbd + 1) + nbd) * sizeof(double) > size_t(150) * 1024)
    HS_FAIL(HS_ERR_INVALID, "too many border unknowns (bias control points) for the LDS-resident dense solve of the border system: at most 22 bias control points per window");
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
    // (Vb / Vb2 also hold the inverted super-blocks of the backward sweep, 576 doubles per four block rows, the last one whole even when the
    //  system ends inside it: for bands of four control points — windows of priors / inertial residuals only — that is MORE than np x ncb, and
    //  the builders wrote up to 432 doubles past the table, the zero operand of k_band_factor_mx behind it included; found by HS_GUARD=1)
    const size_t nv = size_t(np) * ncb, pad = 64, vb_len = std::max(nv, size_t(576) * ((size_t(p->n_cp) + 3) / 4 + 1));
    p->vb_len = vb_len;
    HIP_TRY(p->d_Sb2.reserve(nv));
    for (DBuf<double>* b : {&p->d_Vb, &p->d_Vb2}) HIP_TRY(b->reserve(vb_len + pad));
    for (DBuf<double>* b : {&p->d_yt, &p->d_yt2}) HIP_TRY(b->reserve(size_t(np) + pad));
    const void* now[5] = {p->d_Sb2.p, p->d_Vb.p, p->d_Vb2.p, p->d_yt.p, p->d_yt2.p};
    if (p->zeroed_np != np || p->zeroed_ncb != ncb || std::memcmp(now, p->zeroed_ptr, sizeof(now)) != 0) {
      HIP_TRY(hipMemsetAsync(p->d_Sb2.p, 0, p->d_Sb2.cap * sizeof(double), s));
      for (DBuf<double>* b : {&p->d_Vb, &p->d_Vb2}) HIP_TRY(hipMemsetAsync(b->p + vb_len, 0, pad * sizeof(double), s));
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
  HIP_TRY(p->d_dense_ut.reserve(size_t(2) * kDenseLd * kDenseLd));  // the factor by columns | the dense copy of the system (Tables::dense)
  if (!p->d_join.p) {
    // [0] two-ended factor / sweep hand-over, [1] last-block ticket of the backward sweeps, [2] of k_border_bb, [4 ..] super-block inverses
    HIP_TRY(p->d_join.reserve(kGatherFlag + kProgressStride));  // (+ one flag per column group of k_border_forward2, + the four progress words of a pipelined sweep, + kGatherFlag)
    HIP_TRY(hipMemsetAsync(p->d_join.p, 0, (kGatherFlag + kProgressStride) * sizeof(unsigned), s));
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
    if (!(what & hs_problem::kVis)) {  // (the work list of the visual factors stands)
    } else if (p->fused) {  // the chunks of the fused build take the place of the k_group_gram workgroups
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
    {
      const char* env = std::getenv("HS_WIDE_Q");
      p->wide_q = p->fused && ntile > kBlock && !(env && std::atoi(env) == 0);
      if (p->wide_q) {
        HIP_TRY(p->d_Qw.reserve(size_t(6) * p->n_cp * (6 * vs.bw + 1) + 1));
        p->yt_stride = (p->n_lm + 63) / 64 * 64;
        HIP_TRY(p->d_Yt.reserve(size_t(18) * p->n_cp * p->yt_stride + 2));
      }
    }
  }
  HIP_TRY(p->d_state.reserve(1));

  Tables& T = p->T;
  std::memset(&T, 0, sizeof(T));
  p->dense_border_nb = -1;
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
  if (size_t(T.n_cost_part) > p->d_cost_part.cap || size_t(T.n_cost_part) > p->d_cand_part.cap)  // (the tables are sized above from the same three grid sizes)
    HS_FAIL(HS_ERR_DEVICE, "internal: cost partial tables shorter than the grids that write them");
  T.norm_part = p->d_norm_part.p, T.n_norm_part = nb_norm;
  T.xbuf = p->d_xbuf.p;
  T.xpart = p->d_xpart.p, T.gravity_part = p->d_gravity_part.p, T.segP = p->d_segP.p, T.grpQ = p->d_grpQ.p, T.Qw = p->d_Qw.p, T.wide_q = p->wide_q && n_vis > 0 ? 1 : 0, T.Yt = p->d_Yt.p, T.yt_stride = p->yt_stride, T.gw_ptr = p->d_gw_ptr.p, T.gw_cf = p->d_gw_cf.p, T.sw_ptr = p->d_sw_ptr.p, T.sw_seg = p->d_sw_seg.p;
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
  T.build_stream_lg = p->fused ? build_streams_packed(vs.bw, k) : 0;
  T.ch_gmax = p->d_ch_gmax.p;
  T.rank = p->rank, T.world = p->world;
  // HS_DEBUG_FLAGS (measurement switches only, never needed for correct operation):
  //    1 skip the backward sweep          2 skip the rank-6 updates (timing of the panel chain alone; results are garbage)
  //    4 one-ended pre-look-ahead factorisation kernel                16 phase timestamps of the factorisation -> hs_debug_read
  //   32 per-workgroup timestamps of the linearise / gram kernels   1024 no side stream for the segment partials
  //   64 two-ended factorisation on the VALU look-ahead kernel (k_band_factor_la) instead of k_band_factor_mx (bw <= 14)
  // 2048 one-ended factorisation (no second workgroup)              8192 generalised backward sweep on the one-ended factor
  // 131072 k_band_factor_mfma (trailing window in f64 MFMA tiles) instead of the VALU factorisation kernels
  // 65536 single-wave register backward sweep (k_band_backward_w) instead of the four-wave LDS sweeps
  // 262144 eliminate / sweep the decoupled block rows of leading constant control points like any other     524288 border Cholesky in LDS
  // 128 border forward sweep behind the factorisation instead of alongside it
  //   8 small systems (6 n_free + nb <= 256) on the one-ended band kernels + border chain + k_band_backward instead of k_dense_solve_mx
  // 256 phase timestamps of k_assemble (profiling builds)          512 phase timestamps of k_update_visual (profiling builds)
  //   (64 and 128 are product A/B switches: the stamps of k_assemble / k_update_visual used to share them, so that timing one of those kernels
  //    also changed the factorisation path)
  // 1048576 inertial branch on the main stream    2097152 banded kernels instead of k_dense_factor    4194304 k_landmark<K,4,1> instead of k_landmark_rows
  // 8192 with a border: k_border_solve_reg instead of k_dense_solve_mx on the border Schur complement of a two-ended system
  // 8388608 five finalisation launches for a bordered single shard    16777216 k_commit launch for small windows    33554432 one cost launch per factor type
  // 134217728 prior / inertial candidate costs as launches of their own behind k_update_visual (single shard, fused path)
  // 16384 border gathers / pipelined border sweep joined to the main stream by events instead of device flags (Tables::gather_epoch, sweep_epoch)
  // 268435456 backward sweeps one block row per step    536870912 bordered systems one-ended    1073741824 no speculative linearisation at the candidate    67108864 k_commit in every iteration of a speculative solve
  T.st = p->d_state.p;
  HIP_TRY(p->batch.flush(s));  // (the staging arena outlives this call: no host synchronisation)
  p->dirty = false, p->changed = 0;
  if ((what & hs_problem::kValues) == hs_problem::kValues) p->device_ahead = false;  // (every variable was just sent: host and device agree)
  if (p->host_timing) {
    const auto host_t2 = std::chrono::steady_clock::now();
    p->host_prepare[0] = std::chrono::duration<double, std::milli>(host_t1 - host_t0).count();
    p->host_prepare[1] = std::chrono::duration<double, std::milli>(host_t2 - host_t1).count();
    p->host_ms[0] += p->host_prepare[0], p->host_ms[1] += p->host_prepare[1];
  }
  return HS_OK;
}


}  // namespace
