// hip/hip_runtime.h — TEST INFRASTRUCTURE: a CPU stand-in for the HIP device environment, written for this repository's tests.
//
// tests/emul/ compiles the product's kernel SOURCES (hyperslam_amd/csrc/kernels_*.hpp) for the host with g++ and runs a workgroup as
// one std::thread per lane: __syncthreads() is a barrier over the workgroup's live threads, the wave intrinsics the kernels use
// (__shfl_xor, __ballot, readlane) exchange values through a per-wave buffer. That makes the index arithmetic, LDS layouts and
// summation orders of a kernel checkable against the oracle in the `-m "not gpu"` suite, before a GPU is involved. It says nothing about
// timing, occupancy or memory-model subtleties (x86 is sequentially consistent enough for barrier-synchronised code); the `-m gpu` tests
// remain the parity tests proper. Nothing in the product includes or links this directory.
#pragma once
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HS_DYNAMIC_LDS(name) double* name = hs_emul::dynamic_lds()
#define HS_EMULATED_DEVICE 1     // (kernel sources that spell an instruction in inline assembly keep a C++ statement of it behind this)
#define HS_DEVICE_PRIMITIVES_HPP  // the include guard of csrc/device_primitives.hpp: HS_DYNAMIC_LDS, lds_barrier, wait_lds, wait_vmem are this header's

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct double2 {
  double x, y;
};
inline double2 make_double2(double a, double b) { return double2{a, b}; }
struct int4 {
  int x, y, z, w;
};

namespace hs_emul {

/// Barrier over the threads that have not left the kernel yet (a wave that returned early does not take part on the GPU either).
class LiveBarrier {
 public:
  void reset(int n) {
    std::lock_guard<std::mutex> lk(m_);
    live_ = n, waiting_ = 0, phase_.store(0);
  }
  // The counters change under a mutex (an arrival and a departure must not miss each other); the sleepers wait on the phase word through the
  // futex behind std::atomic::wait and resume without the mutex (with a condition variable every wake-up of 255 threads queued for the mutex
  // again: most of the harness's run time).
  void arrive_and_wait() {
    unsigned ph;
    bool last;
    {
      std::lock_guard<std::mutex> lk(m_);
      ph = phase_.load(std::memory_order_relaxed);
      last = ++waiting_ == live_;
      if (last) waiting_ = 0, phase_.store(ph + 1, std::memory_order_release);
    }
    if (last)
      phase_.notify_all();
    else
      while (phase_.load(std::memory_order_acquire) == ph) phase_.wait(ph, std::memory_order_acquire);
  }
  void leave() {
    bool released = false;
    {
      std::lock_guard<std::mutex> lk(m_);
      --live_;
      if (live_ > 0 && waiting_ == live_) waiting_ = 0, phase_.store(phase_.load(std::memory_order_relaxed) + 1, std::memory_order_release), released = true;
    }
    if (released) phase_.notify_all();
  }

 private:
  std::mutex m_;
  int live_ = 0, waiting_ = 0;
  std::atomic<unsigned> phase_{0};
};

struct Block {
  LiveBarrier barrier;
  LiveBarrier wave_barrier[16];
  unsigned long long wave_bits[2][16][64];  // value exchange of the wave intrinsics (doubles travel as bits); two buffers used in turn, so
                                            // that one barrier per exchange is enough (a lane can be at most one exchange ahead of another)
  std::vector<double> lds;
};
inline Block& block() {
  static Block b;
  return b;
}
inline double* dynamic_lds() { return block().lds.data(); }

}  // namespace hs_emul

extern thread_local dim3 threadIdx;
extern dim3 blockIdx, blockDim, gridDim;

inline void __syncthreads() { hs_emul::block().barrier.arrive_and_wait(); }

namespace hs_emul {
extern thread_local unsigned exchange_count;  // exchanges this lane has taken part in within the current workgroup (selects the buffer)
template <class T>
inline T wave_exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "value exchange through 64-bit slots");
  Block& b = block();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned buf = exchange_count++ & 1u;
  unsigned long long bits = 0;
  std::memcpy(&bits, &v, sizeof(T));
  b.wave_bits[buf][wave][lane] = bits;
  b.wave_barrier[wave].arrive_and_wait();
  const unsigned long long got = b.wave_bits[buf][wave][src_lane & 63];
  T out;
  std::memcpy(&out, &got, sizeof(T));
  return out;
}
}  // namespace hs_emul

template <class T>
inline T __shfl_xor(T v, int mask) { return hs_emul::wave_exchange(v, (threadIdx.x & 63) ^ mask); }
template <class T>
inline T __shfl_up(T v, unsigned delta) {
  const int lane = threadIdx.x & 63;
  const T got = hs_emul::wave_exchange(v, lane >= int(delta) ? lane - int(delta) : lane);
  return got;
}
inline int __builtin_amdgcn_readlane(int v, int lane) { return hs_emul::wave_exchange(v, lane); }
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // (the kernels use it on wave-uniform values only: the wave index as a scalar)
inline void __builtin_amdgcn_s_setprio(int) {}  // (issue priority of a wave: nothing to emulate)
/// DPP move as the kernels use it (row_mask = bank_mask = 0xF, bound_ctrl off: a lane whose source is outside its row of 16 keeps `old`):
/// quad_perm (ctrl < 0x100, two bits per lane of the quad), row_shl:n (0x100 + n: lane i reads lane i + n), row_shr:n (0x110 + n: lane i - n),
/// row_ror:n (0x120 + n: lane i reads lane (i - n) mod 16 of its row).
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) {
  const int lane = threadIdx.x & 63;
  int from = -1;
  if (ctrl < 0x100) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl > 0x100 && ctrl < 0x110) from = ((lane & 15) + (ctrl - 0x100) < 16) ? lane + (ctrl - 0x100) : -1;
  else if (ctrl > 0x110 && ctrl < 0x120) from = ((lane & 15) - (ctrl - 0x110) >= 0) ? lane - (ctrl - 0x110) : -1;
  else if (ctrl > 0x120 && ctrl < 0x130) from = (lane & ~15) | (((lane & 15) - (ctrl - 0x120)) & 15);  // row_ror:n (rotation inside the row of 16)
  const int got = hs_emul::wave_exchange(src, from < 0 ? lane : from);
  return from < 0 ? old : got;
}
/// v_mfma_f64_16x16x4_f64 as the kernels use it (tools/microbench/mfma_probe.hip): lane l supplies A[i = l & 15][k = l >> 4] and
/// B[k = l >> 4][j = l & 15]; register r of lane l holds D[(l >> 4) + 4 r][l & 15].
namespace hs_emul {
inline void wave_allgather(double v, double (&out)[64]) {
  Block& b = block();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned buf = exchange_count++ & 1u;
  unsigned long long bits = 0;
  std::memcpy(&bits, &v, 8);
  b.wave_bits[buf][wave][lane] = bits;
  b.wave_barrier[wave].arrive_and_wait();
  for (int l = 0; l < 64; ++l) std::memcpy(&out[l], &b.wave_bits[buf][wave][l], 8);
}
}  // namespace hs_emul
typedef double hs_emul_f64x4 __attribute__((vector_size(32)));
inline hs_emul_f64x4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, hs_emul_f64x4 c, int, int, int) {
  double A[64], B[64];
  hs_emul::wave_allgather(a, A);
  hs_emul::wave_allgather(b, B);
  const int lane = threadIdx.x & 63, j = lane & 15, g4 = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    double acc = c[r];
    for (int k = 0; k < 4; ++k) acc = std::fma(A[16 * k + g4 + 4 * r], B[16 * k + j], acc);
    c[r] = acc;
  }
  return c;
}
inline int __double2loint(double d) { unsigned long long b; std::memcpy(&b, &d, 8); return int(unsigned(b)); }
inline int __double2hiint(double d) { unsigned long long b; std::memcpy(&b, &d, 8); return int(unsigned(b >> 32)); }
inline double __hiloint2double(int hi, int lo) { const unsigned long long b = (static_cast<unsigned long long>(unsigned(hi)) << 32) | unsigned(lo); double d; std::memcpy(&d, &b, 8); return d; }
inline unsigned long long __ballot(bool pred) {
  hs_emul::Block& b = hs_emul::block();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned buf = hs_emul::exchange_count++ & 1u;
  b.wave_bits[buf][wave][lane] = pred ? 1ull : 0ull;
  b.wave_barrier[wave].arrive_and_wait();
  unsigned long long m = 0;
  const int n_lanes = int(blockDim.x) - 64 * wave < 64 ? int(blockDim.x) - 64 * wave : 64;
  for (int l = 0; l < n_lanes; ++l) m |= b.wave_bits[buf][wave][l] << l;
  return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline void __builtin_amdgcn_wave_barrier() {}
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_sleep(int) { std::this_thread::yield(); }
/// The LDS-only workgroup barrier and the wave-level waits of kernels_common.hpp. On the GPU the lanes of a wave run in lock step, so
/// "this wave's LDS operations have completed" is all an LDS hand-over between lanes of ONE wave needs; here the lanes are threads, and the
/// same call sites need a barrier over the wave (every live lane of the wave reaches them: they sit in wave-uniform control flow).
inline void lds_barrier() { __syncthreads(); }
inline void wait_lds() { hs_emul::block().wave_barrier[threadIdx.x >> 6].arrive_and_wait(); }
inline void wait_vmem() {}
template <int N>
inline void wait_vmem_all_but() {}
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __builtin_nontemporal_load(p) (*(p))
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
namespace hs_emul {
template <class T>
inline T atomic_load(const T* p, int order) {
  T v;
  __atomic_load(p, &v, order);
  return v;
}
template <class T, class V>
inline void atomic_store(T* p, V value, int order) {
  const T v = T(value);
  __atomic_store(p, &v, order);
}
}  // namespace hs_emul
#define __hip_atomic_load(ptr, order, scope) hs_emul::atomic_load(ptr, order)
#define __hip_atomic_store(ptr, value, order, scope) hs_emul::atomic_store(ptr, value, order)
inline double __builtin_amdgcn_rsq(double d) { return 1.0 / std::sqrt(d); }
inline double __builtin_amdgcn_rcp(double d) { return 1.0 / d; }
inline long long wall_clock64() { return 0; }
inline unsigned __builtin_amdgcn_s_getreg(int) { return 0; }
inline double rsqrt(double d) { return 1.0 / std::sqrt(d); }

template <class T>
inline T min(T a, T b) { return b < a ? b : a; }
template <class T>
inline T max(T a, T b) { return a < b ? b : a; }
using std::atan2;
using std::fabs;
using std::fma;
using std::fmax;
using std::fmin;
using std::isfinite;
using std::sqrt;

namespace hs_emul {

/// Runs `kernel` for every workgroup of the grid, one after the other; one thread per lane. `order` (optional): the blockIdx.x values in the
/// order they are run — a kernel whose workgroups wait for flags of other workgroups (the two-ended factorisation, the sweeps and their
/// inverse builders) needs its producers to run first; on the GPU they are resident together.
inline void launch(dim3 grid, dim3 block_dim, size_t lds_bytes, const std::function<void()>& kernel, const std::vector<unsigned>& order = {}) {
  Block& b = block();
  b.lds.assign(lds_bytes / 8 + 64, 0.0);
  gridDim = grid, blockDim = block_dim;
  const int n = int(block_dim.x);
  // the lane threads live for the whole launch and walk the workgroups together (creating 256 threads per workgroup dominated the run time)
  static LiveBarrier frame;
  frame.reset(n);
  std::vector<std::thread> threads;
  threads.reserve(n);
  for (int t = 0; t < n; ++t)
    threads.emplace_back([&, t] {
      for (unsigned bz = 0; bz < grid.z; ++bz)
      for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned ix = 0; ix < grid.x; ++ix) {
          frame.arrive_and_wait();  // every lane has left the previous workgroup
          if (t == 0) {
            blockIdx = dim3(order.empty() ? ix : order[ix], by, bz);
            b.barrier.reset(n);
            for (int w = 0; w < (n + 63) / 64; ++w) b.wave_barrier[w].reset(n - 64 * w < 64 ? n - 64 * w : 64);
          }
          frame.arrive_and_wait();
          threadIdx = dim3(unsigned(t), 0, 0);
          exchange_count = 0;
          kernel();
          b.wave_barrier[t >> 6].leave();
          b.barrier.leave();
        }
    });
  for (std::thread& th : threads) th.join();
}

}  // namespace hs_emul
