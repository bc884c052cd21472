// Probe: hand-off latency from a late wave to waiting waves: s_barrier vs polling an LDS word.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>  // 0: s_barrier, 1: LDS polling (one flag word, monotonic)
__global__ void probe(long long* t, double* out, int n) {
  __shared__ double lds[64];
  __shared__ unsigned flag;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  double a = threadIdx.x;
  long long sum = 0;
  for (int i = 1; i <= n; ++i) {
    if (wave == 0) {  // producer: ~1500 clk of dependent work, an LDS write, then the hand-off
#pragma unroll
      for (int k = 0; k < 256; ++k) a = fma(a, 1.0000001, 1e-9);
      lds[lane] = a;
      const long long t_sig = wall_clock64();
      if (lane == 0) t[2 + (i & 1)] = t_sig;
      if (MODE == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *reinterpret_cast<volatile unsigned*>(&flag) = unsigned(i);
      }
    }
    if (MODE == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (wave != 0) {
      while (*reinterpret_cast<volatile unsigned*>(&flag) < unsigned(i)) {
      }
    }
    const long long t_exit = wall_clock64();
    if (wave == 1 && lane == 0) t[4 + (i & 1)] = t_exit;
    a += lds[lane] * 1e-30;
    // consumers -> producer direction always through the barrier (keeps the iterations in lockstep for the measurement)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (threadIdx.x == 64) sum += t[4 + (i & 1)] - t[2 + (i & 1)];
  }
  out[threadIdx.x] = a;
  if (threadIdx.x == 64) t[0] = sum;
}
int main() {
  long long* t; double* out;
  (void)hipMallocManaged(&t, 64 * 8); (void)hipMalloc(&out, 8192);
  const int n = 2000;
  for (int threads : {256, 384}) {
    for (int rep = 0; rep < 2; ++rep) { probe<0><<<1, threads>>>(t, out, n); (void)hipDeviceSynchronize(); }
    printf("threads %d  s_barrier hand-off : %.0f ns\n", threads, 10.0 * double(t[0]) / n);
    for (int rep = 0; rep < 2; ++rep) { probe<1><<<1, threads>>>(t, out, n); (void)hipDeviceSynchronize(); }
    printf("threads %d  LDS polling hand-off: %.0f ns\n", threads, 10.0 * double(t[0]) / n);
  }
  return 0;
}
