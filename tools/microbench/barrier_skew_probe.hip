// Probe: release latency of the LDS-only workgroup barrier when one wave arrives late (the others are parked in s_barrier).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long* t, double* out, int n) {
  __shared__ double lds[64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double a = threadIdx.x;
  long long sum = 0;
  for (int i = 0; i < n; ++i) {
    long long t_arrive = 0;
    if (wave == 0) {  // the late wave: ~1500 clk of dependent FMAs
#pragma unroll
      for (int k = 0; k < 256; ++k) a = fma(a, 1.0000001, 1e-9);
      lds[lane] = a;  // an LDS write right before the barrier, like the panel
      t_arrive = wall_clock64();
      if (lane == 0) t[2 + (i & 1)] = t_arrive;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const long long t_exit = wall_clock64();
    if (wave == 1 && lane == 0) t[4 + (i & 1)] = t_exit;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (threadIdx.x == 64) sum += t[4 + (i & 1)] - t[2 + (i & 1)];
  }
  out[threadIdx.x] = a;
  if (threadIdx.x == 64) t[0] = sum;
}
int main() {
  long long* t; double* out;
  (void)hipMallocManaged(&t, 64 * 8); (void)hipMalloc(&out, 8192);
  const int n = 2000;
  for (int threads : {128, 384, 512}) {
    for (int rep = 0; rep < 2; ++rep) { probe<<<1, threads>>>(t, out, n); (void)hipDeviceSynchronize(); }
    printf("threads %d: late arrival -> other wave's exit: %.1f ns (wall clock, 10 ns ticks)\n", threads, 10.0 * double(t[0]) / n);
  }
  return 0;
}
