// Probe: cost of an LDS-only workgroup barrier (s_waitcnt lgkmcnt(0); s_barrier) for 4 / 6 / 8 waves. hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long* t, int n, int skew) {
  __shared__ double lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  double a = threadIdx.x;
  long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
    if (wave == (i & 3) && skew)  // one wave does some dependent work before arriving
      for (int k = 0; k < skew; ++k) a = fma(a, 1.0000001, 1e-9);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  long long c1 = clock64();
  lds[threadIdx.x] = a;
  if (threadIdx.x == 0) t[0] = c1 - c0, t[1] = (long long)lds[5];
}
int main() {
  long long* t;
  (void)hipMallocManaged(&t, 64);
  const int n = 10000;
  for (int threads : {256, 384, 512})
    for (int skew : {0, 64}) {
      for (int rep = 0; rep < 2; ++rep) { probe<<<1, threads>>>(t, n, skew); (void)hipDeviceSynchronize(); }
      printf("threads %d skew %d dependent FMAs: %.1f clk per barrier iteration\n", threads, skew, double(t[0]) / n);
    }
  return 0;
}
