// Probe: shader clock under a single-workgroup load, dependent / independent fp64 FMA cost, LDS read cost. Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out, long long* t, int n) {
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0000001, c = 1e-9;
  long long w0 = wall_clock64(), c0 = clock64();
#pragma unroll 32
  for (int i = 0; i < n; ++i) a = fma(a, b, c);  // dependent chain
  long long w1 = wall_clock64(), c1 = clock64();
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
#pragma unroll 8
  for (int i = 0; i < n / 8; ++i) {
    x0 = fma(x0, b, c), x1 = fma(x1, b, c), x2 = fma(x2, b, c), x3 = fma(x3, b, c);
    x4 = fma(x4, b, c), x5 = fma(x5, b, c), x6 = fma(x6, b, c), x7 = fma(x7, b, c);
  }
  long long w2 = wall_clock64(), c2 = clock64();
  double s = 0;
  int idx = threadIdx.x;
#pragma unroll 16
  for (int i = 0; i < n; ++i) {  // dependent LDS reads (address depends on the value read)
    double v = lds[idx & 4095];
    idx = idx + 7 + (v > 2.0 ? 1 : 0);
    s += v;
  }
  long long w3 = wall_clock64(), c3 = clock64();
  double r = a;
#pragma unroll 16
  for (int i = 0; i < n / 4; ++i) r = __builtin_amdgcn_rsq(r + 1.5);
  long long w4 = wall_clock64(), c4 = clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + s + r;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    t[0] = w1 - w0, t[1] = c1 - c0, t[2] = w2 - w1, t[3] = c2 - c1, t[4] = w3 - w2, t[5] = c3 - c2, t[6] = w4 - w3, t[7] = c4 - c3;
  }
}
int main() {
  double* out; long long* t;
  hipMalloc(&out, 8 * 1024 * 1024); hipMallocManaged(&t, 64);
  const int n = 20000;
  for (int blocks : {1, 1, 1024}) for (int threads : {64, 256}) {
    for (int rep = 0; rep < 3; ++rep) { probe<<<blocks, threads>>>(out, t, n); hipDeviceSynchronize(); }
    printf("blocks %4d threads %3d | dep fma: %.2f ns, %.2f clk | indep fma: %.2f ns, %.2f clk | dep LDS read: %.1f ns, %.1f clk | dep rsq: %.1f ns %.1f clk | clk/wall(10ns) = %.2f => %.2f GHz\n",
           blocks, threads, 10.0 * t[0] / n, double(t[1]) / n, 10.0 * t[2] / n, double(t[3]) / n, 10.0 * t[4] / n, double(t[5]) / n, 10.0 * t[6] / (n / 4), double(t[7]) / (n / 4),
           double(t[1]) / t[0], double(t[1]) / t[0] / 10.0);
  }
  return 0;
}
