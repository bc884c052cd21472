// Probe: cycles of one rank-6 tile update (36 ds_read_b128 + 216 fp64 FMA per lane) on 1..4 waves of one workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void probe(double* out, long long* t, int n, int ld) {
  extern __shared__ __attribute__((aligned(16))) double xb[];
  for (int i = threadIdx.x; i < 6 * ld; i += blockDim.x) xb[i] = 1e-3 * (i % 17);
  __syncthreads();
  double acc[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = e;
  const int lane = threadIdx.x & 63;
  const int ca = 6 * (1 + lane % 5), cb = 6 * (1 + lane % 7);
  long long c0 = clock64();
  for (int it = 0; it < n; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double xa[6], xc[6];
#pragma unroll
        for (int c = 0; c < 6; c += 2) {
          const double2 va = *reinterpret_cast<const double2*>(&xb[a * ld + ca + c]);
          const double2 vb = *reinterpret_cast<const double2*>(&xb[a * ld + cb + c]);
          xa[c] = va.x, xa[c + 1] = va.y, xc[c] = vb.x, xc[c + 1] = vb.y;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 0; c < 6; ++c) acc[6 * r + c] = fma(-xa[r], xc[c], acc[6 * r + c]);
      }
    } else if (MODE == 2) {  // LDS reads only: 36 ds_read_b128, results folded with 12 adds per row
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = 0; c < 6; c += 2) {
          const double2 va = *reinterpret_cast<const double2*>(&xb[a * ld + ca + c]);
          const double2 vb = *reinterpret_cast<const double2*>(&xb[a * ld + cb + c]);
          acc[6 * a + c] += va.x + vb.y, acc[6 * a + c + 1] += va.y + vb.x;
        }
      }
    } else {
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
          for (int c = 0; c < 6; ++c) acc[6 * r + c] = fma(-xa[b][r], xc[b][c], acc[6 * r + c]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("" ::: "memory");
  }
  long long c1 = clock64();
  double s = 0;
#pragma unroll
  for (int e = 0; e < 36; ++e) s += acc[e];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) t[0] = c1 - c0;
}
int main() {
  double* out; long long* t;
  (void)hipMalloc(&out, 8192); (void)hipMallocManaged(&t, 64);
  const int n = 2000, ld = 86;
  for (int threads : {64, 192, 256, 512}) {
    for (int rep = 0; rep < 2; ++rep) { probe<0><<<1, threads, 6 * ld * 8>>>(out, t, n, ld); (void)hipDeviceSynchronize(); }
    printf("threads %3d naive      : %.0f clk per tile update (216 FMA)\n", threads, double(t[0]) / n);
    for (int rep = 0; rep < 2; ++rep) { probe<1><<<1, threads, 6 * ld * 8>>>(out, t, n, ld); (void)hipDeviceSynchronize(); }
    printf("threads %3d pipelined  : %.0f clk\n", threads, double(t[0]) / n);
    for (int rep = 0; rep < 2; ++rep) { probe<2><<<1, threads, 6 * ld * 8>>>(out, t, n, ld); (void)hipDeviceSynchronize(); }
    printf("threads %3d loads only : %.0f clk (36 ds_read_b128 + 72 adds)\n", threads, double(t[0]) / n);
  }
  return 0;
}
