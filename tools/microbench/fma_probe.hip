// Probe: issue cost of fp64 FMA forms on one wave: acc += a*b (v_fmac_f64, VOP2) vs acc = fma(-a, b, acc) (v_fma_f64, VOP3 with modifier).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void probe(double* out, long long* t, int n) {
  double acc[36], xa[6], xc[6];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = e + threadIdx.x;
#pragma unroll
  for (int e = 0; e < 6; ++e) xa[e] = 1e-3 * (e + threadIdx.x), xc[e] = 1e-4 * (e + 2 * threadIdx.x);
  long long c0 = clock64();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          if (MODE == 0) acc[6 * r + c] = fma(xa[r], xc[c], acc[6 * r + c]);
          if (MODE == 1) acc[6 * r + c] = fma(-xa[r], xc[c], acc[6 * r + c]);
          if (MODE == 2) acc[6 * r + c] = acc[6 * r + c] - xa[r] * xc[c];
        }
      asm volatile("" : "+v"(xa[0]), "+v"(xc[0]));
    }
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
  const int n = 1000;
  for (int threads : {64, 128, 256}) {
    for (int rep = 0; rep < 2; ++rep) { probe<0><<<1, threads>>>(out, t, n); (void)hipDeviceSynchronize(); }
    printf("threads %3d  acc += a*b      : %.2f clk per FMA\n", threads, double(t[0]) / n / 216);
    for (int rep = 0; rep < 2; ++rep) { probe<1><<<1, threads>>>(out, t, n); (void)hipDeviceSynchronize(); }
    printf("threads %3d  acc -= a*b (neg): %.2f clk per FMA\n", threads, double(t[0]) / n / 216);
  }
  return 0;
}
