// Probe: cost of the redundant 6x6 register Cholesky (+ one column solve) on one wave, in cycles. hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#define UIDX(a, c) ((a) * 6 - (a) * ((a)-1) / 2 + ((c) - (a)))
template <int VARIANT>
__global__ void probe(double* out, long long* t, int n) {
  __shared__ double lds[64];
  if (threadIdx.x < 36) {
    const int a = threadIdx.x / 6, c = threadIdx.x % 6;
    lds[threadIdx.x] = (a == c ? 8.0 : 0.0) + 0.3 / (1 + a + c);
  }
  __syncthreads();
  double carry = 0.0, acc = 0.0;
  long long c0 = clock64();
  for (int it = 0; it < n; ++it) {
    double U[21], inv[6];
    {
      int p = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = a; c < 6; ++c) U[p++] = lds[6 * a + c] + carry;  // carry makes iterations dependent
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double d = U[UIDX(a, a)];
#pragma unroll
      for (int k = 0; k < a; ++k) d = fma(-U[UIDX(k, a)], U[UIDX(k, a)], d);
      double rs;
      if (VARIANT == 0) {
        rs = __builtin_amdgcn_rsq(d);
        rs = rs * fma(-0.5 * d * rs, rs, 1.5);
        rs = rs * fma(-0.5 * d * rs, rs, 1.5);
      } else if (VARIANT == 1) {
        rs = 1.0 / sqrt(d);
      } else {  // rsq + one fused Newton step in residual form
        const double y = __builtin_amdgcn_rsq(d);
        const double e = fma(-d * y, y, 1.0);
        rs = fma(y * e, fma(0.375, e, 0.5), y);
      }
      inv[a] = rs;
      U[UIDX(a, a)] = d * rs;
#pragma unroll
      for (int c = a + 1; c < 6; ++c) {
        double v = U[UIDX(a, c)];
#pragma unroll
        for (int k = 0; k < a; ++k) v = fma(-U[UIDX(k, a)], U[UIDX(k, c)], v);
        U[UIDX(a, c)] = v * rs;
      }
    }
    double x[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double v = lds[a + threadIdx.x % 6];
#pragma unroll
      for (int k = 0; k < a; ++k) v = fma(-U[UIDX(k, a)], x[k], v);
      x[a] = v * inv[a];
    }
    carry = x[5] * 1e-30;
    acc += x[0] + x[3];
  }
  long long c1 = clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) t[0] = c1 - c0;
}
int main() {
  double* out; long long* t;
  (void)hipMalloc(&out, 4096); (void)hipMallocManaged(&t, 64);
  const int n = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    probe<0><<<1, 64>>>(out, t, n); (void)hipDeviceSynchronize(); printf("rsq + 2 Newton      : %.1f clk per factor+solve\n", double(t[0]) / n);
    probe<1><<<1, 64>>>(out, t, n); (void)hipDeviceSynchronize(); printf("1.0 / sqrt(d)       : %.1f clk\n", double(t[0]) / n);
    probe<2><<<1, 64>>>(out, t, n); (void)hipDeviceSynchronize(); printf("rsq + 1 cubic step  : %.1f clk\n", double(t[0]) / n);
  }
  return 0;
}
