// Probe: do v_rsq_f64_dpp and v_fmac_f64_dpp with a negated source behave as their non-DPP forms + a row broadcast on gfx950?
// hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_probe tools/microbench/dpp_f64_probe.hip && /tmp/dpp_probe
// MI355X, ROCm 7.2: v_rsq_f64_dpp returns +inf in every lane (the DPP operand reads as zero); v_mov_b64_dpp exact; v_fmac_f64_dpp with a negated
// second source, also with both sources the same register: correct to rounding.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void probe(const double* in, double* out) {
  const int l = threadIdx.x;
  const double v = in[l], m = in[64 + l];
  double y, acc = 1.0, acc2 = 1.0, mv;
  asm volatile("s_nop 1\n\tv_rsq_f64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "=v"(y) : "v"(v));
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "=v"(mv) : "v"(v));
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(m));
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc2) : "v"(v));
  out[l] = y, out[64 + l] = mv, out[128 + l] = acc, out[192 + l] = acc2;
}
int main() {
  double h[128], o[256], *di, *dout;
  for (int i = 0; i < 64; ++i) h[i] = 1.5 + i * 0.37, h[64 + i] = 0.25 + 0.01 * i;
  hipMalloc(&di, sizeof h), hipMalloc(&dout, sizeof o);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(di, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  for (int l = 0; l < 64; ++l) {
    const double b = h[(l & ~15) | 5];
    e0 = fmax(e0, fabs(o[l] * sqrt(b) - 1.0)), e1 = fmax(e1, fabs(o[64 + l] - b));
    e2 = fmax(e2, fabs(o[128 + l] - (1.0 - b * h[64 + l]))), e3 = fmax(e3, fabs(o[192 + l] - (1.0 - b * h[l])));
  }
  printf("rsq_dpp rel err %.3e (estimate: ~1e-8 expected)  mov_dpp %.3e  fmac_dpp(-src1) %.3e  fmac_dpp(same reg, -src1) %.3e\n", e0, e1, e2, e3);
  printf("lane 0: rsq %.17g  want %.17g ; lane 20: %.17g want %.17g\n", o[0], 1 / sqrt(h[5]), o[20], 1 / sqrt(h[21]));
  return 0;
}
