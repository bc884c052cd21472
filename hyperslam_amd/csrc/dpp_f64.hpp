// dpp_f64.hpp — f64 instructions the panels of the matrix-core factorisations spell themselves: multiply-adds and moves with a DPP row
// broadcast, and plain f64 instructions as ORDERED statements for the hand-scheduled pivot chains (kernels_dense_mx.hpp, kernels_factor_mx.hpp).
// Every statement has its C++ spelling behind HS_EMULATED_DEVICE (tests/emul: the kernel sources on the CPU).
#pragma once
#include "kernels_common.hpp"

namespace hs {

/// acc += (lane R of the caller's row of sixteen lanes' u) * m in ONE instruction (v_fmac_f64_dpp, DPP row_newbcast: the one DPP control the f64
/// instructions have): the multiplier of the elimination goes from the diagonal tile's lane straight into the multiply-add — with v_readlane
/// it was two scalar moves and the FMA, and the panel is instruction issue.
template <int R>
HSD void dx_fmac_bcast(double& acc, double u, double m) {
#if !defined(HS_EMULATED_DEVICE)
  // (no wait states in the statement: the caller keeps two instructions between the one that wrote u and this one)
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(m), "n"(R));
#else
  acc = fma(hs_emul::wave_exchange(u, int((threadIdx.x & 63u) & ~15u) | R), m, acc);
#endif
}

/// acc -= (lane R of the row's u) * m: the negation as a source modifier of the same instruction.
template <int R>
HSD void dx_fnma_bcast(double& acc, double u, double m) {
#if !defined(HS_EMULATED_DEVICE)
  asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(m), "n"(R));
#else
  acc = fma(hs_emul::wave_exchange(u, int((threadIdx.x & 63u) & ~15u) | R), -m, acc);
#endif
}
/// Lane R of the caller's row of sixteen lanes, for every lane of the row; two wait states in front (v was written by the multiply-add just before).
/// (v_rsq_f64 assembles with a DPP control too, which would take the broadcast out of the chain — the hardware returns infinity for it:
///  tools/microbench/dpp_f64_probe.hip. Of the f64 instructions only v_fmac_f64 and v_mov_b64 carry DPP on gfx950.)
template <int R>
HSD double dx_row_bcast(double v) {
#if !defined(HS_EMULATED_DEVICE)
  double out;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(v), "n"(R));
  return out;
#else
  return hs_emul::wave_exchange(v, int((threadIdx.x & 63u) & ~15u) | R);
#endif
}
/// Hardware estimate of 1 / sqrt(d) as an ordered statement (one wait state behind: a transcendental result read by a vector instruction).
HSD double dx_rsq(double d) {
#if !defined(HS_EMULATED_DEVICE)
  double y;
  asm volatile("v_rsq_f64_e32 %0, %1\n\ts_nop 0" : "=v"(y) : "v"(d));
  return y;
#else
  return __builtin_amdgcn_rsq(d);
#endif
}
/// The value is needed here, whatever the code behind does with it.
HSD void dx_pin(double& v) {
#if !defined(HS_EMULATED_DEVICE)
  asm volatile("" : "+v"(v));
#endif
}
/// Plain f64 instructions as ordered statements: the panel below is scheduled BY HAND (the order of the volatile statements is the order of
/// issue), the compiler only allocates registers.
HSD double dx_mul(double a, double b) {
#if !defined(HS_EMULATED_DEVICE)
  double o;
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
  return o;
#else
  return a * b;
#endif
}
HSD double dx_fma(double a, double b, double c) {
#if !defined(HS_EMULATED_DEVICE)
  double o;
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
  return o;
#else
  return fma(a, b, c);
#endif
}
HSD double dx_one_minus(double a, double b) {  // 1 - a b
#if !defined(HS_EMULATED_DEVICE)
  double o;
  asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(o) : "v"(a), "v"(b));
  return o;
#else
  return fma(-a, b, 1.0);
#endif
}
HSD double dx_half_plus(double a, double b) {  // 0.5 + a b
#if !defined(HS_EMULATED_DEVICE)
  double o;
  asm volatile("v_fma_f64 %0, %1, %2, 0.5" : "=v"(o) : "v"(a), "v"(b));
  return o;
#else
  return fma(a, b, 0.5);
#endif
}
/// Pivot row P is final: scale it (two wait states behind the first product: the multiply-adds that follow read it through DPP).
HSD void dx_scale2(double& a, double& b, double r) {
#if !defined(HS_EMULATED_DEVICE)
  asm volatile("v_mul_f64 %0, %0, %2\n\tv_mul_f64 %1, %1, %2\n\ts_nop 0" : "+v"(a), "+v"(b) : "v"(r));
#else
  a *= r, b *= r;
#endif
}

}  // namespace hs
