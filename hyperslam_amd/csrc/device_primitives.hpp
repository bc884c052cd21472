// device_primitives.hpp — the few spellings the kernels use for what only the hardware has: dynamic LDS and the wait / barrier instructions
// (gfx950). Classic include guard on purpose: a translation unit that brings its own definitions of these names defines the guard
// first — nothing else in the product's sources knows about such a unit.
#ifndef HS_DEVICE_PRIMITIVES_HPP
#define HS_DEVICE_PRIMITIVES_HPP
#include "device_math.hpp"

/// Dynamic LDS of a kernel (sized at launch). One spelling for every kernel.
#define HS_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) double name[]

namespace hs {
/// Workgroup barrier that only drains LDS traffic: global loads / stores stay in flight across it (the factorisation
/// prefetches the next band row while the current step runs; __syncthreads() would wait for vmcnt(0) every step).
/// wait_lds / wait_vmem: this wave's LDS / global-memory operations have completed (one wave's LDS traffic is in order: between lanes of a
/// wave this is all the synchronisation an LDS hand-over needs).
HSD void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
HSD void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
HSD void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
/// All but the N most recently issued global-memory operations of this wave have completed (they complete in issue order).
template <int N>
HSD void wait_vmem_all_but() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
}  // namespace hs
#endif
