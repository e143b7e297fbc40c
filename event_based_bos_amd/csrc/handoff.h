// handoff.h -- hand-offs between the workgroups of ONE launch through global memory (cdna guide Guideline 16 / MI355X_MICROARCH
// visibility table): every handed-off byte is an sc1 (write-through) store, consumers poll and read with sc1 loads; where a flag
// announces bulk data every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup barrier behind which ONE lane stores the
// flag; small payloads travel as tagged 8-byte granules {tag, 32 data bits}: the data is its own flag.  No fences, no atomics.
// Used by the resident solver kernels (cmax_resident_core.h) and the fused value kernel (iwe_value_fused.hip).
#pragma once
#include "common.h"

namespace ebos {
namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) float gf32;

__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* p) {
  return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load((gf32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store((gf32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every storing wave, before the barrier behind which the flag is stored (inline asm: invisible to the pass that drops waits)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// a double as two tagged 8-byte granules {tag, 32 bits}: the data is the flag (cdna guide, R2)
__device__ __forceinline__ void put_granules(unsigned long long* g, unsigned tag, double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  st_sc1(g, ((unsigned long long)tag << 32) | (b & 0xffffffffull));
  st_sc1(g + 1, ((unsigned long long)tag << 32) | (b >> 32));
}


// the double behind two granules
__device__ __forceinline__ double granules_double(unsigned long long g0, unsigned long long g1) {
  return __builtin_bit_cast(double, (g0 & 0xffffffffull) | (g1 << 32));
}

}  // namespace
}  // namespace ebos
