// Global-address-space views of the plain pointers the work-item records carry.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace pnmn {

// The records hand the kernels plain pointers that were themselves loaded from memory, so the
// compiler cannot prove they are global addresses and emits FLAT loads -- which also count on the
// LDS counter (lgkmcnt): every wait for an LDS fragment then waits for the weight fetch from L2 issued
// just before it, and the software pipeline of the contraction loop collapses.  Everything the
// records point to is device memory: say so.
using gfloat = __attribute__((address_space(1))) float;
using gf32x4 = __attribute__((address_space(1))) f32x4;
__device__ __forceinline__ const gfloat* as_global(const float* p) { return (const gfloat*)p; }
__device__ __forceinline__ gfloat* as_global(float* p) { return (gfloat*)p; }
__device__ __forceinline__ f32x4 load4(const gfloat* p) { return *(const gf32x4*)p; }
__device__ __forceinline__ void store4(gfloat* p, f32x4 v) { *(gf32x4*)p = v; }

}  // namespace pnmn
