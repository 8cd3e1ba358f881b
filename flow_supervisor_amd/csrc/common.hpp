// Shared declarations for the fsraft HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "split_arith.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FS_OK 0
#define FS_ERR_ARG 1      // bad argument (null pointer, unsupported size)
#define FS_ERR_LAUNCH 2   // hipGetLastError() after launch was not hipSuccess

static inline int fs_launch_status() {
  return hipGetLastError() == hipSuccess ? FS_OK : FS_ERR_LAUNCH;
}

// Explicit global-address-space accesses.  A pointer that reaches a load through integer/select arithmetic (or out
// of a by-value argument struct) can lose its address-space inference and compile to flat_load/flat_store.  Flat
// operations count against BOTH vmcnt and lgkmcnt, so the `s_waitcnt lgkmcnt(0)` that guards an LDS hand-over then
// also waits for every global prefetch in flight -- which serialises the software pipeline of the GEMM cores.
#define FS_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ f32x4 gload4(const void* p) { return *(const FS_GLOBAL f32x4*)p; }
__device__ __forceinline__ float gload1(const float* p) { return *(const FS_GLOBAL float*)p; }
__device__ __forceinline__ void gstore4(void* p, f32x4 v) { *(FS_GLOBAL f32x4*)p = v; }
__device__ __forceinline__ void gstore1(float* p, float v) { *(FS_GLOBAL float*)p = v; }

__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float sigmoidf_dev(float x) { return 1.0f / (1.0f + __expf(-x)); }
