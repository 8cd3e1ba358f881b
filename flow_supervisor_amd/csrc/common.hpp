// Shared declarations for the fsraft HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FS_OK 0
#define FS_ERR_ARG 1      // bad argument (null pointer, unsupported size)
#define FS_ERR_LAUNCH 2   // hipGetLastError() after launch was not hipSuccess

static inline int fs_launch_status() {
  return hipGetLastError() == hipSuccess ? FS_OK : FS_ERR_LAUNCH;
}

__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float sigmoidf_dev(float x) { return 1.0f / (1.0f + __expf(-x)); }
