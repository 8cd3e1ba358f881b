// Normalisation + ReLU fused for the encoders that feed the hot path (pytorch/core/extractor.py:6-57, 118-192: every
// 3x3 convolution is followed by InstanceNorm2d (feature net) or a frozen BatchNorm2d (context net) and a ReLU).
// The convolutions themselves stay MIOpen (BASELINE.json north_star); these kernels replace the
// batch_norm_collect_statistics / transform_input / clamp forward chain and the threshold / batch_norm_backward
// chain with two passes over the data each.  Tensors are NCHW: one workgroup owns one (n, c) plane.
#include "common.hpp"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 256 threads of a workgroup, result on every thread
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// Mean and biased variance of a plane, two passes (the second over x - mean), so there is no sum-of-squares cancellation.
__device__ __forceinline__ void plane_stats(const float* __restrict__ x, int HW, float* red, float& mean, float& var) {
  const bool vec = (HW & 3) == 0 && ((uintptr_t)x & 15) == 0;
  float s = 0.f;
  if (vec) {
    for (int i = threadIdx.x; i < (HW >> 2); i += 256) { const f32x4 v = reinterpret_cast<const f32x4*>(x)[i]; s += (v[0] + v[1]) + (v[2] + v[3]); }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) s += x[i];
  }
  mean = block_sum(s, red) / (float)HW;
  float q = 0.f;
  if (vec) {
    for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
      const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = v[k] - mean; q += d * d; }
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) { const float d = x[i] - mean; q += d * d; }
  }
  var = block_sum(q, red) / (float)HW;
}

// y = relu?((x - mean) * rstd); stats[plane] = (mean, rstd)
__global__ __launch_bounds__(256) void inorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ stats, int HW, float eps, int relu) {
  __shared__ float red[4];
  const float* xp = x + (int64_t)blockIdx.x * HW;
  float* yp = y + (int64_t)blockIdx.x * HW;
  float mean, var;
  plane_stats(xp, HW, red, mean, var);
  const float rstd = rsqrtf(var + eps);
  if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
  const bool vec = (HW & 3) == 0 && (((uintptr_t)xp | (uintptr_t)yp) & 15) == 0;
  if (vec) {
    for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
      f32x4 v = reinterpret_cast<const f32x4*>(xp)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = (v[k] - mean) * rstd; if (relu) v[k] = fmaxf(v[k], 0.f); }
      reinterpret_cast<f32x4*>(yp)[i] = v;
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) { float v = (xp[i] - mean) * rstd; yp[i] = relu ? fmaxf(v, 0.f) : v; }
  }
}

// g' = g * (xhat > 0) if relu;  dx = rstd * (g' - mean(g') - xhat * mean(g' * xhat))
__global__ __launch_bounds__(256) void inorm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                        const float* __restrict__ stats, float* __restrict__ dx, int HW,
                                                        int relu) {
  __shared__ float red[4];
  const float* gp = g + (int64_t)blockIdx.x * HW;
  const float* xp = x + (int64_t)blockIdx.x * HW;
  float* dp = dx + (int64_t)blockIdx.x * HW;
  const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
  const bool vec = (HW & 3) == 0 && (((uintptr_t)gp | (uintptr_t)xp | (uintptr_t)dp) & 15) == 0;
  float s1 = 0.f, s2 = 0.f;
  if (vec) {
    for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
      const f32x4 gv = reinterpret_cast<const f32x4*>(gp)[i], xv = reinterpret_cast<const f32x4*>(xp)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (xv[k] - mean) * rstd;
        const float gg = (relu && xh <= 0.f) ? 0.f : gv[k];
        s1 += gg; s2 += gg * xh;
      }
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float xh = (xp[i] - mean) * rstd;
      const float gg = (relu && xh <= 0.f) ? 0.f : gp[i];
      s1 += gg; s2 += gg * xh;
    }
  }
  s1 = block_sum(s1, red) / (float)HW;
  s2 = block_sum(s2, red) / (float)HW;
  if (vec) {
    for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
      const f32x4 gv = reinterpret_cast<const f32x4*>(gp)[i], xv = reinterpret_cast<const f32x4*>(xp)[i];
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (xv[k] - mean) * rstd;
        const float gg = (relu && xh <= 0.f) ? 0.f : gv[k];
        o[k] = rstd * (gg - s1 - xh * s2);
      }
      reinterpret_cast<f32x4*>(dp)[i] = o;
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float xh = (xp[i] - mean) * rstd;
      const float gg = (relu && xh <= 0.f) ? 0.f : gp[i];
      dp[i] = rstd * (gg - s1 - xh * s2);
    }
  }
}

// Frozen (eval-mode) BatchNorm + ReLU: y = relu?(x * scale[c] + shift[c]), scale = w * rsqrt(rv + eps), shift = b - rm * scale
__global__ __launch_bounds__(256) void affine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y, int C,
                                                         int HW, int relu) {
  const int c = blockIdx.x % C;
  const float a = scale[c], b = shift[c];
  const float* xp = x + (int64_t)blockIdx.x * HW;
  float* yp = y + (int64_t)blockIdx.x * HW;
  const bool vec = (HW & 3) == 0 && (((uintptr_t)xp | (uintptr_t)yp) & 15) == 0;
  if (vec) {
    for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
      f32x4 v = reinterpret_cast<const f32x4*>(xp)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = v[k] * a + b; if (relu) v[k] = fmaxf(v[k], 0.f); }
      reinterpret_cast<f32x4*>(yp)[i] = v;
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) { const float v = xp[i] * a + b; yp[i] = relu ? fmaxf(v, 0.f) : v; }
  }
}

// g' = g * (y > 0) if relu; dx = g' * scale[c]; dshift[c] += sum g'; dscale_x[c] += sum g' * x   (one atomic pair per plane)
__global__ __launch_bounds__(256) void affine_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float* __restrict__ dx, float* __restrict__ dsum_g,
                                                         float* __restrict__ dsum_gx, int C, int HW, int relu) {
  __shared__ float red[4];
  const int c = blockIdx.x % C;
  const float a = scale[c], b = shift[c];
  const float* gp = g + (int64_t)blockIdx.x * HW;
  const float* xp = x + (int64_t)blockIdx.x * HW;
  float* dp = dx + (int64_t)blockIdx.x * HW;
  const bool vec = (HW & 3) == 0 && (((uintptr_t)gp | (uintptr_t)xp | (uintptr_t)dp) & 15) == 0;
  float s1 = 0.f, s2 = 0.f;
  if (vec) {
    for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
      const f32x4 gv = reinterpret_cast<const f32x4*>(gp)[i], xv = reinterpret_cast<const f32x4*>(xp)[i];
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gg = (relu && xv[k] * a + b <= 0.f) ? 0.f : gv[k];
        s1 += gg; s2 += gg * xv[k];
        o[k] = gg * a;
      }
      reinterpret_cast<f32x4*>(dp)[i] = o;
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float gg = (relu && xp[i] * a + b <= 0.f) ? 0.f : gp[i];
      s1 += gg; s2 += gg * xp[i];
      dp[i] = gg * a;
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) { atomicAdd(dsum_g + c, s1); atomicAdd(dsum_gx + c, s2); }
}

}  // namespace

extern "C" int fsraft_inorm_relu_fwd(const float* x, float* y, float* stats, int64_t planes, int HW, float eps, int relu,
                                     hipStream_t s) {
  if (!x || !y || !stats || planes < 1 || planes > 0x7fffffff || HW < 1) return FS_ERR_ARG;
  hipLaunchKernelGGL(inorm_fwd_kernel, dim3((unsigned)planes), dim3(256), 0, s, x, y, stats, HW, eps, relu);
  return fs_launch_status();
}
extern "C" int fsraft_inorm_relu_bwd(const float* g, const float* x, const float* stats, float* dx, int64_t planes, int HW,
                                     int relu, hipStream_t s) {
  if (!g || !x || !stats || !dx || planes < 1 || planes > 0x7fffffff || HW < 1) return FS_ERR_ARG;
  hipLaunchKernelGGL(inorm_bwd_kernel, dim3((unsigned)planes), dim3(256), 0, s, g, x, stats, dx, HW, relu);
  return fs_launch_status();
}
extern "C" int fsraft_affine_relu_fwd(const float* x, const float* scale, const float* shift, float* y, int64_t planes, int C,
                                      int HW, int relu, hipStream_t s) {
  if (!x || !scale || !shift || !y || planes < 1 || planes > 0x7fffffff || C < 1 || HW < 1) return FS_ERR_ARG;
  hipLaunchKernelGGL(affine_fwd_kernel, dim3((unsigned)planes), dim3(256), 0, s, x, scale, shift, y, C, HW, relu);
  return fs_launch_status();
}
extern "C" int fsraft_affine_relu_bwd(const float* g, const float* x, const float* scale, const float* shift, float* dx,
                                      float* dsum_g, float* dsum_gx, int64_t planes, int C, int HW, int relu, hipStream_t s) {
  if (!g || !x || !scale || !shift || !dx || !dsum_g || !dsum_gx || planes < 1 || planes > 0x7fffffff || C < 1 || HW < 1)
    return FS_ERR_ARG;
  hipLaunchKernelGGL(affine_bwd_kernel, dim3((unsigned)planes), dim3(256), 0, s, g, x, scale, shift, dx, dsum_g, dsum_gx, C, HW, relu);
  return fs_launch_status();
}
