// Fused sequence loss (pytorch/train.py:60-96): for predictions p_i [B,2,H,W], ground truth gt, validity mask,
//   loss = sum_i w_i * mean_{b,c,y,x}( mask[b,y,x] * sqrt((p_i - gt)^2 + eps^2) ),   mask = valid >= 0.5 && |gt| < max_flow
// in one pass over all predictions, and (optionally, in the same pass) the gradients
//   dp_i = w_i / numel * mask * (p_i - gt) / sqrt((p_i - gt)^2 + eps^2).
// Also the end-point-error statistics of one chosen prediction (epe sum, counts < 1 / 3 / 5 px, valid count).
#include "common.hpp"
#include <cstddef>

namespace {

constexpr int LOSS_MAX_PRED = 32;
struct LossArgs {
  const float* pred[LOSS_MAX_PRED];
  float* dpred[LOSS_MAX_PRED];          // nullptr: no gradient for that prediction
  float w[LOSS_MAX_PRED];
  int n, metric_idx;
  const float* gt;                      // [B,2,H,W] or nullptr (zero flow)
  const float* valid;                   // [B,H,W] or nullptr (all valid)
  float max_flow, eps2;
  int B, HW;
  float* out;                           // [0] loss, [1] epe sum, [2] n(<1px), [3] n(<3px), [4] n(<5px), [5] n valid(>0.5)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void sequence_loss_kernel(LossArgs a) {
  __shared__ float red[4][6];
  const auto* karg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  typedef const float* cfptr;
  typedef float* fptr;
  const int64_t npix = (int64_t)a.B * a.HW;
  const float inv = 1.0f / (float)(npix * 2);
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < npix; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / a.HW), pix = (int)(e % a.HW);
    const int64_t i0 = ((int64_t)b * 2) * a.HW + pix, i1 = i0 + a.HW;
    const float g0 = a.gt ? a.gt[i0] : 0.f, g1 = a.gt ? a.gt[i1] : 0.f;
    const float v = a.valid ? a.valid[e] : 1.f;
    const bool m = v >= 0.5f && sqrtf(g0 * g0 + g1 * g1) < a.max_flow;
    for (int i = 0; i < a.n; ++i) {                                   // uniform trip count, pointer tables via s_load
      const float* p = ((const cfptr __attribute__((address_space(4)))*)(karg + offsetof(LossArgs, pred)))[i];
      float* dp = ((const fptr __attribute__((address_space(4)))*)(karg + offsetof(LossArgs, dpred)))[i];
      const float wi = ((const float __attribute__((address_space(4)))*)(karg + offsetof(LossArgs, w)))[i];
      const float d0 = p[i0] - g0, d1 = p[i1] - g1;
      const float s0 = sqrtf(d0 * d0 + a.eps2), s1 = sqrtf(d1 * d1 + a.eps2);
      if (m) acc[0] += wi * (s0 + s1);
      if (dp) {
        dp[i0] = m ? wi * inv * d0 / s0 : 0.f;
        dp[i1] = m ? wi * inv * d1 / s1 : 0.f;
      }
      if (i == a.metric_idx && v > 0.5f) {
        const float epe = sqrtf(d0 * d0 + d1 * d1);
        acc[1] += epe; acc[2] += epe < 1.f; acc[3] += epe < 3.f; acc[4] += epe < 5.f; acc[5] += 1.f;
      }
    }
  }
  acc[0] *= inv;
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = wave_sum(acc[k]);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) red[threadIdx.x >> 6][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 6) atomicAdd(a.out + threadIdx.x, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

}  // namespace

// out[6] must be zeroed by the caller.  dpred[i] may be NULL.  preds / gt are [B,2,H,W] contiguous, valid [B,H,W].
extern "C" int fsraft_sequence_loss(const float* const* pred, float* const* dpred, const float* weights, int n, int metric_idx,
                                    const float* gt, const float* valid, float max_flow, float eps, int B, int H, int W,
                                    float* out, hipStream_t s) {
  if (!pred || !weights || !out || n < 1 || n > LOSS_MAX_PRED || B < 1 || H < 1 || W < 1) return FS_ERR_ARG;
  LossArgs a{};
  for (int i = 0; i < n; ++i) {
    if (!pred[i]) return FS_ERR_ARG;
    a.pred[i] = pred[i]; a.dpred[i] = dpred ? dpred[i] : nullptr; a.w[i] = weights[i];
  }
  a.n = n; a.metric_idx = metric_idx; a.gt = gt; a.valid = valid; a.max_flow = max_flow; a.eps2 = eps * eps;
  a.B = B; a.HW = H * W; a.out = out;
  const int64_t npix = (int64_t)B * H * W;
  int blocks = (int)((npix + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sequence_loss_kernel, dim3(blocks), dim3(256), 0, s, a);
  return fs_launch_status();
}
