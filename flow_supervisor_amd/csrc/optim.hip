// Gradient clipping + AdamW over FLAT buffers (parameters, gradients, both moments): the optimizer step of
// pytorch/train.py:137, 280-282 (optim.AdamW(..., weight_decay, eps); clip_grad_norm_(parameters, clip); optimizer.step())
// as one elementwise pass.  torch's fused multi-tensor AdamW takes 13 launches and 0.41 ms for RAFT's 5.3 M parameters in 150
// tensors (0.6 TB/s); one kernel over four flat 21 MB buffers is bound by their 150 MB.
#include "common.hpp"

namespace {

struct AdamState { float coef, step_size, inv_sqrt_bias2, decay; };

// step (fp32 count, on the device: the captured hipGraph replays it) += 1; the scalars of this step
__global__ void adamw_prepare_kernel(float* __restrict__ step, const float* __restrict__ norm, float max_norm,
                                     const float* __restrict__ lr, float beta1, float beta2, float wd, AdamState* __restrict__ st) {
  const float t = step[0] + 1.f;
  step[0] = t;
  const float bias1 = 1.f - powf(beta1, t), bias2 = 1.f - powf(beta2, t), l = lr[0];
  float coef = 1.f;
  if (norm) {                                    // torch.nn.utils.clip_grad_norm_: max_norm / (total + 1e-6), clamped to 1
    coef = max_norm / (norm[0] + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
  }
  st->coef = coef;
  st->step_size = l / bias1;
  st->inv_sqrt_bias2 = 1.f / sqrtf(bias2);
  st->decay = 1.f - l * wd;
}

// torch.optim.AdamW (amsgrad = False, maximize = False), operation for operation as its fused kernel:
//   p *= 1 - lr wd;  m += (1 - b1)(g - m);  v = b2 v + (1 - b2) g^2;  p -= (lr / bias1) m / (sqrt(v) / sqrt(bias2) + eps)
// with g the clipped gradient, which is also written back (clip_grad_norm_ scales the gradients in place).
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, int64_t n4, int64_t n, const AdamState* __restrict__ st,
                                                         float beta1, float beta2, float eps, const unsigned char* __restrict__ skip) {
  const AdamState s = *st;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    // skip[j] != 0: elements [64 j, 64 j + 64) belong to a tensor that received no gradient this step; torch's AdamW leaves
    // such a parameter and its moments untouched (no weight decay either), so do we
    if (skip && skip[i >> 4]) continue;
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = reinterpret_cast<f32x4*>(g)[i], mv = reinterpret_cast<f32x4*>(m)[i],
          vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gv[k] * s.coef;
      gv[k] = gg;
      float pp = pv[k] * s.decay;
      mv[k] = mv[k] + (1.f - beta1) * (gg - mv[k]);
      vv[k] = beta2 * vv[k] + (1.f - beta2) * gg * gg;
      pp -= s.step_size * mv[k] / (sqrtf(vv[k]) * s.inv_sqrt_bias2 + eps);
      pv[k] = pp;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv; reinterpret_cast<f32x4*>(g)[i] = gv; reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3) && !(skip && skip[n4 >> 4])) {   // tail (n need not be a multiple of four)
    const int64_t i = n4 * 4 + threadIdx.x;
    const float gg = g[i] * s.coef;
    g[i] = gg;
    float pp = p[i] * s.decay;
    m[i] = m[i] + (1.f - beta1) * (gg - m[i]);
    v[i] = beta2 * v[i] + (1.f - beta2) * gg * gg;
    pp -= s.step_size * m[i] / (sqrtf(v[i]) * s.inv_sqrt_bias2 + eps);
    p[i] = pp;
  }
}

}  // namespace

// One AdamW step on flat fp32 buffers of n elements (16-byte aligned).  step: device fp32 step count (incremented here);
// norm: device scalar, the gradient's 2-norm (null: no clipping); lr: device scalar; state: 4 floats of scratch;
// skip64 (nullable): ceil(n / 64) bytes, non-zero = leave that 64-element block alone (a parameter without a gradient).
extern "C" int fsraft_adamw_flat(float* p, float* g, float* m, float* v, int64_t n, float* step, const float* norm, float max_norm,
                                 const float* lr, float beta1, float beta2, float eps, float weight_decay, float* state,
                                 const unsigned char* skip64, hipStream_t stream) {
  if (!p || !g || !m || !v || !step || !lr || !state || n < 1) return FS_ERR_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return FS_ERR_ARG;
  hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(1), 0, stream, step, norm, max_norm, lr, beta1, beta2, weight_decay,
                     reinterpret_cast<AdamState*>(state));
  int rc = fs_launch_status();
  if (rc) return rc;
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, n4, n,
                     reinterpret_cast<const AdamState*>(state), beta1, beta2, eps, skip64);
  return fs_launch_status();
}

// 0 when `stream` is not being captured, else the id of the capture (hipStreamGetCaptureInfo): the host-side zero pool
// (ops._ZeroPool) keys its chunks on it, so that two captures never share a chunk whose fill node lives in only one of them.
extern "C" int fsraft_stream_capture_id(hipStream_t stream, unsigned long long* id) {
  if (!id) return FS_ERR_ARG;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long cid = 0;
  if (hipStreamGetCaptureInfo(stream, &st, &cid) != hipSuccess) { (void)hipGetLastError(); *id = 0; return FS_ERR_LAUNCH; }
  *id = st == hipStreamCaptureStatusActive ? (cid ? cid : 1ull) : 0ull;
  return FS_OK;
}
