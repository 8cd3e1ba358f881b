// Channels-last ([B][HW][C], C % 4 == 0, C <= 256) variants of the encoder normalisation + ReLU kernels of norm.hip, for
// the encoder stages whose 3x3 convolutions run on the split-bf16 implicit-GEMM kernels (which produce and consume
// channels-last tensors).  A per-(sample, channel) statistic is a column reduction here: every workgroup walks a strip
// of pixels with one float4 of channels per lane, reduces across its pixel lanes in LDS and adds its partial sums to
// [B][C] accumulators with atomics (the caller zeroes them).
#include "common.hpp"

namespace {

// pixels per workgroup: a few thousand workgroups per launch, so that every CU holds several and the strided walks
// (one 16-byte load per lane and step, unrolled by four) hide each other's latency
constexpr int CL_NSLOT = 8;
int g_cl_target_wgs = 4096;        // workgroups aimed at per launch (fsraft_set_norm_blocks)
inline int pix_per_wg(int B, int HW) {
  int64_t p = ((int64_t)B * HW + g_cl_target_wgs - 1) / g_cl_target_wgs;
  p = (p + 63) / 64 * 64;
  return p < 128 ? 128 : p > 1024 ? 1024 : (int)p;
}

// Space-to-depth addressing of a channels-last tensor (what a stride-2 unit's two convolutions read, fsraft_space_to_depth2:
// [B][H/2][W/2][2][2][C]): float offset of channel 0 of pixel p = y * W + x of sample b.  The norm that feeds such a unit writes
// its result there directly (and its backward reads the gradient / its saved result from there): the 2 x 230 MB layout copies
// per stride-2 unit and direction disappear.  s2w = W (even, H even) or 0: plain [B][HW][C].
__device__ __forceinline__ int64_t cl_pix_off(int b, int p, int HW, int C, int s2w) {
  if (s2w == 0) return ((int64_t)b * HW + p) * C;
  const int y = p / s2w, x = p - y * s2w;
  return (((int64_t)b * (HW >> 2) + (y >> 1) * (s2w >> 1) + (x >> 1)) * 4 + ((y & 1) * 2 + (x & 1))) * C;
}

// sums[b][c] += sum_pix x, sumsq[b][c] += sum_pix x^2
__global__ __launch_bounds__(256) void cl_stats_kernel(const float* __restrict__ x, float* __restrict__ sums,
                                                       float* __restrict__ sumsq, int HW, int C, int PIX_PER_WG) {
  __shared__ f32x4 red[2][256];
  const int c4n = C >> 2, lanes_p = 256 / c4n;            // pixels covered per step
  const int cl = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * PIX_PER_WG, p1 = min(p0 + PIX_PER_WG, HW);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
  if (pl < lanes_p)
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += lanes_p) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((int64_t)b * HW + p) * C + cl * 4);
      s += v; q += v * v;
    }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < c4n) {
    f32x4 ts = {0.f, 0.f, 0.f, 0.f}, tq = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < lanes_p; ++k) { ts += red[0][k * c4n + threadIdx.x]; tq += red[1][k * c4n + threadIdx.x]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = (b * CL_NSLOT + (int)(blockIdx.x % CL_NSLOT)) * C + threadIdx.x * 4 + i;     // partial rows: see CL_NSLOT
      atomicAdd(sums + o, ts[i]);
      atomicAdd(sumsq + o, tq[i]);
    }
  }
}

// y = relu?((x - mean) * rstd) with mean / rstd from the accumulated sums; stats[b][c] = (mean, rstd) written by block (0, b)
__global__ __launch_bounds__(256) void cl_inorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ sums,
                                                             const float* __restrict__ sumsq, float* __restrict__ y,
                                                             float* __restrict__ stats, int HW, int C, float eps, int relu, int PIX_PER_WG,
                                                             const float* __restrict__ res, int s2w, unsigned* __restrict__ y_amax) {
  unsigned amx = 0u;                    // (y_amax, nullable: the amax word of y, raised -- the convolution behind the norm reads it)
  const int c4n = C >> 2, lanes_p = 256 / c4n;
  const int cl = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const int b = blockIdx.y;
  f32x4 mean, rstd;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = cl * 4 + i;
    float su = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < CL_NSLOT; ++k) { su += sums[(b * CL_NSLOT + k) * C + c]; sq += sumsq[(b * CL_NSLOT + k) * C + c]; }
    const float m = su / (float)HW;
    const float var = fmaxf(sq / (float)HW - m * m, 0.f);
    mean[i] = m; rstd[i] = rsqrtf(var + eps);
    if (blockIdx.x == 0 && pl == 0) { stats[(b * C + c) * 2] = m; stats[(b * C + c) * 2 + 1] = rstd[i]; }
  }
  const int p0 = blockIdx.x * PIX_PER_WG, p1 = min(p0 + PIX_PER_WG, HW);
  if (pl < lanes_p)
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += lanes_p) {
      const int64_t o = ((int64_t)b * HW + p) * C + cl * 4;
      f32x4 v = (*reinterpret_cast<const f32x4*>(x + o) - mean) * rstd;
      if (relu) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
      }
      if (res) {                                            // residual unit: y = relu(res + relu?(norm(x)))
        const f32x4 rv = *reinterpret_cast<const f32x4*>(res + o);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] + rv[i], 0.f);
      }
      *reinterpret_cast<f32x4*>(y + (s2w ? cl_pix_off(b, p, HW, C, s2w) + cl * 4 : o)) = v;
      amx = fs_umax(amx, fs_abs_bits4(v));
    }
  __shared__ unsigned ared[4];
  if (y_amax) fs_amax_commit(y_amax, amx, ared);
}

// MODE 0 (instance norm): xhat = (x - mean) * rstd, g' = g * (xhat > 0 | !relu); s1[b][c] += sum g', s2[b][c] += sum g' xhat
// MODE 1 (affine):        t = x * scale + shift,    g' = g * (t > 0 | !relu);    s1[c]    += sum g', s2[c]    += sum g' x; dx = g' * scale
template <int MODE>
__global__ __launch_bounds__(256) void cl_bwd_sums_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                          const float* __restrict__ pa, const float* __restrict__ pb,
                                                          float* __restrict__ s1, float* __restrict__ s2,
                                                          float* __restrict__ dx, int HW, int C, int relu, int PIX_PER_WG,
                                                          const float* __restrict__ out, float* __restrict__ dres, int s2w,
                                                          unsigned* __restrict__ dx_amax) {        // (MODE 1: word of dx, raised)
  __shared__ f32x4 red[2][256];
  unsigned amx = 0u;
  const int c4n = C >> 2, lanes_p = 256 / c4n;
  const int cl = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const int b = blockIdx.y;
  f32x4 A, Bv;                                             // MODE 0: mean, rstd (per b,c)   MODE 1: scale, shift (per c)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = cl * 4 + i;
    if (MODE == 0) { A[i] = pa[(b * C + c) * 2]; Bv[i] = pa[(b * C + c) * 2 + 1]; }
    else { A[i] = pa[c]; Bv[i] = pb[c]; }
  }
  const int p0 = blockIdx.x * PIX_PER_WG, p1 = min(p0 + PIX_PER_WG, HW);
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
  if (pl < lanes_p)
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += lanes_p) {
      const int64_t o = ((int64_t)b * HW + p) * C + cl * 4;
      const int64_t og = s2w ? cl_pix_off(b, p, HW, C, s2w) + cl * 4 : o;      // (the gradient and the saved result: see cl_pix_off)
      f32x4 gv = *reinterpret_cast<const f32x4*>(g + og);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
      if (out) {                                            // fused residual: the gradient first passes relu(res + y)
        const f32x4 ov = *reinterpret_cast<const f32x4*>(out + og);
#pragma unroll
        for (int i = 0; i < 4; ++i) gv[i] = ov[i] > 0.f ? gv[i] : 0.f;
        *reinterpret_cast<f32x4*>(dres + o) = gv;           // ... and this is what the shortcut receives
      }
      f32x4 dv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = MODE == 0 ? (xv[i] - A[i]) * Bv[i] : xv[i] * A[i] + Bv[i];
        const float gg = (relu && t <= 0.f) ? 0.f : gv[i];
        a1[i] += gg;
        a2[i] += gg * (MODE == 0 ? t : xv[i]);
        dv[i] = gg * A[i];
      }
      if (MODE == 1) { *reinterpret_cast<f32x4*>(dx + o) = dv; amx = fs_umax(amx, fs_abs_bits4(dv)); }
    }
  red[0][threadIdx.x] = a1; red[1][threadIdx.x] = a2;
  __syncthreads();
  if (threadIdx.x < c4n) {
    f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < lanes_p; ++k) { t1 += red[0][k * c4n + threadIdx.x]; t2 += red[1][k * c4n + threadIdx.x]; }
    // MODE 1 sums over the whole batch: spread the workgroups over gridDim.y * NSLOT partial rows ([B * NSLOT][C],
    // summed by the caller) -- thousands of atomics on the same C addresses serialise in L2 otherwise
    const int base = (b * CL_NSLOT + (int)(blockIdx.x % CL_NSLOT)) * C + threadIdx.x * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) { atomicAdd(s1 + base + i, t1[i]); atomicAdd(s2 + base + i, t2[i]); }
  }
  if (MODE == 1 && dx_amax) fs_amax_commit(dx_amax, amx, reinterpret_cast<unsigned*>(&red[0][0]));
}

// dx = rstd * (g' - mean(g') - xhat * mean(g' xhat))
__global__ __launch_bounds__(256) void cl_inorm_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                 const float* __restrict__ stats, const float* __restrict__ s1,
                                                                 const float* __restrict__ s2, float* __restrict__ dx, int HW,
                                                                 int C, int relu, int PIX_PER_WG, const float* __restrict__ out, int s2w,
                                                                 unsigned* __restrict__ dx_amax) {
  unsigned amx = 0u;
  const int c4n = C >> 2, lanes_p = 256 / c4n;
  const int cl = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const int b = blockIdx.y;
  f32x4 mean, rstd, m1, m2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = cl * 4 + i;
    mean[i] = stats[(b * C + c) * 2]; rstd[i] = stats[(b * C + c) * 2 + 1];
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < CL_NSLOT; ++k) { a1 += s1[(b * CL_NSLOT + k) * C + c]; a2 += s2[(b * CL_NSLOT + k) * C + c]; }
    m1[i] = a1 / (float)HW; m2[i] = a2 / (float)HW;
  }
  const int p0 = blockIdx.x * PIX_PER_WG, p1 = min(p0 + PIX_PER_WG, HW);
  if (pl < lanes_p)
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += lanes_p) {
      const int64_t o = ((int64_t)b * HW + p) * C + cl * 4;
      const int64_t og = s2w ? cl_pix_off(b, p, HW, C, s2w) + cl * 4 : o;
      f32x4 gv = *reinterpret_cast<const f32x4*>(g + og);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
      if (out) {
        const f32x4 ov = *reinterpret_cast<const f32x4*>(out + og);
#pragma unroll
        for (int i = 0; i < 4; ++i) gv[i] = ov[i] > 0.f ? gv[i] : 0.f;
      }
      f32x4 dv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (xv[i] - mean[i]) * rstd[i];
        const float gg = (relu && xh <= 0.f) ? 0.f : gv[i];
        dv[i] = rstd[i] * (gg - m1[i] - xh * m2[i]);
      }
      *reinterpret_cast<f32x4*>(dx + o) = dv;
      amx = fs_umax(amx, fs_abs_bits4(dv));
    }
  __shared__ unsigned ared[4];
  if (dx_amax) fs_amax_commit(dx_amax, amx, ared);
}

// y = relu?(x * scale[c] + shift[c])
__global__ __launch_bounds__(256) void cl_affine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ y,
                                                            int64_t M, int C, int relu, const float* __restrict__ res, int HW, int s2w,
                                                            unsigned* __restrict__ y_amax) {
  const int c4n = C >> 2;
  const int64_t total = M * c4n;
  unsigned amx = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % c4n) * 4;
    f32x4 v = reinterpret_cast<const f32x4*>(x)[e];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = v[i] * scale[c + i] + shift[c + i]; if (relu) v[i] = fmaxf(v[i], 0.f); }
    if (res) {
      const f32x4 rv = reinterpret_cast<const f32x4*>(res)[e];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] + rv[i], 0.f);
    }
    if (s2w) {
      const int64_t m = e / c4n;
      *reinterpret_cast<f32x4*>(y + cl_pix_off((int)(m / HW), (int)(m % HW), HW, C, s2w) + c) = v;
    } else {
      reinterpret_cast<f32x4*>(y)[e] = v;
    }
    amx = fs_umax(amx, fs_abs_bits4(v));
    if (y_amax && e < (int64_t)gridDim.x * 256) fs_amax_early(y_amax, amx);
  }
  __shared__ unsigned ared[4];
  if (y_amax) fs_amax_commit(y_amax, amx, ared);
}

inline bool cl_ok(int C) { return C >= 4 && C <= 256 && C % 4 == 0; }      // (threads beyond (256 / (C/4)) * (C/4) idle)
inline bool s2d_ok(int HW, int w) { return w == 0 || (w > 0 && (w & 1) == 0 && HW % w == 0 && ((HW / w) & 1) == 0); }   // even width and height

}  // namespace

extern "C" int fsraft_set_norm_blocks(int target_workgroups) {     // tuning hook
  if (target_workgroups < 64) return FS_ERR_ARG;
  g_cl_target_wgs = target_workgroups;
  return FS_OK;
}

// x, y: [B][HW][C].  sums / sumsq: [B * 8][C] partial-row scratch that must be ZERO on entry; stats: [B][C][2] = (mean, rstd) out.
// res (nullable, [B][HW][C]): fused residual unit, y = relu(res + relu?(norm(x))).
extern "C" int fsraft_inorm_relu_cl_fwd(const float* x, const float* res, float* y, float* sums, float* sumsq, float* stats, int B,
                                        int HW, int C, float eps, int relu, int have_sums, int s2d_w, unsigned* y_amax, hipStream_t s) {
  if (!x || !y || !sums || !sumsq || !stats || B < 1 || HW < 1 || !cl_ok(C) || !s2d_ok(HW, s2d_w) || ((uintptr_t)y_amax & 3)) return FS_ERR_ARG;
  const int ppw = pix_per_wg(B, HW);
  dim3 grid(ceil_div(HW, ppw), B);
  // have_sums: the producing convolution already accumulated the partial rows (fsraft_conv_forward_stats): no pass of our own
  if (!have_sums) hipLaunchKernelGGL(cl_stats_kernel, grid, dim3(256), 0, s, x, sums, sumsq, HW, C, ppw);
  hipLaunchKernelGGL(cl_inorm_apply_kernel, grid, dim3(256), 0, s, x, sums, sumsq, y, stats, HW, C, eps, relu, ppw, res, s2d_w, y_amax);
  return fs_launch_status();
}
// s1, s2: [B * 8][C] partial-row scratch, ZERO on entry.  Fused residual unit: out = the forward result y, dres receives the shortcut's
// gradient g * (out > 0), and the norm branch continues from that; both NULL otherwise.
extern "C" int fsraft_inorm_relu_cl_bwd(const float* g, const float* x, const float* stats, const float* out, float* s1, float* s2,
                                        float* dx, float* dres, int B, int HW, int C, int relu, int s2d_w, unsigned* dx_amax, hipStream_t s) {
  if (!g || !x || !stats || !s1 || !s2 || !dx || B < 1 || HW < 1 || !cl_ok(C) || (out != nullptr) != (dres != nullptr) || !s2d_ok(HW, s2d_w) ||
      ((uintptr_t)dx_amax & 3))
    return FS_ERR_ARG;
  const int ppw = pix_per_wg(B, HW);
  dim3 grid(ceil_div(HW, ppw), B);
  hipLaunchKernelGGL((cl_bwd_sums_kernel<0>), grid, dim3(256), 0, s, g, x, stats, nullptr, s1, s2, nullptr, HW, C, relu, ppw, out, dres, s2d_w, nullptr);
  // fused residual unit: the first kernel just wrote dres = g * (out > 0) -- exactly what the second would rebuild from g and out,
  // so it reads that one tensor instead of the two (and in the plain layout, whatever layout g arrived in)
  if (dres) hipLaunchKernelGGL(cl_inorm_bwd_apply_kernel, grid, dim3(256), 0, s, dres, x, stats, s1, s2, dx, HW, C, relu, ppw, nullptr, 0, dx_amax);
  else hipLaunchKernelGGL(cl_inorm_bwd_apply_kernel, grid, dim3(256), 0, s, g, x, stats, s1, s2, dx, HW, C, relu, ppw, out, s2d_w, dx_amax);
  return fs_launch_status();
}
extern "C" int fsraft_affine_relu_cl_fwd(const float* x, const float* res, const float* scale, const float* shift, float* y, int64_t M,
                                         int C, int relu, int HW, int s2d_w, unsigned* y_amax, hipStream_t s) {
  if (!x || !scale || !shift || !y || M < 1 || C < 4 || C % 4 || (s2d_w && (HW < 4 || M % HW || !s2d_ok(HW, s2d_w))) || ((uintptr_t)y_amax & 3))
    return FS_ERR_ARG;
  int64_t blocks = (M * (C / 4) + 255) / 256;
  if (blocks > (y_amax ? 2048 : 8192)) blocks = y_amax ? 2048 : 8192;
  hipLaunchKernelGGL(cl_affine_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, scale, shift, y, M, C, relu, res, HW > 0 ? HW : 1, s2d_w, y_amax);
  return fs_launch_status();
}
// dx = g' * scale[c]; partial sums of g' and g' * x over pixels, spread over B * 8 rows to keep the atomics apart:
// dsum_g, dsum_gx: [B * 8][C], ZERO on entry; the caller adds the rows up.  out / dres as in fsraft_inorm_relu_cl_bwd.
extern "C" int fsraft_affine_relu_cl_bwd(const float* g, const float* x, const float* scale, const float* shift, const float* out,
                                         float* dx, float* dres, float* dsum_g, float* dsum_gx, int B, int HW, int C, int relu,
                                         int s2d_w, unsigned* dx_amax, hipStream_t s) {
  if ((uintptr_t)dx_amax & 3) return FS_ERR_ARG;
  if (!g || !x || !scale || !shift || !dx || !dsum_g || !dsum_gx || B < 1 || HW < 1 || !cl_ok(C) ||
      (out != nullptr) != (dres != nullptr) || !s2d_ok(HW, s2d_w)) return FS_ERR_ARG;
  const int ppw = pix_per_wg(B, HW);
  dim3 grid(ceil_div(HW, ppw), B);
  hipLaunchKernelGGL((cl_bwd_sums_kernel<1>), grid, dim3(256), 0, s, g, x, scale, shift, dsum_g, dsum_gx, dx, HW, C, relu, ppw, out, dres, s2d_w, dx_amax);
  return fs_launch_status();
}

// ---- frozen-BatchNorm parameter folding (pytorch/core/extractor.py's BatchNorm2d layers after freeze_bn) ---------------
// y = relu?(x * scale + shift) with scale = weight * rsqrt(var + eps), shift = bias - (mean - cbias) * scale, where cbias is
// the bias the preceding convolution ran without.  The framework needs five elementwise launches on [C] tensors for this
// per layer and step, and six more for the parameter gradients; here each direction is one launch.
namespace {
__global__ void bn_fold_kernel(const float* __restrict__ weight, const float* __restrict__ bias, const float* __restrict__ rm,
                               const float* __restrict__ rv, const float* __restrict__ cbias, float eps, int C, float* __restrict__ scale,
                               float* __restrict__ shift, float* __restrict__ rs, float* __restrict__ rmc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float r = rsqrtf(rv[c] + eps), sc = weight[c] * r, m = rm[c] - (cbias ? cbias[c] : 0.f);
  rs[c] = r; scale[c] = sc; rmc[c] = m; shift[c] = bias[c] - m * sc;
}
// part: [2][R][C] partial column sums of g' (part 0) and g' * x (part 1) from fsraft_affine_relu_cl_bwd
__global__ void bn_fold_bwd_kernel(const float* __restrict__ part, int R, int C, const float* __restrict__ rs, const float* __restrict__ rmc,
                                   const float* __restrict__ scale, float* __restrict__ dweight, float* __restrict__ dbias,
                                   float* __restrict__ dcbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  // (the per-channel constants and eight rows of both sums are requested before the first add: rolled, the loop was a chain of
  //  R dependent round trips to L2 -- 11 us per launch for R = 32, fifteen launches per step)
  const float rsc = rs[c], rmcc = rmc[c], scc = scale[c];
  float s0 = 0.f, s1 = 0.f;
  int r = 0;
  for (; r + 8 <= R; r += 8) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = part[(int64_t)(r + i) * C + c]; b[i] = part[((int64_t)R + r + i) * C + c]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { s0 += a[i]; s1 += b[i]; }
  }
  for (; r < R; ++r) { s0 += part[(int64_t)r * C + c]; s1 += part[((int64_t)R + r) * C + c]; }
  dweight[c] = rsc * (s1 - rmcc * s0);
  dbias[c] = s0;
  if (dcbias) dcbias[c] = scc * s0;
}
}  // namespace

extern "C" int fsraft_bn_fold(const float* weight, const float* bias, const float* rm, const float* rv, const float* cbias, float eps,
                              int C, float* scale, float* shift, float* rs, float* rmc, hipStream_t stream) {
  if (!weight || !bias || !rm || !rv || !scale || !shift || !rs || !rmc || C < 1) return FS_ERR_ARG;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, weight, bias, rm, rv, cbias, eps, C, scale, shift, rs, rmc);
  return fs_launch_status();
}
extern "C" int fsraft_bn_fold_bwd(const float* part, int R, int C, const float* rs, const float* rmc, const float* scale,
                                  float* dweight, float* dbias, float* dcbias, hipStream_t stream) {
  if (!part || !rs || !rmc || !scale || !dweight || !dbias || R < 1 || C < 1) return FS_ERR_ARG;
  hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, part, R, C, rs, rmc, scale, dweight, dbias, dcbias);
  return fs_launch_status();
}
