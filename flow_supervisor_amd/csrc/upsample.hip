// Learned 8x convex upsampler, forward and backward (row a9 of SURVEY.md section 8;
// reference: pytorch/core/raft.py:72-83 -- softmax over 9 taps of the mask viewed as
// [N,1,9,8,8,H,W], 3x3 unfold of 8*flow with zero padding, weighted sum, pixel shuffle).
//
//   up[n, c, 8y+sy, 8x+sx] = sum_k softmax_k(mask[n, k*64 + sy*8 + sx, y, x]) * 8*flow[n, c, y+ky-1, x+kx-1]
//
// The mask is consumed channels-last ([N,H,W,576]): one wavefront owns one coarse pixel,
// lane = sy*8+sx, so each of the 9 tap reads is one 256-byte coalesced row and the
// softmax lives entirely in registers (no cross-lane traffic).  Eight neighbouring
// coarse pixels share a workgroup so the 2x8 output rows leave as 256-byte runs.
// HBM bytes per coarse pixel: 2304 (mask) + 512 (up) + ~72 (flow halo).
#include "common.hpp"

namespace {

struct Flow2 {            // 2-channel planar-or-interleaved tensor, element (n,c,pix) at n*bs + c*cs + pix*ps
  const float* p;
  int64_t bs, cs, ps;
};

__device__ __forceinline__ void softmax9(const float (&m)[9], float (&p)[9]) {
  float mx = m[0];
#pragma unroll
  for (int k = 1; k < 9; ++k) mx = fmaxf(mx, m[k]);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { p[k] = expf(m[k] - mx); s += p[k]; }
  const float inv = 1.0f / s;
#pragma unroll
  for (int k = 0; k < 9; ++k) p[k] *= inv;
}

__global__ __launch_bounds__(256) void upsample_fwd_kernel(Flow2 flow, const float* __restrict__ mask,
                                                           float* __restrict__ up, int H, int W) {
  __shared__ float tile[2][8][64];
  const int xb = blockIdx.x * 8, y = blockIdx.y, n = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sy = lane >> 3, sx = lane & 7;
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    const int xl = wave * 2 + pp, x = xb + xl;
    if (x < W) {
      const float* mp = mask + (((int64_t)n * H + y) * W + x) * 576 + lane;
      float m[9], p[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) m[k] = mp[k * 64];
      softmax9(m, p);
      float o0 = 0.f, o1 = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        float f0 = 0.f, f1 = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const int64_t a = n * flow.bs + ((int64_t)yy * W + xx) * flow.ps;
          f0 = 8.f * flow.p[a];
          f1 = 8.f * flow.p[a + flow.cs];
        }
        o0 += p[k] * f0;
        o1 += p[k] * f1;
      }
      tile[0][sy][xl * 8 + sx] = o0;
      tile[1][sy][xl * 8 + sx] = o1;
    }
  }
  __syncthreads();
  const int col = threadIdx.x & 63;
  const int W8 = 8 * W, H8 = 8 * H;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int row = (threadIdx.x >> 6) + 4 * jj;   // 0..15 = c*8 + sy
    const int c = row >> 3, r = row & 7;
    if (xb * 8 + col < W8) up[(((int64_t)n * 2 + c) * H8 + 8 * y + r) * W8 + xb * 8 + col] = tile[c][r][col];
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// dmask (channels-last, same layout as mask) and T[n,c,k,y,x] = sum_s p_k[s] * dup[c][s]  (planar: upsample_dflow_kernel gathers along x)
__global__ __launch_bounds__(256) void upsample_bwd_kernel(Flow2 flow, const float* __restrict__ mask,
                                                           const float* __restrict__ dup, float* __restrict__ dmask,
                                                           float* __restrict__ T, int H, int W, unsigned* __restrict__ dmask_amax) {
  __shared__ float tile[2][8][64];
  unsigned amx = 0u;
  const int xb = blockIdx.x * 8, y = blockIdx.y, n = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sy = lane >> 3, sx = lane & 7;
  const int W8 = 8 * W, H8 = 8 * H;
  {
    const int col = threadIdx.x & 63;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int row = (threadIdx.x >> 6) + 4 * jj;
      const int c = row >> 3, r = row & 7;
      const int cc = xb * 8 + col < W8 ? xb * 8 + col : W8 - 1;      // (clamped, unconditional: the four loads go out together)
      const float dv = dup[(((int64_t)n * 2 + c) * H8 + 8 * y + r) * W8 + cc];
      tile[c][r][col] = (xb * 8 + col < W8) ? dv : 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    const int xl = wave * 2 + pp, x = xb + xl;
    if (x >= W) continue;
    const int64_t pix = ((int64_t)n * H + y) * W + x;
    const int64_t HWp = (int64_t)H * W, pl = (int64_t)y * W + x;
    const float* mp = mask + pix * 576 + lane;
    float m[9], p[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = mp[k * 64];
    softmax9(m, p);
    const float g0 = tile[0][sy][xl * 8 + sx], g1 = tile[1][sy][xl * 8 + sx];
    float dp[9];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
      const int64_t a = n * flow.bs + ((int64_t)yc * W + xc) * flow.ps;
      const float l0 = flow.p[a], l1 = flow.p[a + flow.cs];          // (address clamped: always loaded)
      const float f0 = in ? 8.f * l0 : 0.f, f1 = in ? 8.f * l1 : 0.f;
      dp[k] = g0 * f0 + g1 * f1;
      dot += p[k] * dp[k];
    }
    float* dm = dmask + pix * 576 + lane;
#pragma unroll
    for (int k = 0; k < 9; ++k) { const float dv = p[k] * (dp[k] - dot); dm[k * 64] = dv; amx = fs_umax(amx, fs_abs_bits(dv)); }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float t0 = wave_sum(p[k] * g0);
      const float t1 = wave_sum(p[k] * g1);
      if (lane == 0) {
        T[((int64_t)n * 18 + k) * HWp + pl] = t0;
        T[((int64_t)n * 18 + 9 + k) * HWp + pl] = t1;
      }
    }
  }
  if (dmask_amax) fs_amax_commit_wave(dmask_amax, amx);      // (nullable) word of dmask, raised
}

// ---- the same two kernels on 16-byte accesses (mask, up, dup, dmask 16-byte aligned).  A workgroup owns 16 neighbouring coarse
// pixels; 16 lanes share a pixel, a lane holds four neighbouring sub-pixels (sy = sub >> 1, sx = 4 * (sub & 1) ..+3) of all nine
// taps, so a tap of a pixel is one 256-byte run read by 16 lanes and a wave instruction moves 1 KB.  The softmax stays in
// registers; the output / dup rows are 32 contiguous bytes per pixel and 512 per workgroup row, no LDS staging.
__device__ __forceinline__ void softmax9x4(f32x4 (&m)[9]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float mx = m[0][i];
#pragma unroll
    for (int k = 1; k < 9; ++k) mx = fmaxf(mx, m[k][i]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { m[k][i] = expf(m[k][i] - mx); s += m[k][i]; }
    const float inv = 1.0f / s;
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k][i] *= inv;
  }
}

__device__ __forceinline__ void flow_taps(const Flow2& flow, int n, int y, int x, int H, int W, float (&f0)[9], float (&f1)[9]) {
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
    const int64_t a = n * flow.bs + ((int64_t)yc * W + xc) * flow.ps;
    const float l0 = flow.p[a], l1 = flow.p[a + flow.cs];            // (address clamped: always loaded)
    f0[k] = in ? 8.f * l0 : 0.f;
    f1[k] = in ? 8.f * l1 : 0.f;
  }
}

__global__ __launch_bounds__(256) void upsample_fwd_v4_kernel(Flow2 flow, const float* __restrict__ mask, float* __restrict__ up,
                                                              int H, int W) {
  const int y = blockIdx.y, n = blockIdx.z;
  const int xl = threadIdx.x >> 4, sub = threadIdx.x & 15, sy = sub >> 1, sx = (sub & 1) * 4;
  const int x0 = blockIdx.x * 16 + xl;
  const bool ok = x0 < W;
  const int x = ok ? x0 : W - 1;
  const float* mp = mask + (((int64_t)n * H + y) * W + x) * 576 + sy * 8 + sx;
  f32x4 m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = gload4(mp + k * 64);
  float f0[9], f1[9];
  flow_taps(flow, n, y, x, H, W, f0, f1);
  softmax9x4(m);
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
  for (int k = 0; k < 9; ++k) { o0 += m[k] * f0[k]; o1 += m[k] * f1[k]; }
  if (!ok) return;
  const int W8 = 8 * W, H8 = 8 * H;
  float* o = up + (((int64_t)n * 2) * H8 + 8 * y + sy) * W8 + 8 * x + sx;
  gstore4(o, o0);
  gstore4(o + (int64_t)H8 * W8, o1);
}

__global__ __launch_bounds__(256) void upsample_bwd_v4_kernel(Flow2 flow, const float* __restrict__ mask, const float* __restrict__ dup,
                                                              float* __restrict__ dmask, float* __restrict__ T, int H, int W,
                                                              unsigned* __restrict__ dmask_amax) {
  unsigned amx = 0u;
  const int y = blockIdx.y, n = blockIdx.z;
  const int xl = threadIdx.x >> 4, sub = threadIdx.x & 15, sy = sub >> 1, sx = (sub & 1) * 4;
  const int x0 = blockIdx.x * 16 + xl;
  const bool ok = x0 < W;
  const int x = ok ? x0 : W - 1;
  const int W8 = 8 * W, H8 = 8 * H;
  const int64_t pix = ((int64_t)n * H + y) * W + x;
  const int64_t HWp = (int64_t)H * W, pl = (int64_t)y * W + x;
  const float* mp = mask + pix * 576 + sy * 8 + sx;
  const float* gp = dup + (((int64_t)n * 2) * H8 + 8 * y + sy) * W8 + 8 * x + sx;
  f32x4 m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = gload4(mp + k * 64);
  const f32x4 g0 = gload4(gp), g1 = gload4(gp + (int64_t)H8 * W8);
  float f0[9], f1[9];
  flow_taps(flow, n, y, x, H, W, f0, f1);
  softmax9x4(m);
  f32x4 dot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 9; ++k) dot += m[k] * (g0 * f0[k] + g1 * f1[k]);
  float* dm = dmask + pix * 576 + sy * 8 + sx;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const f32x4 dp = g0 * f0[k] + g1 * f1[k];
    if (ok) { const f32x4 dv = m[k] * (dp - dot); gstore4(dm + k * 64, dv); amx = fs_umax(amx, fs_abs_bits4(dv)); }
    const f32x4 a = m[k] * g0, b = m[k] * g1;
    float t0 = (a[0] + a[1]) + (a[2] + a[3]), t1 = (b[0] + b[1]) + (b[2] + b[3]);
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { t0 += __shfl_xor(t0, o, 64); t1 += __shfl_xor(t1, o, 64); }
    if (ok && sub == 0) {
      T[((int64_t)n * 18 + k) * HWp + pl] = t0;
      T[((int64_t)n * 18 + 9 + k) * HWp + pl] = t1;
    }
  }
  __shared__ unsigned red[4];
  if (dmask_amax) fs_amax_commit(dmask_amax, amx, red);      // (one look at the word per workgroup: 21 K of them at the bench shape)
}

// dflow[n,c,y,x] = 8 * sum_k T[n, c, k, y-(ky-1), x-(kx-1)]   (deterministic gather, no atomics)
__global__ __launch_bounds__(256) void upsample_dflow_kernel(const float* __restrict__ T, float* __restrict__ dflow,
                                                             int N, int H, int W) {
  const int64_t total = (int64_t)N * 2 * H * W;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int x = (int)(e % W);
  const int y = (int)((e / W) % H);
  const int c = (int)((e / ((int64_t)W * H)) % 2);
  const int n = (int)(e / ((int64_t)2 * W * H));
  float v[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {            // (addresses clamped, loads unconditional: `in ? *p : 0` is a branch per load, i.e. nine round trips)
    const int yy = y - (k / 3 - 1), xx = x - (k % 3 - 1);
    const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
    const float t = T[((int64_t)n * 18 + c * 9 + k) * ((int64_t)H * W) + yc * W + xc];
    v[k] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? t : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) s += v[k];
  dflow[e] = 8.f * s;
}

// ---- bilinear x8 (raft-small): 8 * interpolate(flow, align_corners=True), core/utils/utils.py:80-82
__global__ __launch_bounds__(256) void upflow8_fwd_kernel(const float* __restrict__ flow, float* __restrict__ up,
                                                          int NC, int H, int W) {
  const int H8 = 8 * H, W8 = 8 * W;
  const int64_t total = (int64_t)NC * H8 * W8;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int X = (int)(e % W8), Y = (int)((e / W8) % H8);
  const int64_t nc = e / ((int64_t)W8 * H8);
  const float sy = H8 > 1 ? (float)(H - 1) / (float)(H8 - 1) : 0.f;
  const float sx = W8 > 1 ? (float)(W - 1) / (float)(W8 - 1) : 0.f;
  const float fy = sy * Y, fx = sx * X;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float ly = fy - y0, lx = fx - x0;
  const float* f = flow + nc * H * W;
  const float v = (1.f - ly) * ((1.f - lx) * f[y0 * W + x0] + lx * f[y0 * W + x1]) +
                  ly * ((1.f - lx) * f[y1 * W + x0] + lx * f[y1 * W + x1]);
  up[e] = 8.f * v;
}

__global__ __launch_bounds__(256) void upflow8_bwd_kernel(const float* __restrict__ dup, float* __restrict__ dflow,
                                                          int NC, int H, int W) {
  // one thread per coarse pixel gathers from the fine pixels whose 2x2 stencil touches it
  const int H8 = 8 * H, W8 = 8 * W;
  const int64_t total = (int64_t)NC * H * W;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int x = (int)(e % W), y = (int)((e / W) % H);
  const int64_t nc = e / ((int64_t)W * H);
  const float sy = H8 > 1 ? (float)(H - 1) / (float)(H8 - 1) : 0.f;
  const float sx = W8 > 1 ? (float)(W - 1) / (float)(W8 - 1) : 0.f;
  // fine rows Y with floor(sy*Y) in {y-1, y}: conservative range then exact test
  const int Ylo = max(0, (int)floorf((y - 1) / fmaxf(sy, 1e-12f)) - 1), Yhi = min(H8 - 1, (int)ceilf((y + 1) / fmaxf(sy, 1e-12f)) + 1);
  const int Xlo = max(0, (int)floorf((x - 1) / fmaxf(sx, 1e-12f)) - 1), Xhi = min(W8 - 1, (int)ceilf((x + 1) / fmaxf(sx, 1e-12f)) + 1);
  const float* g = dup + nc * H8 * W8;
  float s = 0.f;
  for (int Y = Ylo; Y <= Yhi; ++Y) {
    const float fy = sy * Y;
    const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
    const float ly = fy - y0;
    float wy = 0.f;
    if (y0 == y) wy += 1.f - ly;
    if (y1 == y) wy += ly;
    if (wy == 0.f) continue;
    for (int X = Xlo; X <= Xhi; ++X) {
      const float fx = sx * X;
      const int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
      const float lx = fx - x0;
      float wx = 0.f;
      if (x0 == x) wx += 1.f - lx;
      if (x1 == x) wx += lx;
      if (wx != 0.f) s += wy * wx * g[(int64_t)Y * W8 + X];
    }
  }
  dflow[e] = 8.f * s;
}

}  // namespace

int g_upsample_v4 = 1;      // 16-byte kernels (fsraft_set_upsample_kernel; 0: the 4-byte ones, which also take unaligned tensors)
extern "C" int fsraft_set_upsample_kernel(int v4) { g_upsample_v4 = v4 != 0; return FS_OK; }

// flow element (n,c,pix) at flow[n*flow_bs + c*flow_cs + pix*flow_ps]; mask is [N,H,W,576]; up is [N,2,8H,8W].
extern "C" int fsraft_upsample_fwd(const float* flow, int64_t flow_bs, int64_t flow_cs, int64_t flow_ps,
                                   const float* mask_nhwc, float* up, int N, int H, int W, hipStream_t stream) {
  if (!flow || !mask_nhwc || !up || N < 1 || H < 1 || W < 1) return FS_ERR_ARG;
  Flow2 f{flow, flow_bs, flow_cs, flow_ps};
  if (g_upsample_v4 && (((uintptr_t)mask_nhwc | (uintptr_t)up) & 15) == 0)
    hipLaunchKernelGGL(upsample_fwd_v4_kernel, dim3(ceil_div(W, 16), H, N), dim3(256), 0, stream, f, mask_nhwc, up, H, W);
  else
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(ceil_div(W, 8), H, N), dim3(256), 0, stream, f, mask_nhwc, up, H, W);
  return fs_launch_status();
}

// dmask_nhwc [N,H,W,576], dflow [N,2,H,W] contiguous; scratch holds N*H*W*18 floats.
extern "C" int fsraft_upsample_bwd(const float* flow, int64_t flow_bs, int64_t flow_cs, int64_t flow_ps,
                                   const float* mask_nhwc, const float* dup, float* dmask_nhwc, float* dflow,
                                   float* scratch, int N, int H, int W, unsigned* dmask_amax, hipStream_t stream) {
  if (!flow || !mask_nhwc || !dup || !dmask_nhwc || !dflow || !scratch || N < 1 || H < 1 || W < 1 || ((uintptr_t)dmask_amax & 3)) return FS_ERR_ARG;
  Flow2 f{flow, flow_bs, flow_cs, flow_ps};
  if (g_upsample_v4 && (((uintptr_t)mask_nhwc | (uintptr_t)dup | (uintptr_t)dmask_nhwc) & 15) == 0)
    hipLaunchKernelGGL(upsample_bwd_v4_kernel, dim3(ceil_div(W, 16), H, N), dim3(256), 0, stream, f, mask_nhwc, dup, dmask_nhwc, scratch, H, W, dmask_amax);
  else
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(ceil_div(W, 8), H, N), dim3(256), 0, stream, f, mask_nhwc, dup,
                       dmask_nhwc, scratch, H, W, dmask_amax);
  const int64_t total = (int64_t)N * 2 * H * W;
  hipLaunchKernelGGL(upsample_dflow_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, scratch, dflow,
                     N, H, W);
  return fs_launch_status();
}

extern "C" int fsraft_upflow8_fwd(const float* flow, float* up, int N, int C, int H, int W, hipStream_t stream) {
  if (!flow || !up || N < 1 || C < 1 || H < 1 || W < 1) return FS_ERR_ARG;
  const int64_t total = (int64_t)N * C * 64 * H * W;
  hipLaunchKernelGGL(upflow8_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, flow, up, N * C, H, W);
  return fs_launch_status();
}

extern "C" int fsraft_upflow8_bwd(const float* dup, float* dflow, int N, int C, int H, int W, hipStream_t stream) {
  if (!dup || !dflow || N < 1 || C < 1 || H < 1 || W < 1) return FS_ERR_ARG;
  const int64_t total = (int64_t)N * C * H * W;
  hipLaunchKernelGGL(upflow8_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dup, dflow, N * C, H, W);
  return fs_launch_status();
}
