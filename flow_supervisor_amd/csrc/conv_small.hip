// Convolutions with a handful of output channels (the flow head's 3x3 256 -> 2, pytorch/core/update.py:6-14): a GEMM
// tile would be >95 % padding (the 32-wide exact kernel needs 95 us for 0.26 GFLOP), so these are per-pixel dot
// products instead: one wave per pixel, lane l owns input channels [4l, 4l+4) (+256 for a second register set), the
// weights of all taps live in registers, every tap is one coalesced 1 KB row load, and the NOUT partial sums are
// reduced across the wave with shuffles.  Channels-last input [M][ld], planar output out[b][o][pix] (+ strides).
#include "common.hpp"
#include <cstddef>

namespace {


// (taps are compile-time: a run-time tap count leaves the tap loops rolled, and the register arrays indexed by them
// then live in scratch)
template <int KC, int NOUT, int KH, int KW>
__global__ __launch_bounds__(256) void conv_small_fwd_kernel(const float* __restrict__ x, int ld, int C,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ out, int64_t obs, int64_t ocs, int64_t ops,
                                                             int B, int H, int W) {
  const int lane = threadIdx.x & 63;
  constexpr int taps = KH * KW, PH = KH / 2, PW = KW / 2;
  const int HW = H * W;
  const int64_t M = (int64_t)B * HW;
  // weights: w[o][c][tap] (OIHW) -> registers wr[o][tap][kc] as float4 over this lane's channels
  f32x4 wr[NOUT][taps][KC];
#pragma unroll
  for (int o = 0; o < NOUT; ++o)
#pragma unroll
    for (int t = 0; t < taps; ++t)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = kc * 256 + lane * 4 + i;
          if (c < C) v[i] = w[((int64_t)o * C + c) * taps + t];
        }
        wr[o][t][kc] = v;
      }
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t m = wave; m < M; m += nwaves) {
    const int pix = (int)(m % HW), y = pix / W, xx = pix % W;
    float acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = 0.f;
#pragma unroll
    for (int t = 0; t < taps; ++t) {
      const int yy = y + t / KW - PH, xs = xx + t % KW - PW;
      const bool in = (unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W;       // wave-uniform
      const float* row = x + (m + (int64_t)(t / KW - PH) * W + (t % KW - PW)) * ld;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int c = kc * 256 + lane * 4;
        if (in && c < C) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
#pragma unroll
          for (int o = 0; o < NOUT; ++o)
            acc[o] += v[0] * wr[o][t][kc][0] + v[1] * wr[o][t][kc][1] + v[2] * wr[o][t][kc][2] + v[3] * wr[o][t][kc][3];
        }
      }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) acc[o] += __shfl_xor(acc[o], s, 64);
    }
    if (lane == 0) {
      const int b = (int)(m / HW);
#pragma unroll
      for (int o = 0; o < NOUT; ++o) out[b * obs + o * ocs + pix * ops] = acc[o] + (bias ? bias[o] : 0.f);
    }
  }
}

// dw[o][c][tap] += sum_pixels dy[pix][o] * x[pix + shift(tap)][c]  (dw in the packed layout wpk[o][tap * cpad + c]);
// dbias[o] += sum dy[pix][o].  Several (dy, x) segments (the iterations of a step) in one launch.
constexpr int SMALL_MAX_SEG = 16;
struct SmallWgradArgs {
  const float* dy[SMALL_MAX_SEG]; const float* x[SMALL_MAX_SEG]; int nseg;
  int ldy, ldx, C, B, H, W, KH, KW, Ktot;
  float* dwpk; float* dbias;
};

template <int KC, int NOUT, int KH, int KW>
__global__ __launch_bounds__(256) void conv_small_wgrad_kernel(SmallWgradArgs a) {
  __shared__ float red[4][64 * 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int taps = KH * KW, PH = KH / 2, PW = KW / 2;
  const int HW = a.H * a.W;
  const int64_t M = (int64_t)a.B * HW;
  f32x4 acc[NOUT][taps][KC];
#pragma unroll
  for (int o = 0; o < NOUT; ++o)
#pragma unroll
    for (int t = 0; t < taps; ++t)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) acc[o][t][kc] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) bsum[o] = 0.f;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
  const auto* karg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  typedef const float* fptr;
  for (int seg = 0; seg < a.nseg; ++seg) {      // pointer tables read from the kernarg segment (uniform index)
    const float* dy = ((const fptr __attribute__((address_space(4)))*)(karg + offsetof(SmallWgradArgs, dy)))[seg];
    const float* x = ((const fptr __attribute__((address_space(4)))*)(karg + offsetof(SmallWgradArgs, x)))[seg];
    for (int64_t m = wave; m < M; m += nwaves) {
      const int pix = (int)(m % HW), y = pix / a.W, xx = pix % a.W;
      float g[NOUT];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) { g[o] = dy[m * a.ldy + o]; bsum[o] += g[o]; }
#pragma unroll
      for (int t = 0; t < taps; ++t) {
        const int yy = y + t / KW - PH, xs = xx + t % KW - PW;
        const bool in = (unsigned)yy < (unsigned)a.H && (unsigned)xs < (unsigned)a.W;
        const float* row = x + (m + (int64_t)(t / KW - PH) * a.W + (t % KW - PW)) * a.ldx;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int c = kc * 256 + lane * 4;
          if (in && c < a.C) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) acc[o][t][kc] += g[o] * v;
          }
        }
      }
    }
  }
  // reduce the four waves of the workgroup through LDS, then one atomic per (o, tap, channel) per workgroup
  const int cpad = ((a.C + 31) / 32) * 32;
#pragma unroll
  for (int o = 0; o < NOUT; ++o)
#pragma unroll
    for (int t = 0; t < taps; ++t) {
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wv][lane * 4 + i] = acc[o][t][kc][i];
        __syncthreads();
        const int c = kc * 256 + threadIdx.x;
        if (c < a.C && threadIdx.x < 256) {
          const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
          atomicAdd(a.dwpk + (int64_t)o * a.Ktot + t * cpad + c, s);
        }
      }
    }
  if (a.dbias && lane == 0) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) atomicAdd(a.dbias + o, bsum[o]);
  }
}

}  // namespace

// out[b][o][pix] = bias[o] + sum_{c,tap} w[o][c][tap] * x[pixel + shift(tap)][c];  N = 2 outputs, C <= 512, KH*KW <= 9.
extern "C" int fsraft_conv_small_fwd(const float* x, int ld, int C, const float* w_oihw, const float* bias, float* out,
                                     int64_t obs, int64_t ocs, int64_t ops, int N, int B, int H, int W, int KH, int KW,
                                     hipStream_t s) {
  if (!x || !w_oihw || !out || N != 2 || C < 4 || C > 512 || C % 4 || ld % 4 || KH != 3 || KW != 3 || B < 1) return FS_ERR_ARG;
  const int64_t M = (int64_t)B * H * W;
  int blocks = (int)((M + 31) / 32);                  // ~8 pixels per wave
  if (blocks > 2048) blocks = 2048;
  if (C <= 256) hipLaunchKernelGGL((conv_small_fwd_kernel<1, 2, 3, 3>), dim3(blocks), dim3(256), 0, s, x, ld, C, w_oihw, bias, out, obs, ocs, ops, B, H, W);
  else hipLaunchKernelGGL((conv_small_fwd_kernel<2, 2, 3, 3>), dim3(blocks), dim3(256), 0, s, x, ld, C, w_oihw, bias, out, obs, ocs, ops, B, H, W);
  return fs_launch_status();
}

// dwpk[o][tap*cpad + c] += sum over nseg segments and pixels of dy_t[pix][o] * x_t[pix + shift][c]; dbias[o] += sum dy.
extern "C" int fsraft_conv_small_wgrad(const float* const* dy, const float* const* x, int nseg, int ldy, int ldx, int C,
                                       float* dwpk, float* dbias, int N, int B, int H, int W, int KH, int KW, hipStream_t s) {
  if (!dy || !x || !dwpk || nseg < 1 || N != 2 || C < 4 || C > 512 || C % 4 || ldx % 4 || KH != 3 || KW != 3) return FS_ERR_ARG;
  for (int base = 0; base < nseg; base += SMALL_MAX_SEG) {
    SmallWgradArgs a{};
    a.nseg = nseg - base < SMALL_MAX_SEG ? nseg - base : SMALL_MAX_SEG;
    for (int i = 0; i < a.nseg; ++i) { a.dy[i] = dy[base + i]; a.x[i] = x[base + i]; if (!a.dy[i] || !a.x[i]) return FS_ERR_ARG; }
    a.ldy = ldy; a.ldx = ldx; a.C = C; a.B = B; a.H = H; a.W = W; a.KH = KH; a.KW = KW;
    a.Ktot = KH * KW * (((C + 31) / 32) * 32); a.dwpk = dwpk; a.dbias = dbias;
    if (C <= 256) hipLaunchKernelGGL((conv_small_wgrad_kernel<1, 2, 3, 3>), dim3(512), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_small_wgrad_kernel<2, 2, 3, 3>), dim3(512), dim3(256), 0, s, a);
    const int rc = fs_launch_status();
    if (rc) return rc;
  }
  return FS_OK;
}
