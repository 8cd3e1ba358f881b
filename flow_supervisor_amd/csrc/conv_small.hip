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
// One wave walks a RUN of consecutive pixels of an image row and keeps the 3 x 3 window of channel rows in registers: moving one
// pixel to the right shifts the window's columns and loads ONE new column (3 rows of C channels) instead of nine rows --
// measured before this, both kernels here ran at the rate L2 delivers nine 1 KB rows per pixel (5.7-5.9 TB/s).
constexpr int SMALL_RUN = 16;        // weight gradient: long runs (12 segments x 1760 runs keep 2048 waves busy)
constexpr int SMALL_RUN_FWD = 8;     // forward: the launch is latency-bound, more and shorter runs measured faster (strided pixels 43.9 us, runs of 16: 37.3, two runs of 16 per wave: 51.5, runs of 8: 34.5)
constexpr int SMALL_CHUNK = 8;       // weight gradient: pixels whose columns are requested together

// Buffer loads with the validity in the lane offset: `ok ? *p : 0` compiles to one branch per load, which serialises the
// loads of a run (the forward kernel was bound by exactly that: one L2 round trip per pixel, 34 us per launch).
constexpr unsigned SM_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sm_rsrc(const void* p) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uint64_t)hi << 32 | lo), 0, 0x7fffffff, 0x00020000);
}

template <int KC, int NOUT, int KH, int KW>
__global__ __launch_bounds__(256) void conv_small_fwd_kernel(const float* __restrict__ x, int ld, int C,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ out, int64_t obs, int64_t ocs, int64_t ops,
                                                             int B, int H, int W) {
  static_assert(KH == 3 && KW == 3, "sliding window written for 3 x 3");
  const int lane = threadIdx.x & 63;
  constexpr int taps = 9;
  const int HW = H * W;
  // weights: w[o][c][tap] (OIHW) -> registers wr[o][tap][kc] as float4 over this lane's channels
  // (a lane's 4 channels x 9 taps are 36 consecutive floats of the OIHW tensor: nine 16-byte loads, transposed in registers)
  f32x4 wr[NOUT][taps][KC];
#pragma unroll
  for (int o = 0; o < NOUT; ++o)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int c = kc * 256 + lane * 4;
      float buf[36];
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const f32x4 v = c < C ? *reinterpret_cast<const f32x4*>(w + ((int64_t)o * C + c) * taps + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        buf[4 * q] = v[0]; buf[4 * q + 1] = v[1]; buf[4 * q + 2] = v[2]; buf[4 * q + 3] = v[3];
      }
#pragma unroll
      for (int t = 0; t < taps; ++t) wr[o][t][kc] = f32x4{buf[t], buf[9 + t], buf[18 + t], buf[27 + t]};
    }
  const int rpr = (W + SMALL_RUN_FWD - 1) / SMALL_RUN_FWD;                 // runs per image row
  const int64_t nrun = (int64_t)B * H * rpr;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int64_t run = wave; run < nrun; run += nwaves) {
    const int x0 = (int)(run % rpr) * SMALL_RUN_FWD;
    const int64_t by = run / rpr;
    const int y = (int)(by % H), b = (int)(by / H);
    const int n = W - x0 < SMALL_RUN_FWD ? W - x0 : SMALL_RUN_FWD;
    const float* img = x + (int64_t)b * HW * ld;
    // column xs of the three rows y - 1 .. y + 1 (zeros outside the image).  The whole run's columns are requested before
    // the first one is used.
    const __amdgpu_buffer_rsrc_t rs = sm_rsrc(img);
    auto load_col = [&](int xs, f32x4 (&col)[3][KC]) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        const bool in = (unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int c = kc * 256 + lane * 4;
          col[r][kc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (in && c < C) ? (unsigned)((yy * W + xs) * ld + c) * 4u : SM_OOB, 0, 0));
        }
      }
    };
    f32x4 c0[3][KC], c1[3][KC], cn[SMALL_RUN_FWD][3][KC];
    load_col(x0 - 1, c0);
    load_col(x0, c1);
#pragma unroll
    for (int i = 0; i < SMALL_RUN_FWD; ++i) load_col(x0 + i + 1, cn[i]);
#pragma unroll
    for (int i = 0; i < SMALL_RUN_FWD; ++i) {
      if (i < n) {
        float acc[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          f32x4 a4 = zero;
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
              a4 += c0[r][kc] * wr[o][3 * r + 0][kc];
              a4 += c1[r][kc] * wr[o][3 * r + 1][kc];
              a4 += cn[i][r][kc] * wr[o][3 * r + 2][kc];
            }
          acc[o] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
#pragma unroll
          for (int sft = 32; sft > 0; sft >>= 1) acc[o] += __shfl_xor(acc[o], sft, 64);
        }
        if (lane == 0) {
          const int pix = y * W + x0 + i;
#pragma unroll
          for (int o = 0; o < NOUT; ++o) out[b * obs + o * ocs + pix * ops] = acc[o] + (bias ? bias[o] : 0.f);
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) { c0[r][kc] = c1[r][kc]; c1[r][kc] = cn[i][r][kc]; }
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
  constexpr int taps = KH * KW;
  const int HW = a.H * a.W;
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
  static_assert(KH == 3 && KW == 3, "sliding window written for 3 x 3");
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
  const auto* karg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  typedef const float* fptr;
  const int rpr = (a.W + SMALL_RUN - 1) / SMALL_RUN;
  const int64_t nrun = (int64_t)a.B * a.H * rpr;
  for (int seg = 0; seg < a.nseg; ++seg) {      // pointer tables read from the kernarg segment (uniform index)
    const float* dy = ((const fptr __attribute__((address_space(4)))*)(karg + offsetof(SmallWgradArgs, dy)))[seg];
    const float* x = ((const fptr __attribute__((address_space(4)))*)(karg + offsetof(SmallWgradArgs, x)))[seg];
    for (int64_t run = wave; run < nrun; run += nwaves) {
      const int x0 = (int)(run % rpr) * SMALL_RUN;
      const int64_t by = run / rpr;
      const int y = (int)(by % a.H), b = (int)(by / a.H);
      const int n = a.W - x0 < SMALL_RUN ? a.W - x0 : SMALL_RUN;
      const float* img = x + (int64_t)b * HW * a.ldx;
      const __amdgpu_buffer_rsrc_t rs = sm_rsrc(img);
      auto load_col = [&](int xs, f32x4 (&col)[3][KC]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int yy = y + r - 1;
          const bool in = (unsigned)yy < (unsigned)a.H && (unsigned)xs < (unsigned)a.W;
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) {
            const int c = kc * 256 + lane * 4;
            col[r][kc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (in && c < a.C) ? (unsigned)((yy * a.W + xs) * a.ldx + c) * 4u : SM_OOB, 0, 0));
          }
        }
      };
      f32x4 c0[3][KC], c1[3][KC];
      load_col(x0 - 1, c0);
      load_col(x0, c1);
      for (int i0 = 0; i0 < n; i0 += SMALL_CHUNK) {          // eight pixels at a time: their columns and dY values requested together
        f32x4 cn[SMALL_CHUNK][3][KC];
        float g[SMALL_CHUNK][NOUT];
#pragma unroll
        for (int i = 0; i < SMALL_CHUNK; ++i) {
          load_col(x0 + i0 + i + 1, cn[i]);
          const int64_t m = (int64_t)b * HW + (int64_t)y * a.W + x0 + i0 + (i0 + i < n ? i : 0);
#pragma unroll
          for (int o = 0; o < NOUT; ++o) g[i][o] = i0 + i < n ? dy[m * a.ldy + o] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < SMALL_CHUNK; ++i) {
#pragma unroll
          for (int o = 0; o < NOUT; ++o) bsum[o] += g[i][o];
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
              for (int o = 0; o < NOUT; ++o) {
                acc[o][3 * r + 0][kc] += g[i][o] * c0[r][kc];
                acc[o][3 * r + 1][kc] += g[i][o] * c1[r][kc];
                acc[o][3 * r + 2][kc] += g[i][o] * cn[i][r][kc];
              }
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) { c0[r][kc] = c1[r][kc]; c1[r][kc] = cn[i][r][kc]; }
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

// Data gradient of the same 3x3 convolution to two outputs (the flow head's second layer, pytorch/core/update.py:6-14):
//     dx[pix][c] = mask(pix, c) * sum_{o < 2, tap} w[o][c][tap] * dy[pix - shift(tap)][o]
// 18 multiply-adds per element and 1 KB written per pixel: a streaming kernel, not a GEMM -- as an implicit GEMM the two
// input channels are padded to a 32-deep k-tile per tap (K = 288 for 18 useful products) and the launch took 331 us for the
// twelve iterations of a four-pair step.  One wave owns a run of pixels of an image row; a lane owns four channels and keeps
// their 72 weights in registers; the 3 x 3 x 2 gradients around a pixel arrive through wave-uniform (scalar) loads, sliding
// along the run one column at a time.  mask (optional): the ReLU that produced x -- dx is written only where relu_src > 0.
constexpr int SMALL_RUN_DG = 16;
template <int N>
__global__ __launch_bounds__(256) void conv_small_dgrad_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ w_oihw,
                                                               float* __restrict__ dx, int lddx, const float* __restrict__ relu_src,
                                                               int ldm, int C, int B, int H, int W, unsigned* __restrict__ dx_amax) {
  static_assert(N == 2, "two output channels");
  unsigned amx = 0u;
  const int lane = threadIdx.x & 63;
  const int c0 = lane * 4;
  const bool on = c0 < C;
  // this lane's weights: w[o][c0 .. c0 + 3][ky][kx] (rows of 9 floats per (o, c))
  float wr[N][4][9];
#pragma unroll
  for (int o = 0; o < N; ++o)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int t = 0; t < 9; ++t) wr[o][k][t] = (on && c0 + k < C) ? w_oihw[((int64_t)o * C + c0 + k) * 9 + t] : 0.f;
  const int runs_per_row = (W + SMALL_RUN_DG - 1) / SMALL_RUN_DG;
  const int64_t nrun = (int64_t)B * H * runs_per_row;
  const int nwave = gridDim.x * 4;
  for (int64_t run = blockIdx.x * 4 + (threadIdx.x >> 6); run < nrun; run += nwave) {
    const int64_t ru = __builtin_amdgcn_readfirstlane((int)run);       // (nrun < 2^31: checked by the host)
    const int xr = (int)(ru % runs_per_row), row = (int)(ru / runs_per_row), y = row % H, b = row / H;
    const int x0 = xr * SMALL_RUN_DG, x1 = min(W, x0 + SMALL_RUN_DG);
    const float* dyb = dy + (int64_t)b * H * W * ldy;
    // dy[o] at (y + j - 1, x + i - 1): a 3-column window that slides along the run (wave-uniform values)
    float g[3][3][N];
    auto col = [&](int x, float (&gc)[3][N]) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int yy = y + j - 1;
        const bool ok = (unsigned)yy < (unsigned)H && (unsigned)x < (unsigned)W;
        const float* p = dyb + (int64_t)((ok ? yy : y) * W + (ok ? x : x0)) * ldy;
#pragma unroll
        for (int o = 0; o < N; ++o) gc[j][o] = ok ? p[o] : 0.f;
      }
    };
    col(x0 - 1, g[0]);
    col(x0, g[1]);
    for (int x = x0; x < x1; ++x) {
      col(x + 1, g[2]);
      if (on) {
        const int64_t pix = ((int64_t)b * H + y) * W + x;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // forward: y[q] = sum w[ky][kx] x[q + (ky - 1, kx - 1)]  =>  dx[p] = sum w[ky][kx] dy[p - (ky - 1, kx - 1)]: window cell (2 - ky, 2 - kx)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int o = 0; o < N; ++o) {
              const float gv = g[2 - kx][2 - ky][o];
#pragma unroll
              for (int k = 0; k < 4; ++k) acc[k] += wr[o][k][ky * 3 + kx] * gv;
            }
        if (relu_src) {
          const f32x4 m = gload4(relu_src + pix * ldm + c0);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = m[k] > 0.f ? acc[k] : 0.f;
        }
        gstore4(dx + pix * lddx + c0, acc);
        amx = fs_umax(amx, fs_abs_bits4(acc));
      }
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int o = 0; o < N; ++o) { g[0][j][o] = g[1][j][o]; g[1][j][o] = g[2][j][o]; }
    }
    if (dx_amax && run < nwave) fs_amax_early(dx_amax, amx);   // (first run of the wave)
  }
  if (dx_amax) fs_amax_commit_wave(dx_amax, amx);            // (nullable) word of dx, raised
}

}  // namespace

// out[b][o][pix] = bias[o] + sum_{c,tap} w[o][c][tap] * x[pixel + shift(tap)][c];  N = 2 outputs, C <= 512, KH*KW <= 9.
extern "C" int fsraft_conv_small_fwd(const float* x, int ld, int C, const float* w_oihw, const float* bias, float* out,
                                     int64_t obs, int64_t ocs, int64_t ops, int N, int B, int H, int W, int KH, int KW,
                                     hipStream_t s) {
  if (!x || !w_oihw || !out || N != 2 || C < 4 || C > 512 || C % 4 || ld % 4 || KH != 3 || KW != 3 || B < 1 || ((uintptr_t)w_oihw % 16)) return FS_ERR_ARG;
  const int64_t M = (int64_t)B * H * W;
  const int64_t nrun = (int64_t)B * H * ((W + SMALL_RUN_FWD - 1) / SMALL_RUN_FWD);
  int blocks = (int)((nrun + 3) / 4);                 // one run per wave
  if (blocks > 2048) blocks = 2048;
  (void)M;
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

// dx[pix][c] (channels-last, pitch lddx, 16-byte aligned, c < C) = mask * sum_{o, tap} w_oihw[o][c][tap] * dy[pix - shift(tap)][o]: the data
// gradient of fsraft_conv_small_fwd.  dy: channels-last with pitch ldy >= 2 (the two gradients of a pixel side by side);
// relu_src (nullable, pitch ldm): dx is zero where relu_src <= 0 (the ReLU in front of the convolution).  C % 4 == 0, C <= 256.
extern "C" int fsraft_conv_small_dgrad(const float* dy, int ldy, const float* w_oihw, float* dx, int lddx, const float* relu_src, int ldm,
                                       int C, int N, int B, int H, int W, int KH, int KW, unsigned* dx_amax, hipStream_t s) {
  if (((uintptr_t)dx_amax & 3)) return FS_ERR_ARG;
  if (!dy || !w_oihw || !dx || N != 2 || C < 4 || C > 256 || C % 4 || lddx % 4 || ldy < 2 || KH != 3 || KW != 3 || B < 1 ||
      ((uintptr_t)dx % 16) || (relu_src && (ldm % 4 || ((uintptr_t)relu_src % 16))))
    return FS_ERR_ARG;
  const int64_t nrun = (int64_t)B * H * ((W + SMALL_RUN_DG - 1) / SMALL_RUN_DG);
  if (nrun >= ((int64_t)1 << 31)) return FS_ERR_ARG;
  int blocks = (int)((nrun + 3) / 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((conv_small_dgrad_kernel<2>), dim3(blocks), dim3(256), 0, s, dy, ldy, w_oihw, dx, lddx, relu_src, ldm, C, B, H, W, dx_amax);
  return fs_launch_status();
}
