// Radius-r pyramid lookup, forward and backward (rows a3 of SURVEY.md section 8;
// reference: pytorch/core/corr.py:29-50 + core/utils/utils.py:57-71, i.e. four
// grid_sample(align_corners=True, zero padding) calls, permute and cat per iteration).
//
// out[b, l*(2r+1)^2 + i*(2r+1) + j, y, x] = bilinear(V_l[q], cx/2^l + (i-r), cy/2^l + (j-r))
// with q = (b,y,x), (cx,cy) = coords[b,:,y,x]; the x offset i is the SLOW index.
//
// All 81 samples of a level share one fractional offset, so they are bilinear blends
// of a single (2r+2)x(2r+2) integer window.  A workgroup stages the windows of QB
// consecutive queries x all levels in LDS with flat, fully independent loads (every
// lane has ~50 loads in flight), then blends out of LDS.  Taps outside the level read
// as zero.  Algorithmic HBM bytes per query: L*(2r+2)^2*4 read + 8 coords + L*(2r+1)^2*4 out.
//
// Backward: coords are detached in the caller (core/raft.py:123), so only dV is
// produced.  Each query owns its slice of V, so dV accumulation across the 12
// iterations is a plain read-modify-write, no atomics.
#include "common.hpp"

namespace {

struct Pyr {
  float* p[4];
  int h[4];
  int w[4];
};

// coords[b, c, y, x] at  b*bs + c*cs + pix*ps  (covers NCHW and NHWC 2-channel tensors)
struct Coords {
  const float* p;
  int64_t bs, cs, ps;
};

template <int R, int NLEV, int QB>
struct LookupShape {
  static constexpr int N1 = 2 * R + 1, WIN = 2 * R + 2, WIN2 = WIN * WIN;
  static constexpr int CH = NLEV * N1 * N1;
  static constexpr int QLD = NLEV * WIN2 + 1;     // odd pitch: queries land on distinct banks
  static constexpr int GLD = CH + ((CH & 1) ? 0 : 1);
};

struct QInfo {
  int x0, y0;
  float fx, fy;
};

template <int R, int NLEV, int QB>
__device__ __forceinline__ void query_setup(QInfo* qi, const Coords& co, int64_t q0, int64_t nq, int HW) {
  for (int t = threadIdx.x; t < QB * NLEV; t += 256) {
    const int q = t / NLEV, l = t % NLEV;
    QInfo v{0, 0, 0.f, 0.f};
    if (q0 + q < nq) {
      const int64_t Q = q0 + q;
      const int b = (int)(Q / HW), pix = (int)(Q % HW);
      float cx = co.p[b * co.bs + pix * co.ps];
      float cy = co.p[b * co.bs + co.cs + pix * co.ps];
      const float s = 1.0f / (float)(1 << l);
      cx *= s; cy *= s;
      // anything this far out has an all-zero window; the clamp keeps floor->int defined (also NaN)
      cx = (cx > -30000.f && cx < 30000.f) ? cx : -30000.f;
      cy = (cy > -30000.f && cy < 30000.f) ? cy : -30000.f;
      const float flx = floorf(cx), fly = floorf(cy);
      v.x0 = (int)flx; v.y0 = (int)fly; v.fx = cx - flx; v.fy = cy - fly;
    }
    qi[q * NLEV + l] = v;
  }
}

template <int R, int NLEV, int QB>
__global__ __launch_bounds__(256) void corr_lookup_fwd_kernel(Pyr pyr, Coords co, float* __restrict__ out, int nhwc_out,
                                                              int64_t nq, int HW) {
  using S = LookupShape<R, NLEV, QB>;
  __shared__ float win[QB * S::QLD];
  __shared__ QInfo qi[QB * NLEV];
  const int64_t q0 = (int64_t)blockIdx.x * QB;
  query_setup<R, NLEV, QB>(qi, co, q0, nq, HW);
  __syncthreads();

  // Stage the windows.  Loads are issued in batches of UNR with nothing (no LDS store, no
  // LDS load that could alias) between them, so every lane keeps UNR global loads in flight.
  constexpr int TOTAL = QB * NLEV * S::WIN2;
  constexpr int UNR = 10;
  for (int e0 = threadIdx.x; e0 < TOTAL; e0 += 256 * UNR) {
    const float* src[UNR];
    int dst[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int e = e0 + 256 * u;
      src[u] = nullptr; dst[u] = -1;
      if (e < TOTAL) {
        const int q = e / (NLEV * S::WIN2), rem = e % (NLEV * S::WIN2);
        const int l = rem / S::WIN2, w = rem % S::WIN2;
        const int wy = w / S::WIN, wx = w % S::WIN;
        const QInfo v = qi[q * NLEV + l];
        const int gy = v.y0 - R + wy, gx = v.x0 - R + wx;
        dst[u] = q * S::QLD + rem;
        if (q0 + q < nq && gy >= 0 && gy < pyr.h[l] && gx >= 0 && gx < pyr.w[l])
          src[u] = pyr.p[l] + ((q0 + q) * pyr.h[l] + gy) * pyr.w[l] + gx;
      }
    }
    float val[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) val[u] = src[u] ? *src[u] : 0.f;
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (dst[u] >= 0) win[dst[u]] = val[u];
  }
  __syncthreads();

  for (int o = threadIdx.x; o < QB * S::CH; o += 256) {
    int q, ch;
    if (nhwc_out) { q = o / S::CH; ch = o % S::CH; } else { q = o % QB; ch = o / QB; }
    if (q0 + q >= nq) continue;
    const int l = ch / (S::N1 * S::N1), k = ch % (S::N1 * S::N1);
    const int i = k / S::N1, j = k % S::N1;       // i: x offset (slow), j: y offset (fast)
    const QInfo v = qi[q * NLEV + l];
    const float* wp = win + q * S::QLD + l * S::WIN2 + j * S::WIN + i;
    const float w00 = (1.f - v.fx) * (1.f - v.fy), w01 = v.fx * (1.f - v.fy);
    const float w10 = (1.f - v.fx) * v.fy, w11 = v.fx * v.fy;
    const float r = wp[0] * w00 + wp[1] * w01 + wp[S::WIN] * w10 + wp[S::WIN + 1] * w11;
    const int64_t Q = q0 + q;
    if (nhwc_out) {
      out[Q * S::CH + ch] = r;
    } else {
      const int64_t b = Q / HW, pix = Q % HW;
      out[(b * S::CH + ch) * HW + pix] = r;
    }
  }
}

template <int R, int NLEV, int QB>
__global__ __launch_bounds__(256) void corr_lookup_bwd_kernel(Pyr dpyr, Coords co, const float* __restrict__ dout,
                                                              int nhwc_in, int64_t nq, int HW) {
  using S = LookupShape<R, NLEV, QB>;
  __shared__ float g[QB * S::GLD];
  __shared__ QInfo qi[QB * NLEV];
  const int64_t q0 = (int64_t)blockIdx.x * QB;
  query_setup<R, NLEV, QB>(qi, co, q0, nq, HW);

  for (int o = threadIdx.x; o < QB * S::CH; o += 256) {
    int q, ch;
    if (nhwc_in) { q = o / S::CH; ch = o % S::CH; } else { q = o % QB; ch = o / QB; }
    float v = 0.f;
    const int64_t Q = q0 + q;
    if (Q < nq) {
      if (nhwc_in) v = dout[Q * S::CH + ch];
      else { const int64_t b = Q / HW, pix = Q % HW; v = dout[(b * S::CH + ch) * HW + pix]; }
    }
    g[q * S::GLD + ch] = v;
  }
  __syncthreads();

  constexpr int TOTAL = QB * NLEV * S::WIN2;
  constexpr int UNR = 10;
  for (int e0 = threadIdx.x; e0 < TOTAL; e0 += 256 * UNR) {
    float* dst[UNR];
    float d[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int e = e0 + 256 * u;
      dst[u] = nullptr; d[u] = 0.f;
      if (e >= TOTAL) continue;
      const int q = e / (NLEV * S::WIN2), rem = e % (NLEV * S::WIN2);
      const int l = rem / S::WIN2, w = rem % S::WIN2;
      const int wy = w / S::WIN, wx = w % S::WIN;
      if (q0 + q >= nq) continue;
      const QInfo v = qi[q * NLEV + l];
      const int gy = v.y0 - R + wy, gx = v.x0 - R + wx;
      if (gy < 0 || gy >= dpyr.h[l] || gx < 0 || gx >= dpyr.w[l]) continue;
      const float* gp = g + q * S::GLD + l * S::N1 * S::N1;
      float acc = 0.f;
      // window cell (wy,wx) is tap (a,c) of output (j = wy-a, i = wx-c)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int j = wy - a, i = wx - c;
          if (j >= 0 && j < S::N1 && i >= 0 && i < S::N1) {
            const float wgt = (a ? v.fy : 1.f - v.fy) * (c ? v.fx : 1.f - v.fx);
            acc += gp[i * S::N1 + j] * wgt;
          }
        }
      d[u] = acc;
      dst[u] = dpyr.p[l] + ((q0 + q) * dpyr.h[l] + gy) * dpyr.w[l] + gx;
    }
    float old[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) old[u] = dst[u] ? *dst[u] : 0.f;
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (dst[u]) *dst[u] = old[u] + d[u];
  }
}

template <int R, int QB>
int launch_fwd(const Pyr& pyr, const Coords& co, float* out, int nhwc, int64_t nq, int HW, hipStream_t s) {
  const int blocks = (int)((nq + QB - 1) / QB);
  hipLaunchKernelGGL((corr_lookup_fwd_kernel<R, 4, QB>), dim3(blocks), dim3(256), 0, s, pyr, co, out, nhwc, nq, HW);
  return fs_launch_status();
}
template <int R, int QB>
int launch_bwd(const Pyr& pyr, const Coords& co, const float* dout, int nhwc, int64_t nq, int HW, hipStream_t s) {
  const int blocks = (int)((nq + QB - 1) / QB);
  hipLaunchKernelGGL((corr_lookup_bwd_kernel<R, 4, QB>), dim3(blocks), dim3(256), 0, s, pyr, co, dout, nhwc, nq, HW);
  return fs_launch_status();
}

bool fill_pyr(Pyr& pyr, float* const* levels, int num_levels, int H, int W) {
  if (!levels || num_levels != 4) return false;
  int h = H, w = W;
  for (int l = 0; l < 4; ++l) {
    if (!levels[l] || h < 1 || w < 1) return false;
    pyr.p[l] = levels[l]; pyr.h[l] = h; pyr.w[l] = w;
    h /= 2; w /= 2;
  }
  return true;
}

}  // namespace

// coords element (b, c, pix) is read at coords[b*coords_bs + c*coords_cs + pix*coords_ps].
// out: [B, 4*(2r+1)^2, H, W] when nhwc_out == 0, [B, H, W, 4*(2r+1)^2] otherwise.
extern "C" int fsraft_corr_lookup_fwd(float* const* levels, int num_levels, const float* coords, int64_t coords_bs,
                                      int64_t coords_cs, int64_t coords_ps, float* out, int nhwc_out, int B, int H,
                                      int W, int radius, hipStream_t stream) {
  Pyr pyr;
  if (!coords || !out || B < 1 || !fill_pyr(pyr, levels, num_levels, H, W)) return FS_ERR_ARG;
  Coords co{coords, coords_bs, coords_cs, coords_ps};
  const int64_t nq = (int64_t)B * H * W;
  if (radius == 4) return launch_fwd<4, 32>(pyr, co, out, nhwc_out, nq, H * W, stream);
  if (radius == 3) return launch_fwd<3, 32>(pyr, co, out, nhwc_out, nq, H * W, stream);
  return FS_ERR_ARG;
}

// dlevels[l] += d(out)/d(V_l)^T * dout   (accumulates; caller zeroes dlevels once per step)
extern "C" int fsraft_corr_lookup_bwd(float* const* dlevels, int num_levels, const float* coords, int64_t coords_bs,
                                      int64_t coords_cs, int64_t coords_ps, const float* dout, int nhwc_in, int B,
                                      int H, int W, int radius, hipStream_t stream) {
  Pyr pyr;
  if (!coords || !dout || B < 1 || !fill_pyr(pyr, dlevels, num_levels, H, W)) return FS_ERR_ARG;
  Coords co{coords, coords_bs, coords_cs, coords_ps};
  const int64_t nq = (int64_t)B * H * W;
  if (radius == 4) return launch_bwd<4, 32>(pyr, co, dout, nhwc_in, nq, H * W, stream);
  if (radius == 3) return launch_bwd<3, 32>(pyr, co, dout, nhwc_in, nq, H * W, stream);
  return FS_ERR_ARG;
}
