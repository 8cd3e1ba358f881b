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

template <int R, int NLEV, int QB, int NT = 256>
__device__ __forceinline__ void query_setup(QInfo* qi, const Coords& co, int64_t q0, int64_t nq, int HW) {
  for (int t = threadIdx.x; t < QB * NLEV; t += NT) {
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
  constexpr int UNR = (TOTAL + 255) / 256 < 10 ? (TOTAL + 255) / 256 : 10;
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
  constexpr int UNR = (TOTAL + 255) / 256 < 10 ? (TOTAL + 255) / 256 : 10;
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

// ---------------------------------------------------------------------------------------------
// Channels-last fast path.  The flat element-per-lane staging above spends ~70 integer
// instructions per 4-byte load on index arithmetic (measured: VALU-bound at ~20 % of HBM peak).
// Here a lane owns one whole window ROW (q, level, wy): one address computation, three vector
// loads (16+16+8 bytes, dword aligned), three LDS stores.  In the blend phase a lane owns one
// output channel (its level / i / j decoded once) and walks the QB queries, so the stores of a
// wave are 256 contiguous bytes of the [query][324] output.
struct __attribute__((packed, aligned(4))) U4 { float v[4]; };
struct __attribute__((packed, aligned(4))) U2 { float v[2]; };

template <int R>
struct ClShape {
  static constexpr int N1 = 2 * R + 1, WIN = 2 * R + 2, WP = (WIN + 3) / 4 * 4;   // padded row pitch
  static constexpr int NLEV = 4, CH = NLEV * N1 * N1;
  static constexpr int QLD = NLEV * WIN * WP + 4;
};

// loads the WIN floats of one window row into r[]; fully-inside rows use vector loads
template <int WIN>
__device__ __forceinline__ void load_row(const float* __restrict__ lvl, int w, int gx0, int64_t rowbase, bool rowok,
                                         float (&r)[WIN]) {
  static_assert(WIN == 10 || WIN == 8, "radius 4 or 3");
  if (rowok && gx0 >= 0 && gx0 + WIN <= w) {
    const float* p = lvl + rowbase + gx0;
    const U4 a = *reinterpret_cast<const U4*>(p);
    const U4 b = *reinterpret_cast<const U4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = a.v[i]; r[4 + i] = b.v[i]; }
    if (WIN == 10) {
      const U2 c = *reinterpret_cast<const U2*>(p + 8);
      r[8] = c.v[0]; r[9] = c.v[1];
    }
  } else {
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
      const int gx = gx0 + i;
      r[i] = (rowok && gx >= 0 && gx < w) ? lvl[rowbase + gx] : 0.f;
    }
  }
}

template <int R, int QB, int NT>
__global__ __launch_bounds__(NT) void corr_lookup_fwd_cl_kernel(Pyr pyr, Coords co, float* __restrict__ out,
                                                                 int64_t nq, int HW) {
  using S = ClShape<R>;
  __shared__ __attribute__((aligned(16))) float win[QB * S::QLD];
  __shared__ QInfo qi[QB * S::NLEV];
  const int64_t q0 = (int64_t)blockIdx.x * QB;
  query_setup<R, S::NLEV, QB, NT>(qi, co, q0, nq, HW);
  __syncthreads();

  constexpr int ROWS = QB * S::NLEV * S::WIN;
  for (int t = threadIdx.x; t < ROWS; t += NT) {
    const int q = t / (S::NLEV * S::WIN), rem = t % (S::NLEV * S::WIN);
    const int l = rem / S::WIN, wy = rem % S::WIN;
    const QInfo v = qi[q * S::NLEV + l];
    const int gy = v.y0 - R + wy, h = pyr.h[l], w = pyr.w[l];
    const bool rowok = (q0 + q < nq) && gy >= 0 && gy < h;
    float r[S::WIN];
    load_row<S::WIN>(pyr.p[l], w, v.x0 - R, ((q0 + q) * h + (rowok ? gy : 0)) * w, rowok, r);
    float* d = win + q * S::QLD + (l * S::WIN + wy) * S::WP;
#pragma unroll
    for (int i = 0; i < S::WIN; i += 2) *reinterpret_cast<float2*>(d + i) = make_float2(r[i], r[i + 1]);
  }
  __syncthreads();

  for (int ch = threadIdx.x; ch < S::CH; ch += NT) {
    const int l = ch / (S::N1 * S::N1), k = ch % (S::N1 * S::N1);
    const int i = k / S::N1, j = k % S::N1;       // i: x offset (slow), j: y offset (fast)
    const float* wp = win + (l * S::WIN + j) * S::WP + i;
#pragma unroll 4
    for (int q = 0; q < QB; ++q) {
      if (q0 + q >= nq) break;
      const QInfo v = qi[q * S::NLEV + l];
      const float* p = wp + q * S::QLD;
      const float top = p[0] + v.fx * (p[1] - p[0]);
      const float bot = p[S::WP] + v.fx * (p[S::WP + 1] - p[S::WP]);
      // (1-fx)(1-fy) a + fx(1-fy) b + (1-fx) fy c + fx fy d, written as two lerps
      out[(q0 + q) * S::CH + ch] = top + v.fy * (bot - top);
    }
  }
}

template <int R, int QB, int NT>
__global__ __launch_bounds__(NT) void corr_lookup_bwd_cl_kernel(Pyr dpyr, Coords co, const float* __restrict__ dout,
                                                                 int64_t nq, int HW) {
  using S = ClShape<R>;
  constexpr int GLD = S::CH + 1;
  __shared__ float g[QB * GLD];
  __shared__ QInfo qi[QB * S::NLEV];
  const int64_t q0 = (int64_t)blockIdx.x * QB;
  query_setup<R, S::NLEV, QB, NT>(qi, co, q0, nq, HW);
  for (int q = 0; q < QB; ++q)
    for (int ch = threadIdx.x; ch < S::CH; ch += NT) g[q * GLD + ch] = (q0 + q < nq) ? dout[(q0 + q) * S::CH + ch] : 0.f;
  __syncthreads();

  constexpr int ROWS = QB * S::NLEV * S::WIN;
  for (int t = threadIdx.x; t < ROWS; t += NT) {
    const int q = t / (S::NLEV * S::WIN), rem = t % (S::NLEV * S::WIN);
    const int l = rem / S::WIN, wy = rem % S::WIN;
    if (q0 + q >= nq) continue;
    const QInfo v = qi[q * S::NLEV + l];
    const int gy = v.y0 - R + wy, h = dpyr.h[l], w = dpyr.w[l];
    if (gy < 0 || gy >= h) continue;
    // s[i] = dOut(i, j=wy) (1-fy) + dOut(i, j=wy-1) fy   then   d[wx] = (1-fx) s[wx] + fx s[wx-1]
    const float* gp = g + q * GLD + l * S::N1 * S::N1;
    float sv[S::N1];
#pragma unroll
    for (int i = 0; i < S::N1; ++i) {
      const float a = wy < S::N1 ? gp[i * S::N1 + wy] : 0.f;
      const float b = wy >= 1 ? gp[i * S::N1 + wy - 1] : 0.f;
      sv[i] = a * (1.f - v.fy) + b * v.fy;
    }
    float d[S::WIN];
#pragma unroll
    for (int wx = 0; wx < S::WIN; ++wx)
      d[wx] = (wx < S::N1 ? sv[wx] * (1.f - v.fx) : 0.f) + (wx >= 1 ? sv[wx - 1] * v.fx : 0.f);
    const int gx0 = v.x0 - R;
    float* row = dpyr.p[l] + ((q0 + q) * h + gy) * w;
    if (gx0 >= 0 && gx0 + S::WIN <= w) {
      float* p = row + gx0;
      U4 a = *reinterpret_cast<U4*>(p), b = *reinterpret_cast<U4*>(p + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { a.v[i] += d[i]; b.v[i] += d[4 + i]; }
      if (S::WIN == 10) {
        U2 c = *reinterpret_cast<U2*>(p + 8);
        c.v[0] += d[8]; c.v[1] += d[9];
        *reinterpret_cast<U2*>(p + 8) = c;
      }
      *reinterpret_cast<U4*>(p) = a;
      *reinterpret_cast<U4*>(p + 4) = b;
    } else {
#pragma unroll
      for (int i = 0; i < S::WIN; ++i) {
        const int gx = gx0 + i;
        if (gx >= 0 && gx < w) row[gx] += d[i];
      }
    }
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

int g_lookup_qb = 0;   // 0 auto, else 8 / 16 / 32 queries per workgroup (fsraft_set_lookup_qb)
int g_lookup_nt = 256; // threads per workgroup of the channels-last kernels (qb 208 = 8 queries, 128 threads)
int g_lookup_cl = 1;   // channels-last fast path on (qb values >= 100 select the generic kernels: qb-100)

template <int R, int QB, int NT = 256>
int launch_fwd_cl(const Pyr& pyr, const Coords& co, float* out, int64_t nq, int HW, hipStream_t s) {
  hipLaunchKernelGGL((corr_lookup_fwd_cl_kernel<R, QB, NT>), dim3((unsigned)((nq + QB - 1) / QB)), dim3(NT), 0, s, pyr, co, out, nq, HW);
  return fs_launch_status();
}
template <int R, int QB, int NT = 256>
int launch_bwd_cl(const Pyr& pyr, const Coords& co, const float* dout, int64_t nq, int HW, hipStream_t s) {
  hipLaunchKernelGGL((corr_lookup_bwd_cl_kernel<R, QB, NT>), dim3((unsigned)((nq + QB - 1) / QB)), dim3(NT), 0, s, pyr, co, dout, nq, HW);
  return fs_launch_status();
}

template <int R>
int dispatch_fwd(int qb, const Pyr& pyr, const Coords& co, float* out, int nhwc, int64_t nq, int HW, hipStream_t s) {
  if (nhwc && g_lookup_cl) {
    if (qb == 8 && g_lookup_nt == 128) return launch_fwd_cl<R, 8, 128>(pyr, co, out, nq, HW, s);
    if (qb == 8 && g_lookup_nt == 320) return launch_fwd_cl<R, 8, 320>(pyr, co, out, nq, HW, s);
    if (qb == 8) return launch_fwd_cl<R, 8>(pyr, co, out, nq, HW, s);
    if (qb == 32) return launch_fwd_cl<R, 32>(pyr, co, out, nq, HW, s);
    return launch_fwd_cl<R, 16>(pyr, co, out, nq, HW, s);
  }
  if (qb == 8) return launch_fwd<R, 8>(pyr, co, out, nhwc, nq, HW, s);
  if (qb == 16) return launch_fwd<R, 16>(pyr, co, out, nhwc, nq, HW, s);
  return launch_fwd<R, 32>(pyr, co, out, nhwc, nq, HW, s);
}
template <int R>
int dispatch_bwd(int qb, const Pyr& pyr, const Coords& co, const float* dout, int nhwc, int64_t nq, int HW, hipStream_t s) {
  if (nhwc && g_lookup_cl) {
    if (qb == 8 && g_lookup_nt == 128) return launch_bwd_cl<R, 8, 128>(pyr, co, dout, nq, HW, s);
    if (qb == 8 && g_lookup_nt == 320) return launch_bwd_cl<R, 8, 320>(pyr, co, dout, nq, HW, s);
    if (qb == 8) return launch_bwd_cl<R, 8>(pyr, co, dout, nq, HW, s);
    if (qb == 32) return launch_bwd_cl<R, 32>(pyr, co, dout, nq, HW, s);
    return launch_bwd_cl<R, 16>(pyr, co, dout, nq, HW, s);
  }
  if (qb == 8) return launch_bwd<R, 8>(pyr, co, dout, nhwc, nq, HW, s);
  if (qb == 16) return launch_bwd<R, 16>(pyr, co, dout, nhwc, nq, HW, s);
  return launch_bwd<R, 32>(pyr, co, dout, nhwc, nq, HW, s);
}
// channels-last output is contiguous per query, so small workgroups cost no coalescing and let
// several thousand of them be resident at once; planar (NCHW) output wants 32 queries per row segment
inline int pick_qb(int nhwc) { return g_lookup_qb ? g_lookup_qb : (nhwc ? 8 : 32); }

// ceil: level sizes of TensorFlow's padding='SAME' pooling (ceil halving) instead of avg_pool2d's floor halving
bool fill_pyr(Pyr& pyr, float* const* levels, int num_levels, int H, int W, bool ceil = false) {
  if (!levels || num_levels != 4) return false;
  for (int l = 0; l < 4; ++l) {
    const int h = ceil ? (H + (1 << l) - 1) >> l : H >> l, w = ceil ? (W + (1 << l) - 1) >> l : W >> l;
    if (!levels[l] || h < 1 || w < 1) return false;
    pyr.p[l] = levels[l]; pyr.h[l] = h; pyr.w[l] = w;
  }
  return true;
}

}  // namespace

extern "C" int fsraft_set_lookup_qb(int qb) {
  g_lookup_nt = 256;
  if (qb >= 300) { g_lookup_nt = 320; qb -= 300; }      // 8 queries x 4 levels x 10 rows = 320 row loads: one per thread
  if (qb >= 200) { g_lookup_nt = 128; qb -= 200; }
  g_lookup_cl = qb < 100;
  if (qb >= 100) qb -= 100;
  if (qb != 0 && qb != 8 && qb != 16 && qb != 32) return FS_ERR_ARG;
  g_lookup_qb = qb;
  return FS_OK;
}

// coords element (b, c, pix) is read at coords[b*coords_bs + c*coords_cs + pix*coords_ps].
// out: [B, 4*(2r+1)^2, H, W] when nhwc_out == 0, [B, H, W, 4*(2r+1)^2] otherwise.
extern "C" int fsraft_corr_lookup_fwd(float* const* levels, int num_levels, const float* coords, int64_t coords_bs,
                                      int64_t coords_cs, int64_t coords_ps, float* out, int nhwc_out, int B, int H,
                                      int W, int radius, hipStream_t stream) {
  Pyr pyr;
  if (!coords || !out || B < 1 || !fill_pyr(pyr, levels, num_levels, H, W)) return FS_ERR_ARG;
  Coords co{coords, coords_bs, coords_cs, coords_ps};
  const int64_t nq = (int64_t)B * H * W;
  if (radius == 4) return dispatch_fwd<4>(pick_qb(nhwc_out), pyr, co, out, nhwc_out, nq, H * W, stream);
  if (radius == 3) return dispatch_fwd<3>(pick_qb(nhwc_out), pyr, co, out, nhwc_out, nq, H * W, stream);
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
  if (radius == 4) return dispatch_bwd<4>(pick_qb(nhwc_in), pyr, co, dout, nhwc_in, nq, H * W, stream);
  if (radius == 3) return dispatch_bwd<3>(pick_qb(nhwc_in), pyr, co, dout, nhwc_in, nq, H * W, stream);
  return FS_ERR_ARG;
}

// Forward lookup on a pyramid whose levels have TensorFlow 'SAME' sizes (level l: ceil(H / 2^l) x ceil(W / 2^l)); everything
// else as fsraft_corr_lookup_fwd.  (raft/allfield.py:109-135 on the pyramid of raft/allfield.py:94-106.)
extern "C" int fsraft_corr_lookup_fwd_same(float* const* levels, int num_levels, const float* coords, int64_t coords_bs,
                                           int64_t coords_cs, int64_t coords_ps, float* out, int nhwc_out, int B, int H,
                                           int W, int radius, hipStream_t stream) {
  Pyr pyr;
  if (!coords || !out || B < 1 || !fill_pyr(pyr, levels, num_levels, H, W, true)) return FS_ERR_ARG;
  Coords co{coords, coords_bs, coords_cs, coords_ps};
  const int64_t nq = (int64_t)B * H * W;
  if (radius == 4) return dispatch_fwd<4>(pick_qb(nhwc_out), pyr, co, out, nhwc_out, nq, H * W, stream);
  if (radius == 3) return dispatch_fwd<3>(pick_qb(nhwc_out), pyr, co, out, nhwc_out, nq, H * W, stream);
  return FS_ERR_ARG;
}
