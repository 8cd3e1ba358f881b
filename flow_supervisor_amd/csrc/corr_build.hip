// All-pairs correlation volume + average-pool pyramid in one pass (rows a1, a2 of
// SURVEY.md section 8; reference: pytorch/core/corr.py:13-27 and :52-60).
//
//   V0[b, i, j] = (1/sqrt(C)) * sum_c f1[b, c, i] * f2[b, c, j]
//   V(l+1)      = 2x2 mean (floor) of V(l) over the (h2, w2) plane of j
//
// One workgroup computes 64 queries (i) x one 8x32 patch of targets (j) with the
// fp32 MFMA core, then pools the patch 2x2 / 4x4 / 8x8 out of LDS, so every level is
// written exactly once and level 0 is never re-read.  Level-l cells are aligned to
// 2^l, hence an 8-row x 32-col patch contains whole cells of every level; a cell is
// emitted iff its index is below the floor-halved size of its level, which is exactly
// the set avg_pool2d(2, stride=2) keeps.
//
// HBM traffic per sample = read 2*N*C*4 + write N*P*4 (P = sum_l h_l*w_l): 275.7 MB at
// 55x128, C=256.  FLOPs 2*N*N*C = 25.4 G => fp32-MFMA bound.
#include "gemm_core.hpp"
#include "gemm_core_split.hpp"
#include "corr_layout.hpp"
#include "gemm_rec.hpp"

namespace {

using BuildCfg = GemmCfg<64, 256, 16, 1, 4, 0, 0>;

struct F1Loader {            // As[k][i] <- f1[b][c = kt*16 + k][i0 + i]
  static constexpr int NREG = 4, NCH = 4;
  const float* f1b; int N; int i0; int C;
  __device__ __forceinline__ bool fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int i = threadIdx.x & 63, k0 = threadIdx.x >> 6;
    const int c = kt * 16 + k0 + 4 * j;
    const bool ok = i0 + i < N && c < C;
    r[j] = f1b[ok ? (int64_t)c * N + i0 + i : 0];
    return ok;
  }
  __device__ __forceinline__ void store_chunk(float* t, const float (&r)[NREG], int j, bool ok) const {
    const int i = threadIdx.x & 63, k0 = threadIdx.x >> 6;
    t[(k0 + 4 * j) * BuildCfg::LDA + i] = ok ? r[j] : 0.f;
  }
};

// patch position (row 0..7, col 0..31) -> column n of the block tile such that wave
// wn = col/8 owns an 8x8 sub-patch and MFMA n-tile nt = row/4 owns 4 of its rows.
__device__ __forceinline__ int patch_to_n(int row, int col) {
  return (col >> 3) * 64 + (row >> 2) * 32 + (row & 3) * 8 + (col & 7);
}

struct F2Loader {            // Bs[k][n(row,col)] <- f2[b][c][(py0+row)*W + px0+col]
  static constexpr int NREG = 16, NCH = 8;
  const float* f2b; int N; int C; int pix; bool ok;
  __device__ __forceinline__ bool fetch_chunk(int kt, float (&r)[NREG], int j) const {
    // both channels of a chunk share one flag: C is required to be even (checked on the host)
    const bool okc = ok && kt * 16 + 2 * j < C;
#pragma unroll
    for (int k = 2 * j; k < 2 * j + 2; ++k) r[k] = f2b[okc ? (int64_t)(kt * 16 + k) * N + pix : 0];
    return okc;
  }
  __device__ __forceinline__ void store_chunk(float* t, const float (&r)[NREG], int j, bool ok) const {
    const int n = patch_to_n(threadIdx.x >> 5, threadIdx.x & 31);
#pragma unroll
    for (int k = 2 * j; k < 2 * j + 2; ++k) t[k * BuildCfg::LDB + n] = ok ? r[k] : 0.f;
  }
};

struct __attribute__((packed, aligned(4))) PF4 { float v[4]; };

struct Levels {
  float* p[4];
  int h[4];
  int w[4];
};

constexpr int S_LD = 256;
constexpr int EPI_FLOATS = 32 * 256 + 32 * 64 + 32 * 16;
constexpr int LDS_FLOATS = BuildCfg::LDS_FLOATS > EPI_FLOATS ? BuildCfg::LDS_FLOATS : EPI_FLOATS;

// Shared epilogue: park the 64 x (8x32 patch) tile in LDS half by half, store level 0 as 128-byte row
// segments and pool 2x2 / 4x4 / 8x8 out of LDS.
__device__ __forceinline__ void build_epilogue(f32x16 (&acc)[2][2], float* lds, const Levels& lv, int nlev, int H, int W,
                                               int N, int b, int i0, int px0, int py0, float scale) {
  float* S = lds;                 // [32][256]  scaled level-0 patch, patch-linear (row*32+col)
  float* P1 = lds + 32 * 256;     // [32][4*16]
  float* P2 = P1 + 32 * 64;       // [32][2*8]
  const int lane = threadIdx.x & 63, wn = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;

#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int pj = (4 * nt + (l31 >> 3)) * 32 + 8 * wn + (l31 & 7);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        S[i * S_LD + pj] = acc[mt][nt][r] * scale;
      }
    }
    __syncthreads();
    const int qbase = i0 + mt * 32;
    // level 0: 32 queries x 8 rows of 32 contiguous floats; 16 bytes per lane (dword-aligned vector
    // stores: rows start at multiples of W floats, which need not be 16-byte aligned)
    {
      const int c4 = threadIdx.x & 7, slot = threadIdx.x >> 3;       // 8 lanes per 128-byte row segment
      const int x = px0 + 4 * c4;
#pragma unroll 4
      for (int jj = 0; jj < 8; ++jj) {
        const int idx = slot + 32 * jj;
        const int i = idx >> 3, row = idx & 7;
        if (py0 + row < H && qbase + i < N && x < W) {
          float* dst = lv.p[0] + (((int64_t)b * N + qbase + i) * H + py0 + row) * W + x;
          const float* src = S + i * S_LD + row * 32 + 4 * c4;
          if (x + 3 < W) {
            PF4 v; v.v[0] = src[0]; v.v[1] = src[1]; v.v[2] = src[2]; v.v[3] = src[3];
            *reinterpret_cast<PF4*>(dst) = v;
          } else {
            for (int q = 0; q < 4 && x + q < W; ++q) dst[q] = src[q];
          }
        }
      }
    }
    if (nlev > 1) {
      const int h1 = lv.h[1], w1 = lv.w[1];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int e = threadIdx.x + 256 * jj;
        const int i = e >> 6, c = e & 63, y = c >> 4, x = c & 15;
        const float* s = S + i * S_LD + (2 * y) * 32 + 2 * x;
        const float v = (((s[0] + s[1]) + s[32]) + s[33]) * 0.25f;
        P1[i * 64 + c] = v;
        const int gy = (py0 >> 1) + y, gx = (px0 >> 1) + x;
        if (gy < h1 && gx < w1 && qbase + i < N) lv.p[1][(((int64_t)b * N + qbase + i) * h1 + gy) * w1 + gx] = v;
      }
    }
    __syncthreads();
    if (nlev > 2) {
      const int h2 = lv.h[2], w2 = lv.w[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int e = threadIdx.x + 256 * jj;
        const int i = e >> 4, c = e & 15, y = c >> 3, x = c & 7;
        const float* s = P1 + i * 64 + (2 * y) * 16 + 2 * x;
        const float v = (((s[0] + s[1]) + s[16]) + s[17]) * 0.25f;
        P2[i * 16 + c] = v;
        const int gy = (py0 >> 2) + y, gx = (px0 >> 2) + x;
        if (gy < h2 && gx < w2 && qbase + i < N) lv.p[2][(((int64_t)b * N + qbase + i) * h2 + gy) * w2 + gx] = v;
      }
    }
    __syncthreads();
    if (nlev > 3 && threadIdx.x < 128) {
      const int h3 = lv.h[3], w3 = lv.w[3];
      const int i = threadIdx.x >> 2, x = threadIdx.x & 3;
      const float* s = P2 + i * 16 + 2 * x;
      const float v = (((s[0] + s[1]) + s[8]) + s[9]) * 0.25f;
      const int gy = (py0 >> 3), gx = (px0 >> 3) + x;
      if (gy < h3 && gx < w3 && qbase + i < N) lv.p[3][(((int64_t)b * N + qbase + i) * h3 + gy) * w3 + gx] = v;
    }
    __syncthreads();
  }
}

// Tiled-row variant of the epilogue (corr_layout.hpp): the same 64 x (8x32 patch) tile, written as whole 64-byte tiles
// of the query's row.  The patch is aligned to 8 rows / 32 columns, so it holds 2x8 whole level-0 tiles (512 contiguous
// bytes per tile row and query), 1x4 level-1 tiles, two rows of 2 level-2 tiles and one row of 1 level-3 tile.
__device__ __forceinline__ void build_epilogue_tiled(f32x16 (&acc)[2][2], float* lds, float* __restrict__ vol, const VolLayout& L,
                                                     int N, int b, int i0, int px0, int py0, float scale) {
  float* S = lds;                 // [32][256]  scaled level-0 patch, patch-linear (row*32+col)
  float* P1 = lds + 32 * 256;     // [32][4*16]
  float* P2 = P1 + 32 * 64;       // [32][2*8]
  const int lane = threadIdx.x & 63, wn = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nlev = L.nlev;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int pj = (4 * nt + (l31 >> 3)) * 32 + 8 * wn + (l31 & 7);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        S[i * S_LD + pj] = acc[mt][nt][r] * scale;
      }
    }
    __syncthreads();
    const int qbase = i0 + mt * 32;
    float* rows = vol + ((int64_t)b * N + qbase) * L.P;
    // level 0: chunk = (query i, tile row ty, tile t, row r of the tile), 16 bytes; 32 lanes cover one 512-byte run
#pragma unroll 4
    for (int jj = 0; jj < 8; ++jj) {
      const int id = threadIdx.x + 256 * jj;
      const int r = id & 3, t = (id >> 2) & 7, ty = (id >> 5) & 1, i = id >> 6;
      const int gty = (py0 >> 2) + ty, gtx = (px0 >> 2) + t;
      if (gty < L.th[0] && gtx < L.tw[0] && qbase + i < N) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(S + i * S_LD + (4 * ty + r) * 32 + 4 * t);
        gstore4(rows + (int64_t)i * L.P + L.off[0] + (gty * L.tw[0] + gtx) * 16 + r * 4, v);
      }
    }
    if (nlev > 1) {
      const int h1 = L.h[1], w1 = L.w[1];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int e = threadIdx.x + 256 * jj;
        const int i = e >> 6, c = e & 63, y = c >> 4, x = c & 15;
        const float* s = S + i * S_LD + (2 * y) * 32 + 2 * x;
        const float v = (((s[0] + s[1]) + s[32]) + s[33]) * 0.25f;
        // cells beyond the floor-halved size do not exist in the reference pyramid: pad cells of the tile hold 0
        P1[i * 64 + c] = ((py0 >> 1) + y < h1 && (px0 >> 1) + x < w1) ? v : 0.f;
      }
    }
    __syncthreads();
    if (nlev > 1) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int id = threadIdx.x + 256 * jj;
        const int r = id & 3, t = (id >> 2) & 3, i = id >> 4;
        const int gty = py0 >> 3, gtx = (px0 >> 3) + t;
        if (gty < L.th[1] && gtx < L.tw[1] && qbase + i < N) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(P1 + i * 64 + r * 16 + 4 * t);
          gstore4(rows + (int64_t)i * L.P + L.off[1] + (gty * L.tw[1] + gtx) * 16 + r * 4, v);
        }
      }
    }
    if (nlev > 2) {
      const int h2 = L.h[2], w2 = L.w[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int e = threadIdx.x + 256 * jj;
        const int i = e >> 4, c = e & 15, y = c >> 3, x = c & 7;
        const float* s = P1 + i * 64 + (2 * y) * 16 + 2 * x;
        const float v = (((s[0] + s[1]) + s[16]) + s[17]) * 0.25f;
        P2[i * 16 + c] = ((py0 >> 2) + y < h2 && (px0 >> 2) + x < w2) ? v : 0.f;
      }
    }
    __syncthreads();
    if (nlev > 2 && threadIdx.x < 128) {
      const int id = threadIdx.x;
      const int y = id & 1, t = (id >> 1) & 1, i = id >> 2;
      const int gy = (py0 >> 2) + y, gtx = (px0 >> 4) + t;
      if ((gy >> 2) < L.th[2] && gtx < L.tw[2] && qbase + i < N) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(P2 + i * 16 + y * 8 + 4 * t);
        gstore4(rows + (int64_t)i * L.P + L.off[2] + ((gy >> 2) * L.tw[2] + gtx) * 16 + (gy & 3) * 4, v);
      }
    }
    if (nlev > 3 && threadIdx.x >= 128) {
      const int h3 = L.h[3], w3 = L.w[3];
      const int id = threadIdx.x - 128;
      const int i = id >> 2, x = id & 3;
      const float* s = P2 + i * 16 + 2 * x;
      const float v = (((s[0] + s[1]) + s[8]) + s[9]) * 0.25f;
      const int gy = (py0 >> 3), gx = (px0 >> 3) + x;
      if ((gy >> 2) < L.th[3] && (gx >> 2) < L.tw[3] && qbase + i < N)
        gstore1(rows + (int64_t)i * L.P + vol_cell(L, 3, gy, gx), (gy < h3 && gx < w3) ? v : 0.f);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void corr_build_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                         Levels lv, int nlev, int C, int H, int W, float scale) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  const int N = H * W;
  const int npx = ceil_div_dev(W, 32);
  const int px0 = (blockIdx.x % npx) * 32, py0 = (blockIdx.x / npx) * 8;
  const int i0 = blockIdx.y * 64;
  const int b = blockIdx.z;

  F1Loader la{f1 + (int64_t)b * C * N, N, i0, C};
  const int prow = threadIdx.x >> 5, pcol = threadIdx.x & 31;
  const bool pok = (py0 + prow < H) && (px0 + pcol < W);
  F2Loader lb{f2 + (int64_t)b * C * N, N, C, (py0 + prow) * W + px0 + pcol, pok};

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  gemm_mainloop<BuildCfg>(lds, ceil_div_dev(C, 16), la, lb, acc);

  build_epilogue(acc, lds, lv, nlev, H, W, N, b, i0, px0, py0, scale);
}

// ---- split-bf16 build: both operands are k-major ([channel][pixel]), i.e. the transposed-read path ----
using BuildSplitCfg = SplitTnCfg<64, 256, 1, 4, 1>;

struct F1SplitLoader {        // chunk e: channel k = e / 16, queries i0 + 4*(e % 16) .. +3
  static constexpr int NCH = BuildSplitCfg::NCH_A, NREG = NCH * 4;
  const float* f1b; int N, i0, C;
  __device__ __forceinline__ void fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int c = kt * 32 + e / 16, i = i0 + 4 * (e % 16);
    if (c < C && i + 3 < N) {
      const PF4 v = *reinterpret_cast<const PF4*>(f1b + (int64_t)c * N + i);
      r[4 * j + 0] = v.v[0]; r[4 * j + 1] = v.v[1]; r[4 * j + 2] = v.v[2]; r[4 * j + 3] = v.v[3];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) r[4 * j + q] = (c < C && i + q < N) ? f1b[(int64_t)c * N + i + q] : 0.f;
    }
  }
};
struct F2SplitLoader {        // chunk e: channel k = e / 64, tile columns n = 4*(e % 64) .. +3 = 4 consecutive patch columns
  static constexpr int NCH = BuildSplitCfg::NCH_B, NREG = NCH * 4;
  const float* f2b; int N, C, H, W, px0, py0;
  __device__ __forceinline__ void fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int c = kt * 32 + e / 64, n = 4 * (e % 64);
    const int rem = n & 63, row = ((rem >> 5) << 2) + ((rem & 31) >> 3), col = ((n >> 6) << 3) + (rem & 7);
    const int y = py0 + row, x = px0 + col;
    if (c < C && y < H && x + 3 < W) {
      const PF4 v = *reinterpret_cast<const PF4*>(f2b + (int64_t)c * N + y * W + x);
      r[4 * j + 0] = v.v[0]; r[4 * j + 1] = v.v[1]; r[4 * j + 2] = v.v[2]; r[4 * j + 3] = v.v[3];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) r[4 * j + q] = (c < C && y < H && x + q < W) ? f2b[(int64_t)c * N + y * W + x + q] : 0.f;
    }
  }
};

constexpr int LDS_SPLIT_BYTES = BuildSplitCfg::LDS_BYTES > EPI_FLOATS * 4 ? BuildSplitCfg::LDS_BYTES : EPI_FLOATS * 4;

// am1 / am2: amax words of the two feature maps (split_arith.hpp; NULL: scale 1)
__global__ __launch_bounds__(256) void corr_build_split_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                               Levels lv, int nlev, int C, int H, int W, float scale,
                                                               const unsigned* am1, const unsigned* am2) {
  const float s1 = fs_scale_of_amax(fs_amax_load(am1)), s2 = fs_scale_of_amax(fs_amax_load(am2));
  scale *= fs_inv_scale(s1) * fs_inv_scale(s2);
  __shared__ __attribute__((aligned(16))) char lds[LDS_SPLIT_BYTES];
  const int N = H * W;
  const int npx = ceil_div_dev(W, 32);
  const int px0 = (blockIdx.x % npx) * 32, py0 = (blockIdx.x / npx) * 8;
  const int i0 = blockIdx.y * 64;
  const int b = blockIdx.z;
  F1SplitLoader la{f1 + (int64_t)b * C * N, N, i0, C};
  F2SplitLoader lb{f2 + (int64_t)b * C * N, N, C, H, W, px0, py0};
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  split_mainloop_tn<BuildSplitCfg>(lds, ceil_div_dev(C, 32), la, lb, acc, nullptr, s1, s2);
  __syncthreads();
  build_epilogue(acc, reinterpret_cast<float*>(lds), lv, nlev, H, W, N, b, i0, px0, py0, scale);
}

// ---- the same two kernels writing the tiled-row layout (corr_layout.hpp) ----
__global__ __launch_bounds__(256) void corr_build_tiled_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                               float* __restrict__ vol, VolLayout L, int C, float scale) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  const int H = L.H, W = L.W, N = H * W;
  const int npx = ceil_div_dev(W, 32);
  const int px0 = (blockIdx.x % npx) * 32, py0 = (blockIdx.x / npx) * 8;
  const int i0 = blockIdx.y * 64;
  const int b = blockIdx.z;
  F1Loader la{f1 + (int64_t)b * C * N, N, i0, C};
  const int prow = threadIdx.x >> 5, pcol = threadIdx.x & 31;
  const bool pok = (py0 + prow < H) && (px0 + pcol < W);
  F2Loader lb{f2 + (int64_t)b * C * N, N, C, (py0 + prow) * W + px0 + pcol, pok};
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  gemm_mainloop<BuildCfg>(lds, ceil_div_dev(C, 16), la, lb, acc);
  build_epilogue_tiled(acc, lds, vol, L, N, b, i0, px0, py0, scale);
}

__global__ __launch_bounds__(256) void corr_build_tiled_split_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                                     float* __restrict__ vol, VolLayout L, int C, float scale,
                                                                     const unsigned* am1, const unsigned* am2) {
  const float s1 = fs_scale_of_amax(fs_amax_load(am1)), s2 = fs_scale_of_amax(fs_amax_load(am2));
  scale *= fs_inv_scale(s1) * fs_inv_scale(s2);
  __shared__ __attribute__((aligned(16))) char lds[LDS_SPLIT_BYTES];
  const int H = L.H, W = L.W, N = H * W;
  const int npx = ceil_div_dev(W, 32);
  const int px0 = (blockIdx.x % npx) * 32, py0 = (blockIdx.x / npx) * 8;
  const int i0 = blockIdx.y * 64;
  const int b = blockIdx.z;
  F1SplitLoader la{f1 + (int64_t)b * C * N, N, i0, C};
  F2SplitLoader lb{f2 + (int64_t)b * C * N, N, C, H, W, px0, py0};
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  split_mainloop_tn<BuildSplitCfg>(lds, ceil_div_dev(C, 32), la, lb, acc, nullptr, s1, s2);
  __syncthreads();
  build_epilogue_tiled(acc, reinterpret_cast<float*>(lds), vol, L, N, b, i0, px0, py0, scale);
}

// ---- build on the record core (gemm_rec.hpp): feature maps arrive PRE-SPLIT, pixel-major ([B][N][C / 32 records]) ----
// 256 queries x one 8x16 patch of targets per workgroup, K = C.  Both operands are staged by LDS-DMA (the target rows of
// a patch are gathered through the per-lane source offsets; pixels outside the image read as zeros), eight waves, the
// three-slot ring of rec_mainloop.  Epilogue as above, in two halves of 128 queries: parked in LDS, level 0 written as
// whole 64-byte tiles, levels 1-3 pooled out of LDS.
// store of the finished volume with a cache policy chosen at run time (fsraft_set_build_kernel bits 16..17; A/B in
// round 3, docs/history): 0 plain, 1 `sc1` (write through and DROP the line from the XCD's L2: the 1.08 GB of output then
// does not evict the 58 MB of feature records every workgroup re-reads), 2 `nt`.
__device__ __forceinline__ void vstore4(float* p, f32x4 v, int policy) {
  if (policy == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (policy == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else gstore4(p, v);
}
using BR = RecCfg<256, 128, 4, 2>;
constexpr int BR_PH = 8, BR_PW = 16;                       // the target patch
constexpr int BR_EPI_FLOATS = 128 * 128 + 128 * 32 + 128 * 8;

// stagger: every CU runs one workgroup at a time and all of them take the same time, so the whole chip alternates between a
// phase in which nothing is stored and a phase in which 45 MB are; workgroups of the first dispatch round with an odd index
// sleep `stagger` x 64 x 127 cycles first, which puts their CUs half a period out of step with the others for the whole launch.
__device__ __forceinline__ void build_stagger(int stagger) {
  const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (stagger > 0 && id < 256u && (id & 1u))
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
}

__global__ __launch_bounds__(512) void corr_build_rec_kernel(const char* __restrict__ f1r, const char* __restrict__ f2r,
                                                             float* __restrict__ vol, VolLayout L, int C, float scale, int stagger, int policy,
                                                             const unsigned* am1, const unsigned* am2) {   // the words the records were split with
  scale *= fs_inv_scale(fs_scale_of_amax(fs_amax_load(am1))) * fs_inv_scale(fs_scale_of_amax(fs_amax_load(am2)));
  __shared__ __attribute__((aligned(1024))) char lds[BR::LDS_BYTES > BR_EPI_FLOATS * 4 ? BR::LDS_BYTES : BR_EPI_FLOATS * 4];
  build_stagger(stagger);
  const int H = L.H, W = L.W, N = H * W;
  const int npx = ceil_div_dev(W, BR_PW);
  const int px0 = (blockIdx.x % npx) * BR_PW, py0 = (blockIdx.x / npx) * BR_PH;
  const int i0 = blockIdx.y * BR::BM;
  const int b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned pitch = (unsigned)C * 4u;
  RecOperands<BR> o;
  const char* A = f1r + ((int64_t)b * N + i0) * pitch;
  const int ar = min(BR::BM, N - i0);
  o.da = rec_desc(A, (unsigned)ar * pitch);
  o.db = rec_desc(f2r + (int64_t)b * N * pitch, (unsigned)min((int64_t)N * pitch, (int64_t)0x7fffffff));
  o.b_step = 128u;
  RecPlainA<BR> pa;
#pragma unroll
  for (int j = 0; j < BR::NPA; ++j) pa.va[j] = rec_piece_voff(wave + BR::NWAVE * j, lane, ar, pitch);
  pa.kt0 = 0; pa.step = 128u;
#pragma unroll
  for (int j = 0; j < BR::NPB; ++j) {                      // tile row n = patch position (n / 16, n % 16)
    const int n = (wave + BR::NWAVE * j) * 8 + (lane >> 3);
    const int y = py0 + n / BR_PW, x = px0 + n % BR_PW;
    const int ls = (lane & 7) ^ ((n >> 1) & 7);
    o.vb[j] = (y < H && x < W) ? (unsigned)(y * W + x) * pitch + (unsigned)ls * 16u : 0x80000000u;
  }
  f32x16 acc[BR::TM][BR::TN];
#pragma unroll
  for (int a = 0; a < BR::TM; ++a)
#pragma unroll
    for (int c = 0; c < BR::TN; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  rec_mainloop<BR>(lds, o, pa, 0, C / 32, acc);

  float* S = reinterpret_cast<float*>(lds);   // [128 queries][128 = 8 x 16 patch]
  float* P1 = S + 128 * 128;                  // [128][4 x 8]
  float* P2 = P1 + 128 * 32;                  // [128][2 x 4]
  const int wm = wave / BR::WN, wn = wave % BR::WN, l31 = lane & 31, lh = lane >> 5;
  const int nlev = L.nlev;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if ((wm >> 1) == half) {
#pragma unroll
      for (int mt = 0; mt < BR::TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < BR::TN; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = (wm & 1) * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            S[i * 128 + wn * 64 + nt * 32 + l31] = acc[mt][nt][r] * scale;
          }
    }
    __syncthreads();
    const int qbase = i0 + half * 128;
    float* rows = vol + ((int64_t)b * N + qbase) * L.P;
    // level 0: chunk = (query i, tile row ty, tile t, row r of the tile) = 16 bytes; 16 lanes cover one 256-byte run
#pragma unroll 4
    for (int jj = 0; jj < 8; ++jj) {
      const int id = threadIdx.x + 512 * jj;
      const int r = id & 3, t = (id >> 2) & 3, ty = (id >> 4) & 1, i = id >> 5;
      const int gty = (py0 >> 2) + ty, gtx = (px0 >> 2) + t;
      if (gty < L.th[0] && gtx < L.tw[0] && qbase + i < N) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(S + i * 128 + (4 * ty + r) * BR_PW + 4 * t);
        vstore4(rows + (int64_t)i * L.P + L.off[0] + (gty * L.tw[0] + gtx) * 16 + r * 4, v, policy);
      }
    }
    if (nlev > 1) {
      const int h1 = L.h[1], w1 = L.w[1];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int e = threadIdx.x + 512 * jj;
        const int i = e >> 5, c = e & 31, y = c >> 3, x = c & 7;
        const float* s = S + i * 128 + (2 * y) * BR_PW + 2 * x;
        const float v = (((s[0] + s[1]) + s[BR_PW]) + s[BR_PW + 1]) * 0.25f;
        // cells beyond the floor-halved size do not exist in the reference pyramid: pad cells of the tile hold 0
        P1[i * 32 + c] = ((py0 >> 1) + y < h1 && (px0 >> 1) + x < w1) ? v : 0.f;
      }
    }
    __syncthreads();
    if (nlev > 1) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int id = threadIdx.x + 512 * jj;
        const int r = id & 3, t = (id >> 2) & 1, i = id >> 3;
        const int gty = py0 >> 3, gtx = (px0 >> 3) + t;
        if (gty < L.th[1] && gtx < L.tw[1] && qbase + i < N) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(P1 + i * 32 + r * 8 + 4 * t);
          vstore4(rows + (int64_t)i * L.P + L.off[1] + (gty * L.tw[1] + gtx) * 16 + r * 4, v, policy);
        }
      }
    }
    if (nlev > 2) {
      const int h2 = L.h[2], w2 = L.w[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int e = threadIdx.x + 512 * jj;
        const int i = e >> 3, c = e & 7, y = c >> 2, x = c & 3;
        const float* s = P1 + i * 32 + (2 * y) * 8 + 2 * x;
        const float v = (((s[0] + s[1]) + s[8]) + s[9]) * 0.25f;
        P2[i * 8 + c] = ((py0 >> 2) + y < h2 && (px0 >> 2) + x < w2) ? v : 0.f;
      }
    }
    __syncthreads();
    if (nlev > 2 && threadIdx.x < 256) {
      const int id = threadIdx.x;
      const int y = id & 1, i = id >> 1;
      const int gy = (py0 >> 2) + y, gtx = px0 >> 4;
      if ((gy >> 2) < L.th[2] && gtx < L.tw[2] && qbase + i < N) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(P2 + i * 8 + y * 4);
        vstore4(rows + (int64_t)i * L.P + L.off[2] + ((gy >> 2) * L.tw[2] + gtx) * 16 + (gy & 3) * 4, v, policy);
      }
    }
    if (nlev > 3 && threadIdx.x >= 256) {
      const int h3 = L.h[3], w3 = L.w[3];
      const int id = threadIdx.x - 256;
      const int i = id >> 1, x = id & 1;
      const float* s = P2 + i * 8 + 2 * x;
      const float v = (((s[0] + s[1]) + s[4]) + s[5]) * 0.25f;
      const int gy = (py0 >> 3), gx = (px0 >> 3) + x;
      if ((gy >> 2) < L.th[3] && (gx >> 2) < L.tw[3] && qbase + i < N)
        gstore1(rows + (int64_t)i * L.P + vol_cell(L, 3, gy, gx), (gy < h3 && gx < w3) ? v : 0.f);
    }
    __syncthreads();
  }
}


int g_build_policy = -1;  // cache policy of the volume stores (vstore4): -1 auto, 0 plain, 1 sc1, 2 nt
int g_build_stagger = 0;  // see build_stagger (fsraft_set_build_kernel: bits 8.. of the argument)
int g_build_split = 1;    // 0: exact fp32 MFMA build, 1: split-bf16 (fsraft_set_build_split)

// Backward of the pooling chain, folded into level 0 in place:
//   g0[y][x] += 1/4 * ( g1[y/2][x/2] + 1/4 * ( g2[y/4][x/4] + 1/4 * g3[y/8][x/8] ) )
// where a level-l cell only feeds back if it exists (index < size of that level);
// existence at level l+1 implies existence at level l, so the chain is cut from the top.
__global__ __launch_bounds__(256) void corr_unpool_bwd_kernel(Levels lv, int nlev, int64_t nq) {
  const int H = lv.h[0], W = lv.w[0];
  const int64_t total = nq * H * W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % W);
    const int64_t t = e / W;
    const int y = (int)(t % H);
    const int64_t q = t / H;
    float up = 0.f;   // dL/d(cell of level l) including what flows down from coarser levels
    for (int l = nlev - 1; l >= 1; --l) {
      const int yl = y >> l, xl = x >> l;
      const bool ok = yl < lv.h[l] && xl < lv.w[l];   // floor pooling dropped this cell otherwise
      up = ok ? (lv.p[l][(q * lv.h[l] + yl) * lv.w[l] + xl] + 0.25f * up) : 0.f;
    }
    lv.p[0][e] += 0.25f * up;
  }
}

// W % 4 == 0 fast path: one thread per 4 consecutive x (16-byte load/store of level 0); the 4 pixels share
// their level >= 2 ancestors and touch two level-1 cells.
__global__ __launch_bounds__(256) void corr_unpool_bwd_vec_kernel(Levels lv, int nlev, int64_t nrows) {
  const int H = lv.h[0], W = lv.w[0], W4 = W >> 2;
  const int64_t total = nrows * W4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % W4) * 4;
    const int64_t row = e / W4;            // = q*H + y
    const int y = (int)(row % H);
    const int64_t q = row / H;
    float up = 0.f;                        // common ancestors: levels >= 2 (cells of >= 4 pixels)
    for (int l = nlev - 1; l >= 2; --l) {
      const int yl = y >> l, xl = x >> l;
      const bool ok = yl < lv.h[l] && xl < lv.w[l];
      up = ok ? (lv.p[l][(q * lv.h[l] + yl) * lv.w[l] + xl] + 0.25f * up) : 0.f;
    }
    float g1[2] = {0.f, 0.f};
    if (nlev > 1) {
      const int y1 = y >> 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int x1 = (x >> 1) + i;
        const bool ok = y1 < lv.h[1] && x1 < lv.w[1];
        g1[i] = ok ? (lv.p[1][(q * lv.h[1] + y1) * lv.w[1] + x1] + 0.25f * up) : 0.f;
      }
    }
    f32x4* p = reinterpret_cast<f32x4*>(lv.p[0] + row * W + x);
    f32x4 v = *p;
    v[0] += 0.25f * g1[0]; v[1] += 0.25f * g1[0]; v[2] += 0.25f * g1[1]; v[3] += 0.25f * g1[1];
    *p = v;
  }
}

// dst[row][y][x] = 2x2 average of src[row][2y..][2x..] (floor sizes), the same expression the fused build uses
__global__ __launch_bounds__(256) void corr_pool_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows,
                                                        int h, int w, int h2, int w2) {
  const int64_t total = rows * h2 * w2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % w2), y = (int)((e / w2) % h2);
    const int64_t r = e / ((int64_t)w2 * h2);
    const float* s = src + (r * h + 2 * y) * w + 2 * x;
    dst[e] = (((s[0] + s[1]) + s[w]) + s[w + 1]) * 0.25f;
  }
}

// TensorFlow avg_pool2d(x, k, k, 'SAME') of the level-0 volume: out = ceil(size / k), total padding out * k - size split
// floor / ceil between the leading and the trailing side, the average taken over the in-range elements only.
__global__ __launch_bounds__(256) void corr_pool_same_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows,
                                                             int h, int w, int k) {
  const int h2 = (h + k - 1) / k, w2 = (w + k - 1) / k;
  const int py = (h2 * k - h) / 2, px = (w2 * k - w) / 2;
  const int64_t total = rows * h2 * w2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % w2), y = (int)((e / w2) % h2);
    const int64_t r = e / ((int64_t)w2 * h2);
    const int y0 = max(y * k - py, 0), y1 = min(y * k - py + k, h), x0 = max(x * k - px, 0), x1 = min(x * k - px + k, w);
    float acc = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) acc += src[(r * h + yy) * w + xx];
    dst[e] = acc / (float)((y1 - y0) * (x1 - x0));
  }
}

}  // namespace

// Pyramid of a level-0 volume that already exists: levels[0] is [rows][H2][W2] (rows = B*H1*W1), levels[l >= 1] receive the
// 2x2 averages of levels[l-1].  raft/allfield.py:94-106 build_pyramid -- applied to the transposed volume for the backward
// flow at raft/semi.py:250-251, 257-258.
extern "C" int fsraft_corr_pool_pyramid(float* const* levels, int num_levels, int64_t rows, int H2, int W2, hipStream_t stream) {
  if (!levels || num_levels < 1 || num_levels > 8 || rows < 1 || H2 < 1 || W2 < 1) return FS_ERR_ARG;
  int h = H2, w = W2;
  for (int l = 1; l < num_levels; ++l) {
    const int h2 = h / 2, w2 = w / 2;
    if (!levels[l - 1] || !levels[l] || h2 < 1 || w2 < 1) return FS_ERR_ARG;
    int64_t blocks = (rows * h2 * w2 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(corr_pool_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, levels[l - 1], levels[l], rows, h, w, h2, w2);
    h = h2; w = w2;
  }
  return fs_launch_status();
}

extern "C" int fsraft_corr_build(const float* fmap1, const float* fmap2, float* const* levels, int num_levels,
                                 int B, int C, int H, int W, const unsigned* amax1, const unsigned* amax2, hipStream_t stream) {
  if (!fmap1 || !fmap2 || !levels || num_levels < 1 || num_levels > 4 || B < 1 || C < 2 || (C & 1) || H < 1 || W < 1) return FS_ERR_ARG;
  Levels lv;
  int h = H, w = W;
  for (int l = 0; l < 4; ++l) {
    lv.p[l] = l < num_levels ? levels[l] : nullptr;
    lv.h[l] = h; lv.w[l] = w;
    if (l < num_levels && (!levels[l] || h < 1 || w < 1)) return FS_ERR_ARG;
    h /= 2; w /= 2;
  }
  const int N = H * W;
  dim3 grid(ceil_div(W, 32) * ceil_div(H, 8), ceil_div(N, 64), B);
  if (g_build_split)
    hipLaunchKernelGGL(corr_build_split_kernel, grid, dim3(256), 0, stream, fmap1, fmap2, lv, num_levels, C, H, W,
                       1.0f / sqrtf((float)C), amax1, amax2);
  else
    hipLaunchKernelGGL(corr_build_kernel, grid, dim3(256), 0, stream, fmap1, fmap2, lv, num_levels, C, H, W,
                       1.0f / sqrtf((float)C));
  return fs_launch_status();
}

// Volume + pyramid of one batch in the tiled-row layout: vol [B*H*W][P] (fsraft_vol_layout gives P and the level offsets).
extern "C" int fsraft_corr_build_tiled(const float* fmap1, const float* fmap2, float* vol, int num_levels, int B, int C, int H,
                                       int W, const unsigned* amax1, const unsigned* amax2, hipStream_t stream) {
  VolLayout L;
  if (!fmap1 || !fmap2 || !vol || B < 1 || C < 2 || (C & 1) || !vol_layout_make(H, W, num_levels, L)) return FS_ERR_ARG;
  if ((uintptr_t)vol % 16) return FS_ERR_ARG;
  const int N = H * W;
  dim3 grid(ceil_div(W, 32) * ceil_div(H, 8), ceil_div(N, 64), B);
  if (g_build_split)
    hipLaunchKernelGGL(corr_build_tiled_split_kernel, grid, dim3(256), 0, stream, fmap1, fmap2, vol, L, C, 1.0f / sqrtf((float)C), amax1, amax2);
  else
    hipLaunchKernelGGL(corr_build_tiled_kernel, grid, dim3(256), 0, stream, fmap1, fmap2, vol, L, C, 1.0f / sqrtf((float)C));
  return fs_launch_status();
}

// out[0..19] = nlev, H, W, P, h[4], w[4], th[4], tw[4] ... see fsraft.h
extern "C" int fsraft_vol_layout(int H, int W, int num_levels, int* out) {
  VolLayout L;
  if (!out || !vol_layout_make(H, W, num_levels, L)) return FS_ERR_ARG;
  out[0] = L.nlev; out[1] = L.H; out[2] = L.W; out[3] = L.P;
  for (int l = 0; l < 4; ++l) { out[4 + l] = L.h[l]; out[8 + l] = L.w[l]; out[12 + l] = L.th[l]; out[16 + l] = L.tw[l]; out[20 + l] = L.off[l]; }
  return FS_OK;
}

// The same build from PRE-SPLIT feature maps: f1r, f2r = [B][H*W][C / 32] records (fsraft_to_records of the channels-last
// maps, split with the words amax1 / amax2), C % 32 == 0.  Always split arithmetic (the exact-fp32 build is fsraft_corr_build_tiled with the split switched off).
extern "C" int fsraft_corr_build_rec(const void* f1r, const void* f2r, float* vol, int num_levels, int B, int C, int H, int W,
                                     const unsigned* amax1, const unsigned* amax2, hipStream_t stream) {
  VolLayout L;
  if (!f1r || !f2r || !vol || B < 1 || C < 32 || (C % 32) || !vol_layout_make(H, W, num_levels, L)) return FS_ERR_ARG;
  if (((uintptr_t)vol % 16) || ((uintptr_t)f1r % 16) || ((uintptr_t)f2r % 16) || (int64_t)H * W * C * 4 >= 0x7fffffff) return FS_ERR_ARG;
  const int N = H * W;
  dim3 grid(ceil_div(W, BR_PW) * ceil_div(H, BR_PH), ceil_div(N, BR::BM), B);
  // `nt` stores once the volume is larger than the Infinity Cache: measured 497 -> 430 us at 4 x 55x128 (1.08 GB), 196 -> 165 us
  // at 8 x 46x62; at one pair (270 MB) plain stores are as fast or faster (round 3, docs/history)
  const int policy = g_build_policy >= 0 ? g_build_policy : ((int64_t)B * N * L.P * 4 > ((int64_t)300 << 20) ? 2 : 0);
  hipLaunchKernelGGL(corr_build_rec_kernel, grid, dim3(512), 0, stream, (const char*)f1r, (const char*)f2r, vol, L, C,
                     1.0f / sqrtf((float)C), g_build_stagger, policy, amax1, amax2);
  return fs_launch_status();
}


extern "C" int fsraft_set_build_split(int on) {
  g_build_split = on ? 1 : 0;
  return FS_OK;
}
extern "C" int fsraft_set_build_kernel(int which) {      // bits 8..15: start-up stagger (build_stagger); bits 16..18: store policy
  g_build_stagger = (which >> 8) & 0xff;
  g_build_policy = ((which >> 16) & 7) - 1;           // bits 16..18: 0 auto, 1 plain, 2 sc1, 3 nt
  return FS_OK;
}

// In place: levels[0] += unpooled(levels[1..]).  levels[l] hold dL/dV_l on entry.
extern "C" int fsraft_corr_unpool_bwd(float* const* dlevels, int num_levels, int B, int H, int W, hipStream_t stream) {
  if (!dlevels || num_levels < 1 || num_levels > 4) return FS_ERR_ARG;
  if (num_levels == 1) return FS_OK;
  Levels lv;
  int h = H, w = W;
  for (int l = 0; l < 4; ++l) {
    lv.p[l] = l < num_levels ? dlevels[l] : nullptr;
    lv.h[l] = h; lv.w[l] = w;
    h /= 2; w /= 2;
  }
  const int64_t nq = (int64_t)B * H * W;
  const int64_t total = nq * H * W;
  if (W % 4 == 0 && ((uintptr_t)dlevels[0] % 16) == 0) {
    const int64_t work = total / 4;
    int blocks = (int)((work + 255) / 256 < 16384 ? (work + 255) / 256 : 16384);
    hipLaunchKernelGGL(corr_unpool_bwd_vec_kernel, dim3(blocks), dim3(256), 0, stream, lv, num_levels, nq * H);
    return fs_launch_status();
  }
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(corr_unpool_bwd_kernel, dim3(blocks), dim3(256), 0, stream, lv, num_levels, nq);
  return fs_launch_status();
}

// TensorFlow semantics of the same pyramid (raft/allfield.py:99-104: avg_pool2d(level 0, scale, scale, 'SAME') for scale =
// 2, 4, ...): levels[l] is [rows][ceil(H2 / 2^l)][ceil(W2 / 2^l)], every level pooled from level 0 with partial edge windows
// averaged over their in-range elements.  Equals fsraft_corr_pool_pyramid (to rounding) when H2 and W2 divide by 2^(L-1).
extern "C" int fsraft_corr_pool_pyramid_same(float* const* levels, int num_levels, int64_t rows, int H2, int W2, hipStream_t stream) {
  if (!levels || num_levels < 1 || num_levels > 8 || rows < 1 || H2 < 1 || W2 < 1 || !levels[0]) return FS_ERR_ARG;
  for (int l = 1; l < num_levels; ++l) {
    if (!levels[l]) return FS_ERR_ARG;
    const int k = 1 << l, h2 = (H2 + k - 1) / k, w2 = (W2 + k - 1) / k;
    int64_t blocks = (rows * h2 * w2 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(corr_pool_same_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, levels[0], levels[l], rows, H2, W2, k);
  }
  return fs_launch_status();
}
