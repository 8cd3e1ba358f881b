// Memory-efficient correlation lookup: windowed dot products straight from the feature
// maps, never materialising the N x N volume (rows a4, a5 of SURVEY.md section 8).
// Drop-in for alt_cuda_corr.forward / .backward
// (pytorch/alt_cuda_corr/correlation.cpp:23-54, correlation_kernel.cu:18-119, 122-256):
// same tensor layouts, same channel order (iy + (2r+1)*ix, i.e. x offset slow), same
// zero-outside-the-map rule, unscaled output.  This is a from-scratch wave64 design, not
// a translation of the 32-thread CUDA blocks:
//
//   one wavefront = one query pixel.  Lane l holds channels {l, l+64, l+128, l+192} of the
//   query's fmap1 vector in registers.  For each of the (2r+2)^2 integer window positions
//   the wave reads the fmap2 pixel's channel vector as 256-byte coalesced rows and each
//   lane accumulates a partial dot.  The 64 partials-per-position are combined with a
//   halving butterfly (63 DPP/shuffle steps for 64 positions instead of 6 per position),
//   leaving position p's dot on lane p.  The bilinear blend of the 4 neighbouring dots
//   then produces the (2r+1)^2 outputs.
//
// Backward mirrors it: per window position g = blend^T(corr_grad); fmap1_grad accumulates
// g * fmap2 rows in registers; fmap2_grad receives g * fmap1 via fp32 atomics issued as
// 256-byte contiguous wave instructions (the shape that runs at the full atomic rate).
#include "common.hpp"
#include "gemm_rec.hpp"

namespace {

constexpr int MAXK = 4;   // up to 256 channels (64 lanes x 4)

// v[i] (i < 64) are per-lane partial sums for position i; returns on lane i the total over lanes.
// Step HALF: lanes with bit HALF set keep positions [HALF, 2*HALF) of what they hold, the
// others keep [0, HALF); the discarded half goes to the partner lane ^ HALF.  After the six
// steps lane l holds position l summed over all 64 lanes (63 shuffles in total).
template <int HALF>
__device__ __forceinline__ void bfly_step(float (&v)[64], bool upper) {
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const float keep = upper ? v[i + HALF] : v[i];
    const float send = upper ? v[i] : v[i + HALF];
    v[i] = keep + __shfl_xor(send, HALF, 64);
  }
}
__device__ __forceinline__ float butterfly64(float (&v)[64]) {
  const int lane = threadIdx.x & 63;
  bfly_step<32>(v, (lane & 32) != 0);
  bfly_step<16>(v, (lane & 16) != 0);
  bfly_step<8>(v, (lane & 8) != 0);
  bfly_step<4>(v, (lane & 4) != 0);
  bfly_step<2>(v, (lane & 2) != 0);
  bfly_step<1>(v, (lane & 1) != 0);
  return v[0];
}

template <int R>
__global__ __launch_bounds__(256) void altcorr_fwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                          const float* __restrict__ coords, float* __restrict__ corr,
                                                          int B, int N, int H1, int W1, int H2, int W2, int C) {
  constexpr int RD = 2 * R + 1, WIN = RD + 1, NPOS = WIN * WIN;
  __shared__ float dots[4][NPOS + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nq = (int64_t)B * H1 * W1;
  const bool active = q < nq;
  const int64_t qq = active ? q : nq - 1;
  const int b = (int)(qq / (H1 * W1)), pix = (int)(qq % (H1 * W1));

  float a[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) a[k] = (lane + 64 * k < C) ? f1[qq * C + lane + 64 * k] : 0.f;

  const float* f2b = f2 + (int64_t)b * H2 * W2 * C;
  // N coordinate sets per query pixel (coords [B,N,H1,W1,2] -> corr [B,N,RD*RD,H1,W1], correlation_kernel.cu:34,59):
  // the pixel's feature vector stays in registers across them
#pragma unroll 1
  for (int n = 0; n < N; ++n) {
  const int64_t cq = ((int64_t)b * N + n) * H1 * W1 + pix;
  float cx = coords[cq * 2], cy = coords[cq * 2 + 1];
  cx = (cx > -30000.f && cx < 30000.f) ? cx : -30000.f;
  cy = (cy > -30000.f && cy < 30000.f) ? cy : -30000.f;
  const float flx = floorf(cx), fly = floorf(cy);
  const int x0 = (int)flx, y0 = (int)fly;
  const float dx = cx - flx, dy = cy - fly;
#pragma unroll 1
  for (int base = 0; base < NPOS; base += 64) {
    float part[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const int p = base + i;
      const int iy = p / WIN, ix = p % WIN;
      const int h2 = y0 - R + iy, w2 = x0 - R + ix;
      float s = 0.f;
      if (p < NPOS && h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
        const float* row = f2b + ((int64_t)h2 * W2 + w2) * C + lane;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
          if (lane + 64 * k < C) s += a[k] * row[64 * k];
      }
      part[i] = s;
    }
    const float tot = butterfly64(part);
    // after the butterfly lane l holds the position whose index is the bit-reversal-free
    // mapping below: bit (5-step) of the lane selected the upper half at each step, so
    // position index == lane.
    if (base + lane < NPOS) dots[wave][base + lane] = tot;
  }
  __syncthreads();
  float* out = corr + ((int64_t)b * N + n) * RD * RD * H1 * W1 + pix;
  for (int o = lane; active && o < RD * RD; o += 64) {
    const int iyo = o % RD, ixo = o / RD;          // channel = iy + RD*ix
    const float* d = dots[wave] + iyo * WIN + ixo;
    // dot at (iy,ix) contributes to out(iy-1,ix-1)*dy*dx, out(iy-1,ix)*dy*(1-dx), out(iy,ix-1)*(1-dy)*dx, out(iy,ix)*(1-dy)*(1-dx)
    const float v = (1.f - dy) * (1.f - dx) * d[0] + (1.f - dy) * dx * d[1] + dy * (1.f - dx) * d[WIN] + dy * dx * d[WIN + 1];
    out[(int64_t)o * H1 * W1] = v;
  }
  __syncthreads();
  }
}

template <int R>
__global__ __launch_bounds__(256) void altcorr_bwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                          const float* __restrict__ coords, const float* __restrict__ cg,
                                                          float* __restrict__ g1, float* __restrict__ g2, int B, int N,
                                                          int H1, int W1, int H2, int W2, int C) {
  constexpr int RD = 2 * R + 1, WIN = RD + 1, NPOS = WIN * WIN;
  __shared__ float gout[4][RD * RD + 3];
  __shared__ float gpos[4][NPOS + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nq = (int64_t)B * H1 * W1;
  const bool active = q < nq;
  const int64_t qq = active ? q : nq - 1;
  const int b = (int)(qq / (H1 * W1)), pix = (int)(qq % (H1 * W1));
  float a[MAXK], acc[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) { a[k] = (lane + 64 * k < C) ? f1[qq * C + lane + 64 * k] : 0.f; acc[k] = 0.f; }
  const float* f2b = f2 + (int64_t)b * H2 * W2 * C;
  float* g2b = g2 + (int64_t)b * H2 * W2 * C;
#pragma unroll 1
  for (int n = 0; n < N; ++n) {                      // the N coordinate sets of a pixel add into the same fmap1_grad row
    const int64_t cq = ((int64_t)b * N + n) * H1 * W1 + pix;
    float cx = coords[cq * 2], cy = coords[cq * 2 + 1];
    cx = (cx > -30000.f && cx < 30000.f) ? cx : -30000.f;
    cy = (cy > -30000.f && cy < 30000.f) ? cy : -30000.f;
    const float flx = floorf(cx), fly = floorf(cy);
    const int x0 = (int)flx, y0 = (int)fly;
    const float dx = cx - flx, dy = cy - fly;

    const float* gin = cg + ((int64_t)b * N + n) * RD * RD * H1 * W1 + pix;
    for (int o = lane; o < RD * RD; o += 64) gout[wave][o] = active ? gin[(int64_t)o * H1 * W1] : 0.f;
    __syncthreads();
    for (int p = lane; p < NPOS; p += 64) {
      const int iy = p / WIN, ix = p % WIN;
      float g = 0.f;
      // position (iy,ix) is tap (a,c) of output (iy-a, ix-c); channel = iyo + RD*ixo
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int iyo = iy - aa, ixo = ix - cc;
          if (iyo >= 0 && iyo < RD && ixo >= 0 && ixo < RD)
            g += gout[wave][iyo + RD * ixo] * (aa ? dy : 1.f - dy) * (cc ? dx : 1.f - dx);
        }
      gpos[wave][p] = g;
    }
    __syncthreads();
    for (int p = 0; active && p < NPOS; ++p) {
      const int h2 = y0 - R + p / WIN, w2 = x0 - R + p % WIN;
      if (h2 < 0 || h2 >= H2 || w2 < 0 || w2 >= W2) continue;
      const float g = gpos[wave][p];
      const int64_t off = ((int64_t)h2 * W2 + w2) * C + lane;
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (lane + 64 * k < C) {
          acc[k] += g * f2b[off + 64 * k];
          atomicAdd(g2b + off + 64 * k, g * a[k]);
        }
    }
    __syncthreads();
  }
  if (!active) return;
#pragma unroll
  for (int k = 0; k < MAXK; ++k)
    if (lane + 64 * k < C) g1[qq * C + lane + 64 * k] = acc[k];
}

// ---------------------------------------------------------------------------------------------------------------
// All pyramid levels in ONE launch, written straight into the channels-last [B,H,W,L*(2r+1)^2] tensor the update block
// consumes, scaled by 1/sqrt(C): what AlternateCorrBlock.__call__ (pytorch/core/corr.py:74-91) assembles from four
// extension calls, a stack, a reshape and a division per iteration.  Same wave-per-query dot-product scheme as above;
// level l reads the l-times average-pooled target map f2[l] ([B,h_l,w_l,C] channels-last) at coords / 2^l.
struct AltLevels {
  const float* f2[4];
  int h[4], w[4];
};
struct AltCoords {
  const float* p;
  int64_t bs, cs, ps;
  int grid_w;          // > 0: p holds the flow, the query position is pixel grid + flow
};

template <int R>
__global__ __launch_bounds__(256) void altcorr_fused_fwd_kernel(const float* __restrict__ f1, AltLevels lv, AltCoords co,
                                                                float* __restrict__ out, int nlev, int B, int HW, int C, float scale,
                                                                const int* __restrict__ regime) {
  constexpr int RD = 2 * R + 1, WIN = RD + 1, NPOS = WIN * WIN;
  __shared__ float dots[4][NPOS + 4];
  if (regime && regime[0] == 0) return;                 // (dispatched launch: the matrix-pipe kernel takes the smooth regime)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nq = (int64_t)B * HW;
  if (q >= nq) return;                                  // (wave-uniform; no workgroup barrier below)
  const int b = (int)(q / HW), pix = (int)(q % HW);
  float cx0 = co.p[b * co.bs + pix * co.ps], cy0 = co.p[b * co.bs + co.cs + pix * co.ps];
  if (co.grid_w > 0) { cx0 += (float)(pix % co.grid_w); cy0 += (float)(pix / co.grid_w); }
  float a[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) a[k] = (lane + 64 * k < C) ? f1[q * C + lane + 64 * k] : 0.f;
  const int CH = nlev * RD * RD;
  for (int l = 0; l < nlev; ++l) {
    const float s = 1.0f / (float)(1 << l);
    float cx = cx0 * s, cy = cy0 * s;
    cx = (cx > -30000.f && cx < 30000.f) ? cx : -30000.f;
    cy = (cy > -30000.f && cy < 30000.f) ? cy : -30000.f;
    const float flx = floorf(cx), fly = floorf(cy);
    const int x0 = (int)flx, y0 = (int)fly;
    const float dx = cx - flx, dy = cy - fly;
    const int H2 = lv.h[l], W2 = lv.w[l];
    const float* f2b = lv.f2[l] + (int64_t)b * H2 * W2 * C;
#pragma unroll 1
    for (int base = 0; base < NPOS; base += 64) {
      float part[64];
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        const int p = base + i;
        const int iy = p / WIN, ix = p % WIN;
        const int h2 = y0 - R + iy, w2 = x0 - R + ix;
        float sum = 0.f;
        if (p < NPOS && h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
          const float* row = f2b + ((int64_t)h2 * W2 + w2) * C + lane;
#pragma unroll
          for (int k = 0; k < MAXK; ++k)
            if (lane + 64 * k < C) sum += a[k] * row[64 * k];
        }
        part[i] = sum;
      }
      const float tot = butterfly64(part);
      if (base + lane < NPOS) dots[wave][base + lane] = tot;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float* o = out + q * CH + l * RD * RD;
    for (int oc = lane; oc < RD * RD; oc += 64) {
      const int iyo = oc % RD, ixo = oc / RD;          // channel = iy + RD * ix (x offset slow, as the reference)
      const float* d = dots[wave] + iyo * WIN + ixo;
      o[oc] = scale * ((1.f - dy) * (1.f - dx) * d[0] + (1.f - dy) * dx * d[1] + dy * (1.f - dx) * d[WIN] + dy * dx * d[WIN + 1]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Tile variant of the fused lookup: the wave-per-query kernel above reads (2r+2)^2 target rows of C floats per query and
// level straight from L2 (400 KB per query, 2.9 GB per lookup at 47x156: L2-bandwidth bound), although the windows of
// neighbouring queries overlap almost completely when the flow is smooth.  Here a workgroup of SIXTEEN waves owns a 4x4
// tile of queries: per level it stages ONE 16x16 region of target rows (anchored at the first query's window) in LDS, 64
// channels at a time, and every wave (= query) takes its (2r+2)^2 dot products out of that region -- lane = window
// position, running over the channels -- falling back to global reads only for positions the region does not cover
// (flow discontinuities).  L2 traffic per tile and level: 256 rows instead of 16 x 100.
constexpr int AT_TQ = 4, AT_RS = 16, AT_CS = 64, AT_PITCH = AT_CS + 1;   // odd pitch: lane = position reads hit 32 distinct banks

template <int R>
__global__ __launch_bounds__(1024) void altcorr_tile_fwd_kernel(const float* __restrict__ f1, AltLevels lv, AltCoords co,
                                                                float* __restrict__ out, int nlev, int H, int W, int C, float scale,
                                                                const int* __restrict__ regime) {
  constexpr int RD = 2 * R + 1, WIN = RD + 1, NPOS = WIN * WIN, NRND = (NPOS + 63) / 64;
  if (regime && regime[0] == 0) return;                   // (dispatched launch: the matrix-pipe kernel takes the smooth regime)
  __shared__ float region[AT_RS * AT_RS * AT_PITCH];      // 65 KB
  __shared__ __attribute__((aligned(16))) float f1s[16][256];
  __shared__ float dots[16][NPOS + 4];
  __shared__ int org[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_x = (W + AT_TQ - 1) / AT_TQ;
  const int b = blockIdx.y, tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int qx = tx * AT_TQ + (wave & 3), qy = ty * AT_TQ + (wave >> 2);
  const bool active = qx < W && qy < H;                   // (wave-uniform)
  const int pix = active ? qy * W + qx : 0;
  const int64_t q = (int64_t)b * H * W + pix;
  float cx0 = 0.f, cy0 = 0.f;
  if (active) {
    cx0 = co.p[b * co.bs + pix * co.ps]; cy0 = co.p[b * co.bs + co.cs + pix * co.ps];
    if (co.grid_w > 0) { cx0 += (float)qx; cy0 += (float)qy; }
    for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<f32x4*>(&f1s[wave][c]) = *reinterpret_cast<const f32x4*>(f1 + q * C + c);
  }
  const int CH = nlev * RD * RD;
  for (int l = 0; l < nlev; ++l) {
    const float s = 1.0f / (float)(1 << l);
    float cx = cx0 * s, cy = cy0 * s;
    cx = (cx > -30000.f && cx < 30000.f) ? cx : -30000.f;
    cy = (cy > -30000.f && cy < 30000.f) ? cy : -30000.f;
    const float flx = floorf(cx), fly = floorf(cy);
    const int wx0 = (int)flx - R, wy0 = (int)fly - R;
    const float dx = cx - flx, dy = cy - fly;
    const int H2 = lv.h[l], W2 = lv.w[l];
    const float* f2b = lv.f2[l] + (int64_t)b * H2 * W2 * C;
    if (threadIdx.x == 0) { org[0] = wx0 - 1; org[1] = wy0 - 1; }     // (query 0 of a tile is always inside the image)
    __syncthreads();
    const int rx0 = org[0], ry0 = org[1];
    // this lane's window positions (one per round) and where they sit in the region / in the level
    int rpos[NRND], gpos[NRND];
#pragma unroll
    for (int k = 0; k < NRND; ++k) {
      const int p = lane + 64 * k, iy = p / WIN, ix = p % WIN;
      const int gx = wx0 + ix, gy = wy0 + iy, ux = gx - rx0, uy = gy - ry0;
      const bool inimg = active && p < NPOS && gx >= 0 && gx < W2 && gy >= 0 && gy < H2;
      gpos[k] = inimg ? gy * W2 + gx : -1;
      rpos[k] = (inimg && ux >= 0 && ux < AT_RS && uy >= 0 && uy < AT_RS) ? uy * AT_RS + ux : -1;
    }
    float acc[NRND];
#pragma unroll
    for (int k = 0; k < NRND; ++k) acc[k] = 0.f;
    for (int c0 = 0; c0 < C; c0 += AT_CS) {
      // stage the region's rows, channels [c0, c0 + 64): 16 lanes x 16 bytes per row, 64 rows per pass
      for (int rp = threadIdx.x >> 4; rp < AT_RS * AT_RS; rp += 64) {
        const int gx = rx0 + (rp & (AT_RS - 1)), gy = ry0 + rp / AT_RS, cc = (threadIdx.x & 15) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gx >= 0 && gx < W2 && gy >= 0 && gy < H2 && c0 + cc < C) v = *reinterpret_cast<const f32x4*>(f2b + ((int64_t)gy * W2 + gx) * C + c0 + cc);
        float* d = region + rp * AT_PITCH + cc;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NRND; ++k) {
        if (gpos[k] < 0) continue;
        float sum = 0.f;
        if (rpos[k] >= 0) {
          const float* rr = region + rpos[k] * AT_PITCH;
#pragma unroll 16
          for (int c = 0; c < AT_CS; c += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&f1s[wave][c0 + c]);     // (same address in every lane: broadcast)
            sum += a[0] * rr[c] + a[1] * rr[c + 1] + a[2] * rr[c + 2] + a[3] * rr[c + 3];
          }
        } else {                                            // outside the staged region: the row comes from L2
          const float* row = f2b + (int64_t)gpos[k] * C + c0;
          for (int c = 0; c < AT_CS && c0 + c < C; c += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&f1s[wave][c0 + c]);
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
            sum += a[0] * v[0] + a[1] * v[1] + a[2] * v[2] + a[3] * v[3];
          }
        }
        acc[k] += sum;
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NRND; ++k)
      if (lane + 64 * k < NPOS) dots[wave][lane + 64 * k] = acc[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (active) {
      float* o = out + q * CH + l * RD * RD;
      for (int oc = lane; oc < RD * RD; oc += 64) {
        const int iyo = oc % RD, ixo = oc / RD;          // channel = iy + RD * ix (x offset slow, as the reference)
        const float* d = dots[wave] + iyo * WIN + ixo;
        o[oc] = scale * ((1.f - dy) * (1.f - dx) * d[0] + (1.f - dy) * dx * d[1] + dy * (1.f - dx) * d[WIN] + dy * dx * d[WIN + 1]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The tile lookup on the matrix pipe.  Per tile of 6 x 4 queries and pyramid level, the dot products of
// the tile's queries with a REGION of target rows are one small GEMM -- region rows (<= 256) x queries (24 of 32) x C -- on
// the record core (gemm_rec.hpp: operands pre-split to [hi | lo] bf16 once per pair, staged by LDS-DMA, bf16x3 products,
// fp32 accumulation: the arithmetic of the volume build this path must agree with).  The region is anchored at the
// component-wise minimum of the tile's window origins and sized per level for the spread of a smooth flow plus two cells
// of slack (17 x 15, 14 x 13, 13 x 12, 12 x 12: coarser levels see the tile shrink); rows outside the image are not
// fetched (zero fill = the reference's zero padding).  The products are parked in LDS [query][region cell]; every wave then
// gathers its queries' (2r+2)^2 window positions out of them -- a position the region does not cover (flow discontinuity
// inside the tile) is computed the slow way from the fp32 maps -- and blends the (2r+1)^2 outputs as before.
// One workgroup (4 waves) per (tile, level): 1248 workgroups of ~8 k-tiles at 47 x 156.
using AM = RecCfg<256, 32, 4, 1>;
constexpr int AM_TW = 6, AM_TH = 4, AM_NQ = AM_TW * AM_TH;
constexpr int AM_DP = 257;                                  // pitch of a query's row of parked products (odd: conflict-free)

struct AltRecLevels {
  const char* f2r[4];       // [B][h_l * w_l][C / 32] records of the pooled target maps
  const float* f2[4];       // the same maps in fp32 channels-last (positions outside the region)
  int h[4], w[4];
  const unsigned* am2[4];   // the amax word each level's records were split with (NULL: scale 1)
};

template <int R>
__global__ __launch_bounds__(256) void altcorr_mfma_fwd_kernel(const char* __restrict__ f1r, const float* __restrict__ f1, AltRecLevels lv,
                                                               AltCoords co, float* __restrict__ out, int nlev, int H, int W, int C,
                                                               float scale, const unsigned* am1,      // am1: the word f1r was split with
                                                               const int* __restrict__ regime) {
  constexpr int RD = 2 * R + 1, WIN = RD + 1, NPOS = WIN * WIN, NRND = (NPOS + 63) / 64;
  if (regime && regime[0] != 0) return;                    // (dispatched launch: rough flow goes to the fp32 tile kernel)
  constexpr int PARK = AM_NQ * AM_DP * 4 + AM_NQ * (NPOS + 1) * 4;      // parked products + every query's window values
  __shared__ __attribute__((aligned(1024))) char lds[AM::LDS_BYTES > PARK ? AM::LDS_BYTES : PARK];
  __shared__ int org[2];
  __shared__ float qxy[AM_NQ][2];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tiles_x = (W + AM_TW - 1) / AM_TW;
  const int b = blockIdx.z, l = blockIdx.y, tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const float rec_inv = fs_inv_scale(fs_scale_of_amax(fs_amax_load(am1))) * fs_inv_scale(fs_scale_of_amax(fs_amax_load(lv.am2[l])));
  const int H2 = lv.h[l], W2 = lv.w[l], N = H * W;
  const unsigned pitch = (unsigned)C * 4u;
  const int RW = l == 0 ? AM_TW + 2 * R + 3 : l == 1 ? 14 : l == 2 ? 13 : 12;       // region width / height per level
  const int RH = l == 0 ? AM_TH + 2 * R + 3 : l == 1 ? 13 : 12;
  const float sl = 1.0f / (float)(1 << l);
  // window origins of the tile's queries at this level; the region starts at their component-wise minimum
  if (wave == 0) {
    const int n = lane < AM_NQ ? lane : 0;
    const int qx = tx * AM_TW + n % AM_TW, qy = ty * AM_TH + n / AM_TW;
    const bool act = lane < AM_NQ && qx < W && qy < H;
    float cx = 0.f, cy = 0.f;
    if (act) {
      const int pix = qy * W + qx;
      cx = co.p[b * co.bs + pix * co.ps]; cy = co.p[b * co.bs + co.cs + pix * co.ps];
      if (co.grid_w > 0) { cx += (float)qx; cy += (float)qy; }
      cx *= sl; cy *= sl;
      cx = (cx > -30000.f && cx < 30000.f) ? cx : -30000.f;
      cy = (cy > -30000.f && cy < 30000.f) ? cy : -30000.f;
    }
    if (lane < AM_NQ) { qxy[lane][0] = cx; qxy[lane][1] = cy; }
    int mx = act ? (int)floorf(cx) - R : 0x7fffffff, my = act ? (int)floorf(cy) - R : 0x7fffffff;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) { mx = min(mx, __shfl_xor(mx, d, 64)); my = min(my, __shfl_xor(my, d, 64)); }
    if (lane == 0) { org[0] = mx; org[1] = my; }
  }
  __syncthreads();
  const int rx0 = org[0], ry0 = org[1];

  // ---- region rows (A, 256) x queries (B, 32) x C on the record core
  RecOperands<AM> o;
  o.da = rec_desc(lv.f2r[l] + (int64_t)b * H2 * W2 * pitch, (unsigned)min((int64_t)H2 * W2 * pitch, (int64_t)0x7fffffff));
  o.db = rec_desc(f1r + (int64_t)b * N * pitch, (unsigned)min((int64_t)N * pitch, (int64_t)0x7fffffff));
  o.b_step = 128u;
  RecPlainA<AM> pa;
#pragma unroll
  for (int j = 0; j < AM::NPA; ++j) {
    const int m = (wave + AM::NWAVE * j) * 8 + (lane >> 3);
    const int uy = m / RW, ux = m - uy * RW, gx = rx0 + ux, gy = ry0 + uy;
    const int ls = (lane & 7) ^ ((m >> 1) & 7);
    pa.va[j] = (uy < RH && gx >= 0 && gx < W2 && gy >= 0 && gy < H2) ? (unsigned)(gy * W2 + gx) * pitch + (unsigned)ls * 16u : 0x80000000u;
  }
  pa.kt0 = 0; pa.step = 128u;
  {
    const int n = wave * 8 + (lane >> 3);
    const int qx = tx * AM_TW + n % AM_TW, qy = ty * AM_TH + n / AM_TW;
    const int ls = (lane & 7) ^ ((n >> 1) & 7);
    o.vb[0] = (n < AM_NQ && qx < W && qy < H) ? (unsigned)(qy * W + qx) * pitch + (unsigned)ls * 16u : 0x80000000u;
  }
  f32x16 acc[AM::TM][AM::TN];
#pragma unroll
  for (int a = 0; a < AM::TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;
  rec_mainloop<AM>(lds, o, pa, 0, C / 32, acc);
  // park [query][region cell] (the ring is free: rec_mainloop ends behind a barrier)
  float* dots = reinterpret_cast<float*>(lds);
  {
    const int l31 = lane & 31, lh = lane >> 5;
    if (l31 < AM_NQ) {
#pragma unroll
      for (int mt = 0; mt < AM::TM; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dots[l31 * AM_DP + wave * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc[mt][0][r] * rec_inv;
    }
  }
  __syncthreads();

  // ---- all four waves over all (query, window position) pairs, then over all (query, output channel) pairs (round 5).
  // Round 3 gave every wave six queries and walked them one after the other -- gather 100 window positions, fence, blend 81
  // channels, fence: ~29 latency-bound (query, level) rounds per wave at four waves per CU, which is what the 55-60 us of a lookup
  // were (larger query tiles at the coarse levels, which halve the operand staging, changed nothing: measured).  Now the window
  // values of ALL the tile's queries go to LDS in one pass of the 256 threads (the ring is free; a position outside the staged
  // region -- flow discontinuity -- is still computed in fp32 from the channels-last maps by the thread that owns it), one barrier,
  // and the blends of all queries run in one pass.
  const int CH = nlev * RD * RD;
  const float* f2b = lv.f2[l] + (int64_t)b * H2 * W2 * C;
  float* wins = dots + AM_NQ * AM_DP;                       // [AM_NQ][NPOS], behind the parked products
  constexpr int WP = NPOS + 1;
  for (int e = threadIdx.x; e < AM_NQ * NPOS; e += 256) {
    const int n = e / NPOS, p = e - n * NPOS;
    const int qx = tx * AM_TW + n % AM_TW, qy = ty * AM_TH + n / AM_TW;
    float v = 0.f;
    if (qx < W && qy < H) {
      const float cx = qxy[n][0], cy = qxy[n][1];
      const int wx0 = (int)floorf(cx) - R, wy0 = (int)floorf(cy) - R;
      const int iy = p / WIN, ix = p - iy * WIN;
      const int gx = wx0 + ix, gy = wy0 + iy, ux = gx - rx0, uy = gy - ry0;
      if (gx >= 0 && gx < W2 && gy >= 0 && gy < H2) {
        if (ux >= 0 && ux < RW && uy >= 0 && uy < RH) {
          v = dots[n * AM_DP + uy * RW + ux];
        } else {                                           // outside the staged region: fp32 rows from L2
          const float* a = f1 + ((int64_t)b * N + qy * W + qx) * C;
          const float* row = f2b + (int64_t)(gy * W2 + gx) * C;
          for (int c = 0; c < C; c += 4) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a + c), rv = *reinterpret_cast<const f32x4*>(row + c);
            v += av[0] * rv[0] + av[1] * rv[1] + av[2] * rv[2] + av[3] * rv[3];
          }
        }
      }
    }
    wins[n * WP + p] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < AM_NQ * RD * RD; e += 256) {
    const int n = e / (RD * RD), oc = e - n * (RD * RD);
    const int qx = tx * AM_TW + n % AM_TW, qy = ty * AM_TH + n / AM_TW;
    if (qx >= W || qy >= H) continue;
    const float cx = qxy[n][0], cy = qxy[n][1];
    const float dx = cx - floorf(cx), dy = cy - floorf(cy);
    const int iyo = oc % RD, ixo = oc / RD;                 // channel = iy + RD * ix (x offset slow, as the reference)
    const float* d = wins + n * WP + iyo * WIN + ixo;
    out[((int64_t)b * N + qy * W + qx) * CH + l * RD * RD + oc] =
        scale * ((1.f - dy) * (1.f - dx) * d[0] + (1.f - dy) * dx * d[1] + dy * (1.f - dx) * d[WIN] + dy * dx * d[WIN + 1]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Which of the two lookups a launch should use.  The matrix-pipe kernel multiplies a tile's 24 queries with
// ONE region of target rows per level (17 x 15 cells at level 0: the windows of a smooth flow + two cells); a window position
// outside the region costs a C-long fp32 dot product from L2 in one thread.  On a flow whose neighbouring queries look at
// unrelated places most positions go that way and the kernel is 1.4x SLOWER than the fp32 tile kernel, which it beats 3x on
// smooth flow (profiles/r06_altcorr_regimes.txt).  The statistic: over all 6 x 4 tiles, the number of queries whose level-0
// window leaves their tile's region, counted among the queries whose window meets the image at all (a window outside the
// image costs neither kernel anything), as a fraction u of ALL queries.  regime[0] = (100 u > g_alt_rough_pct) -- the two
// kernels cross at u ~ 0.7; regime[1..3]: uncovered / in-image queries and the ticket of the last-workgroup-decides
// reduction, zero between launches.
template <int R>
__global__ __launch_bounds__(256) void altcorr_regime_kernel(AltCoords co, int B, int H, int W, int rough_pct, int* __restrict__ regime) {
  constexpr int WIN = 2 * R + 2, RW = AM_TW + 2 * R + 3, RH = AM_TH + 2 * R + 3;
  __shared__ int red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const int tiles_x = (W + AM_TW - 1) / AM_TW, tiles_y = (H + AM_TH - 1) / AM_TH, ntile = B * tiles_x * tiles_y;
  int unc = 0, cnt = 0;
  // a tile per half wave (24 of its 32 lanes), two tiles per wave and trip
  for (int t = (blockIdx.x * 4 + wave) * 2 + half; t < ntile; t += gridDim.x * 8) {
    const int b = t / (tiles_x * tiles_y), tt = t - b * tiles_x * tiles_y, ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int n = l31 < AM_NQ ? l31 : 0;
    const int qx = tx * AM_TW + n % AM_TW, qy = ty * AM_TH + n / AM_TW;
    const bool act = l31 < AM_NQ && qx < W && qy < H;
    float cx = 0.f, cy = 0.f;
    if (act) {
      const int pix = qy * W + qx;
      cx = co.p[b * co.bs + pix * co.ps]; cy = co.p[b * co.bs + co.cs + pix * co.ps];
      if (co.grid_w > 0) { cx += (float)qx; cy += (float)qy; }
      cx = (cx > -30000.f && cx < 30000.f) ? cx : -30000.f;
      cy = (cy > -30000.f && cy < 30000.f) ? cy : -30000.f;
    }
    const int ox = (int)floorf(cx) - R, oy = (int)floorf(cy) - R;
    int mx = act ? ox : 0x7fffffff, my = act ? oy : 0x7fffffff;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) { mx = min(mx, __shfl_xor(mx, d, 64)); my = min(my, __shfl_xor(my, d, 64)); }   // (stays inside the half)
    const bool meets = act && ox + WIN > 0 && ox < W && oy + WIN > 0 && oy < H;
    const bool out = meets && (ox - mx + WIN > RW || oy - my + WIN > RH);
    unc += out ? 1 : 0; cnt += meets ? 1 : 0;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { unc += __shfl_xor(unc, d, 64); cnt += __shfl_xor(cnt, d, 64); }
  if (lane == 0) { red[0][wave] = unc; red[1][wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unc = red[0][0] + red[0][1] + red[0][2] + red[0][3]; cnt = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (unc) atomicAdd(&regime[1], unc);
    if (cnt) atomicAdd(&regime[2], cnt);
    __threadfence();
    if (atomicAdd(&regime[3], 1) == (int)gridDim.x - 1) {          // the last workgroup decides and clears the sums for the next launch
      const int u = atomicExch(&regime[1], 0), c = atomicExch(&regime[2], 0);
      atomicExch(&regime[3], 0);
      regime[0] = (rough_pct >= 0 && (int64_t)u * 100 > (int64_t)B * H * W * rough_pct) ? 1 : 0;
      regime[4] = u; regime[5] = c;                                // (kept for reports: what the decision was taken on)
    }
  }
}

int g_alt_tile = 1;       // 1: tile kernel where it applies (C a multiple of 64), 0: wave-per-query kernel
int g_alt_rough_pct = 65; // dispatched launches: fp32 tile kernel when more than this per cent of the queries leave their tile's region (-1: never)

}  // namespace

extern "C" int fsraft_altcorr_fwd(const float* fmap1, const float* fmap2, const float* coords, float* corr, int B, int N,
                                  int H1, int W1, int H2, int W2, int C, int radius, hipStream_t stream) {
  if (!fmap1 || !fmap2 || !coords || !corr || B < 1 || N < 1 || H1 < 1 || W1 < 1 || H2 < 1 || W2 < 1 || C < 1 || C > 64 * MAXK)
    return FS_ERR_ARG;
  const int64_t nq = (int64_t)B * H1 * W1;
  dim3 grid((unsigned)((nq + 3) / 4));
  if (radius == 4) hipLaunchKernelGGL(altcorr_fwd_kernel<4>, grid, dim3(256), 0, stream, fmap1, fmap2, coords, corr, B, N, H1, W1, H2, W2, C);
  else if (radius == 3) hipLaunchKernelGGL(altcorr_fwd_kernel<3>, grid, dim3(256), 0, stream, fmap1, fmap2, coords, corr, B, N, H1, W1, H2, W2, C);
  else return FS_ERR_ARG;
  return fs_launch_status();
}

extern "C" int fsraft_altcorr_bwd(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad,
                                  float* fmap1_grad, float* fmap2_grad, int B, int N, int H1, int W1, int H2, int W2,
                                  int C, int radius, hipStream_t stream) {
  if (!fmap1 || !fmap2 || !coords || !corr_grad || !fmap1_grad || !fmap2_grad || B < 1 || N < 1 || C < 1 || C > 64 * MAXK)
    return FS_ERR_ARG;
  const int64_t nq = (int64_t)B * H1 * W1;
  dim3 grid((unsigned)((nq + 3) / 4));
  if (radius == 4) hipLaunchKernelGGL(altcorr_bwd_kernel<4>, grid, dim3(256), 0, stream, fmap1, fmap2, coords, corr_grad, fmap1_grad, fmap2_grad, B, N, H1, W1, H2, W2, C);
  else if (radius == 3) hipLaunchKernelGGL(altcorr_bwd_kernel<3>, grid, dim3(256), 0, stream, fmap1, fmap2, coords, corr_grad, fmap1_grad, fmap2_grad, B, N, H1, W1, H2, W2, C);
  else return FS_ERR_ARG;
  return fs_launch_status();
}

// f1 [B,H,W,C] channels-last, f2[l] [B, H >> l, W >> l, C] channels-last (l < num_levels <= 4), coords element (b, c, pix) at
// coords[b*bs + c*cs + pix*ps] (add_grid != 0: the tensor holds the flow); out [B,H,W,num_levels*(2r+1)^2] = the lookup of
// CorrBlock on the same maps (scaled by 1/sqrt(C)), without the volume.
static int altcorr_fp32_fwd(const float* fmap1, const float* const* fmap2_levels, int num_levels, const float* coords,
                            int64_t coords_bs, int64_t coords_cs, int64_t coords_ps, int add_grid, float* out, int B,
                            int H, int W, int C, int radius, const int* regime, hipStream_t stream) {   // regime: run only if regime[0] != 0
  if (!fmap1 || !fmap2_levels || !coords || !out || num_levels < 1 || num_levels > 4 || B < 1 || H < 1 || W < 1 || C < 1 ||
      C > 64 * MAXK)
    return FS_ERR_ARG;
  AltLevels lv;
  int h = H, w = W;
  for (int l = 0; l < 4; ++l) {
    lv.f2[l] = l < num_levels ? fmap2_levels[l] : nullptr;
    lv.h[l] = h; lv.w[l] = w;
    if (l < num_levels && (!fmap2_levels[l] || h < 1 || w < 1)) return FS_ERR_ARG;
    h /= 2; w /= 2;
  }
  AltCoords co{coords, coords_bs, coords_cs, coords_ps, add_grid ? W : 0};
  const int64_t nq = (int64_t)B * H * W;
  dim3 grid((unsigned)((nq + 3) / 4));
  const float scale = 1.0f / sqrtf((float)C);
  if (g_alt_tile && C % 64 == 0 && C <= 256 && ((uintptr_t)fmap1 % 16) == 0) {
    dim3 tg((unsigned)(((W + AT_TQ - 1) / AT_TQ) * ((H + AT_TQ - 1) / AT_TQ)), (unsigned)B);
    if (radius == 4) hipLaunchKernelGGL(altcorr_tile_fwd_kernel<4>, tg, dim3(1024), 0, stream, fmap1, lv, co, out, num_levels, H, W, C, scale, regime);
    else if (radius == 3) hipLaunchKernelGGL(altcorr_tile_fwd_kernel<3>, tg, dim3(1024), 0, stream, fmap1, lv, co, out, num_levels, H, W, C, scale, regime);
    else return FS_ERR_ARG;
    return fs_launch_status();
  }
  if (radius == 4) hipLaunchKernelGGL(altcorr_fused_fwd_kernel<4>, grid, dim3(256), 0, stream, fmap1, lv, co, out, num_levels, B, H * W, C, scale, regime);
  else if (radius == 3) hipLaunchKernelGGL(altcorr_fused_fwd_kernel<3>, grid, dim3(256), 0, stream, fmap1, lv, co, out, num_levels, B, H * W, C, scale, regime);
  else return FS_ERR_ARG;
  return fs_launch_status();
}

extern "C" int fsraft_altcorr_fused_fwd(const float* fmap1, const float* const* fmap2_levels, int num_levels, const float* coords,
                                        int64_t coords_bs, int64_t coords_cs, int64_t coords_ps, int add_grid, float* out, int B,
                                        int H, int W, int C, int radius, hipStream_t stream) {
  return altcorr_fp32_fwd(fmap1, fmap2_levels, num_levels, coords, coords_bs, coords_cs, coords_ps, add_grid, out, B, H, W, C, radius, nullptr, stream);
}

// The same lookup with the region products on the matrix pipe: f1r [B][H*W][C/32] and f2r_levels[l] [B][(H>>l)*(W>>l)][C/32]
// are the records (fsraft_to_records) of the channels-last maps, which are passed too (window positions a tile's region
// does not cover are taken from them).  C % 32 == 0, C <= 256.
extern "C" int fsraft_altcorr_mfma_fwd(const void* f1r, const void* const* f2r_levels, const float* fmap1, const float* const* fmap2_levels,
                                       int num_levels, const float* coords, int64_t coords_bs, int64_t coords_cs, int64_t coords_ps,
                                       int add_grid, float* out, int B, int H, int W, int C, int radius, const unsigned* amax1,
                                       const unsigned* const* amax2_levels,      // the words f1r / each f2r level were split with
                                       int* regime, hipStream_t stream) {        // 8 zeroed ints: dispatch by flow regime; NULL: always this kernel
  if (!f1r || !f2r_levels || !fmap1 || !fmap2_levels || !coords || !out || num_levels < 1 || num_levels > 4 || B < 1 || H < 1 || W < 1 ||
      C < 32 || C % 32 || C > 256 || ((uintptr_t)f1r % 16) || ((uintptr_t)fmap1 % 16) || (int64_t)H * W * C * 4 >= 0x7fffffff)
    return FS_ERR_ARG;
  AltRecLevels lv;
  int h = H, w = W;
  for (int l = 0; l < 4; ++l) {
    lv.f2r[l] = l < num_levels ? (const char*)f2r_levels[l] : nullptr;
    lv.f2[l] = l < num_levels ? fmap2_levels[l] : nullptr;
    lv.am2[l] = (l < num_levels && amax2_levels) ? amax2_levels[l] : nullptr;
    lv.h[l] = h; lv.w[l] = w;
    if (l < num_levels && (!f2r_levels[l] || !fmap2_levels[l] || h < 1 || w < 1 || ((uintptr_t)f2r_levels[l] % 16) || ((uintptr_t)fmap2_levels[l] % 16)))
      return FS_ERR_ARG;
    h /= 2; w /= 2;
  }
  AltCoords co{coords, coords_bs, coords_cs, coords_ps, add_grid ? W : 0};
  dim3 grid((unsigned)(((W + AM_TW - 1) / AM_TW) * ((H + AM_TH - 1) / AM_TH)), (unsigned)num_levels, (unsigned)B);
  const float scale = 1.0f / sqrtf((float)C);
  if (radius != 3 && radius != 4) return FS_ERR_ARG;
  if (regime) {
    if ((uintptr_t)regime & 3) return FS_ERR_ARG;
    const int ntile = (int)(grid.x * (unsigned)B);
    const dim3 rg((unsigned)((ntile + 7) / 8 < 64 ? (ntile + 7) / 8 : 64));
    if (radius == 4) hipLaunchKernelGGL(altcorr_regime_kernel<4>, rg, dim3(256), 0, stream, co, B, H, W, g_alt_rough_pct, regime);
    else hipLaunchKernelGGL(altcorr_regime_kernel<3>, rg, dim3(256), 0, stream, co, B, H, W, g_alt_rough_pct, regime);
  }
  if (radius == 4) hipLaunchKernelGGL(altcorr_mfma_fwd_kernel<4>, grid, dim3(256), 0, stream, (const char*)f1r, fmap1, lv, co, out, num_levels, H, W, C, scale, amax1, regime);
  else hipLaunchKernelGGL(altcorr_mfma_fwd_kernel<3>, grid, dim3(256), 0, stream, (const char*)f1r, fmap1, lv, co, out, num_levels, H, W, C, scale, amax1, regime);
  int rc = fs_launch_status();
  if (rc || !regime) return rc;
  return altcorr_fp32_fwd(fmap1, fmap2_levels, num_levels, coords, coords_bs, coords_cs, coords_ps, add_grid, out, B, H, W, C, radius, regime, stream);
}

extern "C" int fsraft_set_alt_tile(int on) {
  g_alt_tile = on;
  return FS_OK;
}

extern "C" int fsraft_set_alt_rough_pct(int pct) {
  g_alt_rough_pct = pct;
  return FS_OK;
}
