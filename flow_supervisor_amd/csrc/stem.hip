// The encoders' stem: 7x7 stride-2 convolution of the 3-channel image (pytorch/core/extractor.py:125, 201: conv1), forward
// and weight gradient, on planar NCHW images and channels-last [B][Ho][Wo][N] outputs (N = 64 BasicEncoder, 32 SmallEncoder).
//
// MIOpen ran this layer as an NHWC implicit GEMM with layout copies around it (image in, 230 MB result out, and once more
// into this library's channels-last layout; backward the same in reverse plus a 330 us backward-weights kernel): 0.9 ms
// forward + backward per 8 x 440 x 1024 batch for 17 GFLOP.  Both directions are bound by the 230 MB activation that is
// written resp. read once; the GEMM is small (K = 147).
//
// GEMM view.  k' = (c, ky, kx) with ky and kx padded from 7 to 8: K' = 3 x 8 x 8 = 192, so that the eight kx of one (c, ky)
// are eight CONSECUTIVE pixels of one image row -- an MFMA operand fragment (8 consecutive k of one row) of the im2col
// matrix is 16 bytes of the input patch, no gather.  Split-bf16 arithmetic as everywhere (3 bf16 MFMAs per product).
#include "common.hpp"
#include "gemm_core_split.hpp"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ST_TW = 32;                        // output columns per tile
constexpr int ST_K = 192;                        // 3 x 8 x 8
constexpr int ST_PC = 72;                        // patch columns: 2 * 32 + 5 = 69 used by the taps, + 1 for the padded kx, rounded up

struct StemArgs {
  const float* x;                                // [B][3][H][W]
  const float* w;                                // [N][3][7][7]
  const float* bias;                             // [N] or null
  float* out;                                    // forward: [B][Ho][Wo][N]
  const float* dy;                               // weight gradient: [B][Ho][Wo][N]
  float* scratch;                                // weight gradient: [slots][64][192] partial sums
  int B, H, W, Ho, Wo, N, tx, ty, ntiles;
};

__device__ __forceinline__ void split1(float v, unsigned short& hi, unsigned short& lo) {
  const __bf16 h = (__bf16)v;
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
}

// ---------------------------------------------------------------- forward
// A workgroup owns 8 x 32 output pixels; wave w the tile's output row w, all N channels.  The packed weights ([n][k'] bf16
// hi / lo, 400-byte pitch: the 16-byte fragments of 32 rows fall on different bank groups) are converted once per workgroup
// and stay in LDS; the workgroup then walks over tiles: input patch (22 x 70 pixels x 3 planes) -> bf16 hi / lo planes,
// twelve k-steps straight out of the planes, results stored from the accumulators (32 lanes = 128 contiguous bytes).
constexpr int SF_TH = 8, SF_PR = 2 * SF_TH + 6, SF_WPITCH = 400;
constexpr int SF_WBYTES = 64 * SF_WPITCH, SF_PBYTES = 3 * SF_PR * ST_PC * 2;

__global__ __launch_bounds__(512) void stem_fwd_kernel(const StemArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[2 * SF_WBYTES + 2 * SF_PBYTES];
  char* wh = lds;
  char* wl = lds + SF_WBYTES;
  char* ph = lds + 2 * SF_WBYTES;
  char* pl = ph + SF_PBYTES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
  for (int e = threadIdx.x; e < 64 * ST_K; e += 512) {
    const int n = e / ST_K, k = e - n * ST_K, c = k >> 6, ky = (k >> 3) & 7, kx = k & 7;
    const float v = (n < a.N && ky < 7 && kx < 7) ? a.w[((n * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
    unsigned short h, l;
    split1(v, h, l);
    *reinterpret_cast<unsigned short*>(wh + n * SF_WPITCH + k * 2) = h;
    *reinterpret_cast<unsigned short*>(wl + n * SF_WPITCH + k * 2) = l;
  }
  const int nb = (a.N + 31) >> 5;                // column blocks: 1 or 2
  const int tpi = a.tx * a.ty;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / tpi, tr = tile - b * tpi, oy0 = (tr / a.tx) * SF_TH, ox0 = (tr % a.tx) * ST_TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const float* img = a.x + (int64_t)b * 3 * a.H * a.W;
    __syncthreads();                             // the previous tile's fragment reads are done (first pass: the weights are written)
    for (int e = threadIdx.x; e < 3 * SF_PR * ST_PC; e += 512) {
      const int c = e / (SF_PR * ST_PC), r = e - c * (SF_PR * ST_PC), py = r / ST_PC, px = r - py * ST_PC;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const float v = ok ? img[((int64_t)c * a.H + iy) * a.W + ix] : 0.f;
      unsigned short h, l;
      split1(v, h, l);
      *reinterpret_cast<unsigned short*>(ph + e * 2) = h;
      *reinterpret_cast<unsigned short*>(pl + e * 2) = l;
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < ST_K / 16; ++s) {
      const int g = 2 * s + lh, c = g >> 3, ky = g & 7;              // this lane's (c, ky): eight kx = eight consecutive patch pixels
      const int ao = ((c * SF_PR + 2 * wave + ky) * ST_PC + 2 * l31) * 2;   // (4-byte aligned: four dword reads)
      u32x4 ahv, alv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ahv[i] = *reinterpret_cast<const unsigned*>(ph + ao + 4 * i);
        alv[i] = *reinterpret_cast<const unsigned*>(pl + ao + 4 * i);
      }
      const bf16x8 ah = __builtin_bit_cast(bf16x8, ahv), al = __builtin_bit_cast(bf16x8, alv);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j < nb) {
          const int bo = (j * 32 + l31) * SF_WPITCH + (s * 16 + lh * 8) * 2;
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wh + bo), bl = *reinterpret_cast<const bf16x8*>(wl + bo);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
        }
      }
    }
    const int oy = oy0 + wave;
    if (oy < a.Ho) {
      float* orow = a.out + ((int64_t)(b * a.Ho + oy) * a.Wo) * a.N;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = j * 32 + l31;
        if (j < nb && n < a.N) {
          const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (ox < a.Wo) orow[(int64_t)ox * a.N + n] = acc[j][r] + bv;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- weight gradient
//   dW[n][k'] = sum over pixels p of dY[p][n] * im2col(X)[p][k']          (both operands pixel-major: transposed LDS reads)
// A workgroup walks over tiles of 2 x 32 output pixels.  Per tile: dY (64 pixels x N) -> bf16 hi / lo planes [pixel][n]; the
// input patch (10 x 70 x 3) -> LDS as fp32, from there the im2col rows [pixel][k'] (eight consecutive patch pixels per
// (c, ky)) -> bf16 hi / lo planes; four k-steps of 16 pixels.  Waves: 2 (pixel row of the tile) x 2 (n half) x 2 (k' half of
// three 32-column blocks); the 64 x 192 sums of a workgroup stay in registers over all its tiles and leave as two partial
// matrices (one per pixel row) that stem_wgrad_reduce_kernel adds up, dropping the padded taps.
constexpr int SW_TH = 2, SW_PX = SW_TH * ST_TW, SW_PR = 2 * SW_TH + 6;
constexpr int SW_DYPITCH = 64 * 2 + 64, SW_IMPITCH = ST_K * 2 + 64;   // transposed reads: row pitch = data + 64 bytes
constexpr int SW_DYPLANE = SW_PX * SW_DYPITCH, SW_IMPLANE = SW_PX * SW_IMPITCH, SW_PATCH = 3 * SW_PR * ST_PC * 4;
constexpr int SW_NCHP = (3 * SW_PR * ST_PC + 511) / 512;

__global__ __launch_bounds__(512) void stem_wgrad_kernel(const StemArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[2 * SW_DYPLANE + 2 * SW_IMPLANE + SW_PATCH];
  char* dyh = lds;
  char* imh = lds + 2 * SW_DYPLANE;
  float* patch = reinterpret_cast<float*>(lds + 2 * SW_DYPLANE + 2 * SW_IMPLANE);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kg = wave >> 2, mi = wave & 1, ns = (wave >> 1) & 1;       // pixel row of the tile, n half, k' half
  const int lh = lane >> 5, gb = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3, l31 = lane & 31;
  const int tpi = a.tx * a.ty;
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // this thread's share of a tile's global data: two float4 of dY (pixel e / 16, channels 4 (e % 16) ..) and SW_NCHP patch words
  float rdy[8], rp[SW_NCHP];
  auto fetch = [&](int tile) {
    const int b = tile / tpi, tr = tile - b * tpi, oy0 = (tr / a.tx) * SW_TH, ox0 = (tr % a.tx) * ST_TW;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int e = threadIdx.x + 512 * c, px = e >> 4, ch = (e & 15) * 4;
      const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
      const bool ok = oy < a.Ho && ox < a.Wo && ch < a.N;
      const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(a.dy + ((int64_t)(b * a.Ho + oy) * a.Wo + ox) * a.N + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
      rdy[4 * c] = v[0]; rdy[4 * c + 1] = v[1]; rdy[4 * c + 2] = v[2]; rdy[4 * c + 3] = v[3];
    }
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const float* img = a.x + (int64_t)b * 3 * a.H * a.W;
#pragma unroll
    for (int j = 0; j < SW_NCHP; ++j) {
      const int e = threadIdx.x + 512 * j;
      const int c = e / (SW_PR * ST_PC), r = e - c * (SW_PR * ST_PC), py = r / ST_PC, pxx = r - py * ST_PC;
      const int iy = iy0 + py, ix = ix0 + pxx;
      const bool ok = e < 3 * SW_PR * ST_PC && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      rp[j] = ok ? img[((int64_t)c * a.H + iy) * a.W + ix] : 0.f;
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) fetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    // ---- registers -> LDS: dY planes (k-major: [pixel][n]) and the fp32 patch
#pragma unroll
    for (int c = 0; c < 2; ++c) stage_convert_kmajor<64, SW_DYPITCH, SW_DYPLANE>(dyh, threadIdx.x + 512 * c, rdy + 4 * c);
#pragma unroll
    for (int j = 0; j < SW_NCHP; ++j)
      if (threadIdx.x + 512 * j < 3 * SW_PR * ST_PC) patch[threadIdx.x + 512 * j] = rp[j];
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);      // in flight during the rest of this tile
    // ---- im2col rows out of the patch: item = (pixel, (c, ky)): eight consecutive patch pixels -> 16 bytes hi, 16 bytes lo
    for (int it = threadIdx.x; it < SW_PX * 24; it += 512) {
      const int px = it / 24, g = it - px * 24, c = g >> 3, ky = g & 7;
      const float* src = patch + (c * SW_PR + 2 * (px >> 5) + ky) * ST_PC + 2 * (px & 31);
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (ky < 7 && i < 7) ? src[i] : 0.f;
      uint2 h0, l0, h1, l1;
      split4(v, h0, l0);
      split4(v + 4, h1, l1);
      char* dst = imh + px * SW_IMPITCH + g * 16;
      *reinterpret_cast<uint4*>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4*>(dst + SW_IMPLANE) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    __syncthreads();
    // ---- multiply: this wave's pixel row (two k-steps of 16 pixels), n half mi, k' blocks 3 ns .. 3 ns + 2
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int k0 = (kg * 2 + s) * 16;
      const char* Ap = dyh + (k0 + 8 * lh + q) * SW_DYPITCH + (mi * 32 + 16 * gb + 4 * p4) * 2;
      const bf16x8 ah = tr_frag(Ap, SW_DYPITCH), al = tr_frag(Ap + SW_DYPLANE, SW_DYPITCH);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const char* Bp = imh + (k0 + 8 * lh + q) * SW_IMPITCH + ((3 * ns + j) * 32 + 16 * gb + 4 * p4) * 2;
        const bf16x8 bh = tr_frag(Bp, SW_IMPITCH), bl = tr_frag(Bp + SW_IMPLANE, SW_IMPITCH);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
      }
    }
    __syncthreads();                             // the planes are free for the next tile
  }
  float* part = a.scratch + ((int64_t)blockIdx.x * 2 + kg) * 64 * ST_K;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      part[n * ST_K + (3 * ns + j) * 32 + l31] = acc[j][r];
    }
}

__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ scratch, int slots, float* __restrict__ dw, int N) {
  const int e = blockIdx.x * 256 + threadIdx.x;                        // element of dw [N][3][7][7]
  if (e >= N * 147) return;
  const int n = e / 147, r = e - n * 147, c = r / 49, ky = (r % 49) / 7, kx = r % 7;
  const float* p = scratch + n * ST_K + (c * 8 + ky) * 8 + kx;
  float s = 0.f;
  for (int i = 0; i < slots; ++i) s += p[(int64_t)i * 64 * ST_K];
  dw[e] = s;
}

int stem_check(const void* x, const void* w, int B, int H, int W, int N) {
  if (!x || !w || B < 1 || H < 7 || W < 7 || (N != 32 && N != 64)) return FS_ERR_ARG;
  if ((int64_t)B * 3 * H * W >= (int64_t)0x7fffffff) return FS_ERR_ARG;
  return FS_OK;
}

}  // namespace

extern "C" int fsraft_stem_slots(void) { return 512 * 2; }

extern "C" int fsraft_stem7x7s2_fwd(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int N,
                                    hipStream_t stream) {
  const int rc = stem_check(x, w, B, H, W, N);
  if (rc || !out) return rc ? rc : FS_ERR_ARG;
  StemArgs a{};
  a.x = x; a.w = w; a.bias = bias; a.out = out;
  a.B = B; a.H = H; a.W = W; a.N = N;
  a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  a.tx = ceil_div(a.Wo, ST_TW); a.ty = ceil_div(a.Ho, SF_TH);
  a.ntiles = B * a.tx * a.ty;
  const int grid = a.ntiles < 512 ? a.ntiles : 512;                    // two workgroups per CU (70 KB of LDS each)
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid), dim3(512), 0, stream, a);
  return fs_launch_status();
}

// dw [N][3][7][7] is overwritten; scratch: fsraft_stem_slots() * 64 * 192 floats
extern "C" int fsraft_stem7x7s2_wgrad(const float* x, const float* dy, float* dw, float* scratch, int B, int H, int W, int N,
                                      hipStream_t stream) {
  const int rc = stem_check(x, dy, B, H, W, N);
  if (rc || !dw || !scratch || ((uintptr_t)dy & 15)) return rc ? rc : FS_ERR_ARG;
  StemArgs a{};
  a.x = x; a.dy = dy; a.scratch = scratch;
  a.B = B; a.H = H; a.W = W; a.N = N;
  a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  a.tx = ceil_div(a.Wo, ST_TW); a.ty = ceil_div(a.Ho, SW_TH);
  a.ntiles = B * a.tx * a.ty;
  const int grid = a.ntiles < 256 ? a.ntiles : 256;                    // one workgroup per CU (the planes take 91 KB)
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(512), 0, stream, a);
  int rc2 = fs_launch_status();
  if (rc2) return rc2;
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(ceil_div(N * 147, 256)), dim3(256), 0, stream, scratch, grid * 2, dw, N);
  return fs_launch_status();
}
