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

// buffer loads with the validity in the lane offset (an offset past num_records reads zeros): no branch around a load,
// so a thread's loads of a tile are issued together (as conv_igemm.hip's loaders)
constexpr unsigned ST_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t st_rsrc(const void* p) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uint64_t)hi << 32 | lo), 0, 0x7fffffff, 0x00020000);
}

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
  const unsigned* x_amax;                        // amax word of the images (NULL: scale 1, |x| < 2^15)
  const unsigned* dy_amax;                       // weight gradient: amax word of dy
};

__device__ __forceinline__ void split1(float v, unsigned short& hi, unsigned short& lo, float s) {
  _Float16 h, l;
  fs_split1(v, s, h, l);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}

// ---------------------------------------------------------------- forward
// A workgroup owns 8 x 32 output pixels; wave w the tile's output row w, all N channels.  The packed weights ([n][k'] bf16
// hi / lo, 400-byte pitch: the 16-byte fragments of 32 rows fall on different bank groups) are converted once per workgroup
// and stay in LDS; the workgroup then walks over tiles: input patch (22 x 70 pixels x 3 planes) -> bf16 hi / lo planes,
// twelve k-steps straight out of the planes, results stored from the accumulators (32 lanes = 128 contiguous bytes).
constexpr int SF_TH = 8, SF_PR = 2 * SF_TH + 6, SF_WPITCH = 400;
constexpr int SF_WBYTES = 64 * SF_WPITCH, SF_PBYTES = 3 * SF_PR * ST_PC * 2, SF_NCH = (3 * SF_PR * ST_PC + 511) / 512;

__global__ __launch_bounds__(512) void stem_fwd_kernel(const StemArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[2 * SF_WBYTES + 4 * SF_PBYTES];
  char* wh = lds;
  char* wl = lds + SF_WBYTES;
  char* pbuf = lds + 2 * SF_WBYTES;          // two patch buffers of (hi plane, lo plane)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
  // scales (split_arith.hpp): the images' from their word; the weights' from their largest magnitude, which every workgroup
  // works out for itself (it reads all N x 147 of them anyway)
  __shared__ unsigned wred[8];
  unsigned wm = 0u;
  for (int e = threadIdx.x; e < a.N * 147; e += 512) wm = fs_umax(wm, fs_abs_bits(a.w[e]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wm = fs_umax(wm, (unsigned)__shfl_xor((int)wm, o, 64));
  if (lane == 0) wred[wave] = wm;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) wm = fs_umax(wm, wred[i]);
  const float sw = fs_scale_of_amax(wm), sx = fs_scale_of_amax(fs_amax_load(a.x_amax));
  const float inv = fs_inv_scale(sw) * fs_inv_scale(sx);
  for (int e = threadIdx.x; e < 64 * ST_K; e += 512) {
    const int n = e / ST_K, k = e - n * ST_K, c = k >> 6, ky = (k >> 3) & 7, kx = k & 7;
    const float v = (n < a.N && ky < 7 && kx < 7) ? a.w[((n * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
    unsigned short h, l;
    split1(v, h, l, sw);
    *reinterpret_cast<unsigned short*>(wh + n * SF_WPITCH + k * 2) = h;
    *reinterpret_cast<unsigned short*>(wl + n * SF_WPITCH + k * 2) = l;
  }
  const int nb = (a.N + 31) >> 5;                // column blocks: 1 or 2
  const int tpi = a.tx * a.ty;
  // this thread's words of a patch: e = tid + 512 j -> (plane c, patch row, patch column); the next tile's are in flight
  // (registers) while this tile is multiplied
  float rp[SF_NCH];
  int pos[SF_NCH];                               // plane << 16 | patch row << 8 | patch column of word j (-1: beyond the patch)
#pragma unroll
  for (int j = 0; j < SF_NCH; ++j) {
    const int e = threadIdx.x + 512 * j;
    const int c = e / (SF_PR * ST_PC), r = e - c * (SF_PR * ST_PC), py = r / ST_PC, px = r - py * ST_PC;
    pos[j] = e < 3 * SF_PR * ST_PC ? c << 16 | py << 8 | px : -1;
  }
  auto fetch = [&](int tile) {
    const int b = tile / tpi, tr = tile - b * tpi, oy0 = (tr / a.tx) * SF_TH, ox0 = (tr % a.tx) * ST_TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const __amdgpu_buffer_rsrc_t rs = st_rsrc(a.x + (int64_t)b * 3 * a.H * a.W);
#pragma unroll
    for (int j = 0; j < SF_NCH; ++j) {
      const int c = pos[j] >> 16, iy = iy0 + ((pos[j] >> 8) & 255), ix = ix0 + (pos[j] & 255);
      const bool ok = pos[j] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      rp[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (unsigned)((c * a.H + iy) * a.W + ix) * 4u : ST_OOB, 0, 0));
    }
  };
  auto stage = [&](char* ph) {
#pragma unroll
    for (int j = 0; j < SF_NCH; ++j) {
      const int e = threadIdx.x + 512 * j;
      if (e < 3 * SF_PR * ST_PC) {
        unsigned short h, l;
        split1(rp[j], h, l, sx);
        *reinterpret_cast<unsigned short*>(ph + e * 2) = h;
        *reinterpret_cast<unsigned short*>(ph + SF_PBYTES + e * 2) = l;
      }
    }
  };
  // Order inside an iteration: multiply tile t, THEN put tile t + 1 (loaded during the multiply) into the other patch buffer,
  // THEN store tile t.  vmcnt is one in-order counter for loads and stores: a wait for the next tile's loads that comes
  // after this tile's 32 stores waits for the stores to drain (184 us per launch instead of the time below).
  const float bv0 = (a.bias && l31 < a.N) ? a.bias[l31] : 0.f, bv1 = (a.bias && 32 + l31 < a.N) ? a.bias[32 + l31] : 0.f;
  if ((int)blockIdx.x < a.ntiles) {
    fetch(blockIdx.x);
    stage(pbuf);
  }
  __syncthreads();
  int it = 0;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, ++it) {
    const int b = tile / tpi, tr = tile - b * tpi, oy0 = (tr / a.tx) * SF_TH, ox0 = (tr % a.tx) * ST_TW;
    const char* ph = pbuf + (it & 1) * 2 * SF_PBYTES;
    const char* pl = ph + SF_PBYTES;
    const bool more = tile + (int)gridDim.x < a.ntiles;
    if (more) fetch(tile + gridDim.x);
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
      const p16x8 ah = __builtin_bit_cast(p16x8, ahv), al = __builtin_bit_cast(p16x8, alv);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j < nb) {
          const int bo = (j * 32 + l31) * SF_WPITCH + (s * 16 + lh * 8) * 2;
          const p16x8 bh = *reinterpret_cast<const p16x8*>(wh + bo), bl = *reinterpret_cast<const p16x8*>(wl + bo);
          acc[j] = fs_mfma_32x32x16(al, bh, acc[j]);
          acc[j] = fs_mfma_32x32x16(ah, bl, acc[j]);
          acc[j] = fs_mfma_32x32x16(ah, bh, acc[j]);
        }
      }
      if (s & 1) __builtin_amdgcn_sched_barrier(0);   // (or every fragment of all twelve k-steps is loaded up front: 220 registers, one workgroup per CU)
    }
    if (more) stage(pbuf + ((it + 1) & 1) * 2 * SF_PBYTES);
    // stores: one lane offset per column block plus a scalar offset per accumulator row
    const int oy = oy0 + wave;
    if (oy < a.Ho) {
      const __amdgpu_buffer_rsrc_t ro = st_rsrc(a.out + (int64_t)b * a.Ho * a.Wo * a.N);
      const int oxl = ox0 + 4 * lh;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = j * 32 + l31;
        if (j < nb) {
          const float bv = j ? bv1 : bv0;
          const unsigned vo = n < a.N ? (unsigned)((oy * a.Wo + oxl) * a.N + n) * 4u : ST_OOB;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dx = (r & 3) + 8 * (r >> 2);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_fmaf(acc[j][r], inv, bv)), ro, oxl + dx < a.Wo ? vo : ST_OOB,
                                                  (unsigned)(dx * a.N) * 4u, 0);
          }
        }
      }
    }
    __syncthreads();                             // tile t + 1 is in its buffer, nobody reads tile t's any more
  }
}

// ---------------------------------------------------------------- weight gradient
//   dW[n][k'] = sum over pixels p of dY[p][n] * im2col(X)[p][k']          (both operands pixel-major: transposed LDS reads)
// A workgroup walks over tiles of SW_TH x 32 output pixels.  Per tile: dY (pixels x N) -> bf16 hi / lo planes [pixel][n]; the
// input patch ((2 SW_TH + 6) x 70 x 3) -> LDS as fp32, from there the im2col rows [pixel][k'] (eight consecutive patch pixels
// per (c, ky)) -> bf16 hi / lo planes; 2 SW_TH k-steps of 16 pixels.  Waves: 2 (halves of the tile's k-steps) x 2 (n half) x 2
// (k' half of three 32-column blocks); the 64 x 192 sums of a workgroup stay in registers over all its tiles and leave as two
// partial matrices that stem_wgrad_reduce_kernel adds up, dropping the padded taps.  Three barriers per tile and little work
// between them: one-row tiles (48 KB of LDS) with two workgroups per CU measured better than two-row tiles with one.
constexpr int SW_TH = 1, SW_PX = SW_TH * ST_TW, SW_PR = 2 * SW_TH + 6, SW_NDY = SW_PX * 16 / 512;
constexpr int SW_DYPITCH = 64 * 2 + 64, SW_IMPITCH = ST_K * 2 + 64;   // transposed reads: row pitch = data + 64 bytes
constexpr int SW_DYPLANE = SW_PX * SW_DYPITCH, SW_IMPLANE = SW_PX * SW_IMPITCH, SW_PATCH = 3 * SW_PR * ST_PC * 4;
constexpr int SW_NCHP = (3 * SW_PR * ST_PC + 511) / 512;

__global__ __launch_bounds__(512, 2) void stem_wgrad_kernel(const StemArgs a) {
  constexpr int SW_LDS = 2 * SW_DYPLANE + 2 * SW_IMPLANE + SW_PATCH > 64 * ST_K * 4 ? 2 * SW_DYPLANE + 2 * SW_IMPLANE + SW_PATCH : 64 * ST_K * 4;
  __shared__ __attribute__((aligned(16))) char lds[SW_LDS];
  char* dyh = lds;
  char* imh = lds + 2 * SW_DYPLANE;
  float* patch = reinterpret_cast<float*>(lds + 2 * SW_DYPLANE + 2 * SW_IMPLANE);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kg = wave >> 2, mi = wave & 1, ns = (wave >> 1) & 1;       // pixel row of the tile, n half, k' half
  const int lh = lane >> 5, gb = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3, l31 = lane & 31;
  const int tpi = a.tx * a.ty;
  const float sdy = fs_scale_of_amax(fs_amax_load(a.dy_amax)), sx = fs_scale_of_amax(fs_amax_load(a.x_amax));
  const float inv = fs_inv_scale(sdy) * fs_inv_scale(sx);
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // this thread's share of a tile's global data: two float4 of dY (pixel e / 16, channels 4 (e % 16) ..) and SW_NCHP patch words
  float rdy[4 * SW_NDY], rp[SW_NCHP];
  int pos[SW_NCHP];                              // plane << 16 | patch row << 8 | patch column of word j (-1: beyond the patch)
#pragma unroll
  for (int j = 0; j < SW_NCHP; ++j) {
    const int e = threadIdx.x + 512 * j;
    const int c = e / (SW_PR * ST_PC), r = e - c * (SW_PR * ST_PC), py = r / ST_PC, pxx = r - py * ST_PC;
    pos[j] = e < 3 * SW_PR * ST_PC ? c << 16 | py << 8 | pxx : -1;
  }
  auto fetch = [&](int tile) {
    const int b = tile / tpi, tr = tile - b * tpi, oy0 = (tr / a.tx) * SW_TH, ox0 = (tr % a.tx) * ST_TW;
    const __amdgpu_buffer_rsrc_t rd = st_rsrc(a.dy + (int64_t)b * a.Ho * a.Wo * a.N);
#pragma unroll
    for (int c = 0; c < SW_NDY; ++c) {
      const int e = threadIdx.x + 512 * c, px = e >> 4, ch = (e & 15) * 4;
      const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
      const bool ok = oy < a.Ho && ox < a.Wo && ch < a.N;
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, ok ? (unsigned)((oy * a.Wo + ox) * a.N + ch) * 4u : ST_OOB, 0, 0));
      rdy[4 * c] = v[0]; rdy[4 * c + 1] = v[1]; rdy[4 * c + 2] = v[2]; rdy[4 * c + 3] = v[3];
    }
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const __amdgpu_buffer_rsrc_t rs = st_rsrc(a.x + (int64_t)b * 3 * a.H * a.W);
#pragma unroll
    for (int j = 0; j < SW_NCHP; ++j) {
      const int c = pos[j] >> 16, iy = iy0 + ((pos[j] >> 8) & 255), ix = ix0 + (pos[j] & 255);
      const bool ok = pos[j] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      rp[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (unsigned)((c * a.H + iy) * a.W + ix) * 4u : ST_OOB, 0, 0));
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) fetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    // ---- registers -> LDS: dY planes (k-major: [pixel][n]) and the fp32 patch
#pragma unroll
    for (int c = 0; c < SW_NDY; ++c) stage_convert_kmajor<64, SW_DYPITCH, SW_DYPLANE>(dyh, threadIdx.x + 512 * c, rdy + 4 * c, sdy);
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
      split4(v, h0, l0, sx);
      split4(v + 4, h1, l1, sx);
      char* dst = imh + px * SW_IMPITCH + g * 16;
      *reinterpret_cast<uint4*>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4*>(dst + SW_IMPLANE) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    __syncthreads();
    // ---- multiply: this wave's pixel row (two k-steps of 16 pixels), n half mi, k' blocks 3 ns .. 3 ns + 2
#pragma unroll
    for (int s = 0; s < SW_TH; ++s) {
      const int k0 = (kg * SW_TH + s) * 16;
      const char* Ap = dyh + (k0 + 8 * lh + q) * SW_DYPITCH + (mi * 32 + 16 * gb + 4 * p4) * 2;
      const p16x8 ah = tr_frag(Ap, SW_DYPITCH), al = tr_frag(Ap + SW_DYPLANE, SW_DYPITCH);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const char* Bp = imh + (k0 + 8 * lh + q) * SW_IMPITCH + ((3 * ns + j) * 32 + 16 * gb + 4 * p4) * 2;
        const p16x8 bh = tr_frag(Bp, SW_IMPITCH), bl = tr_frag(Bp + SW_IMPLANE, SW_IMPITCH);
        acc[j] = fs_mfma_32x32x16(al, bh, acc[j]);
        acc[j] = fs_mfma_32x32x16(ah, bl, acc[j]);
        acc[j] = fs_mfma_32x32x16(ah, bh, acc[j]);
      }
    }
    __syncthreads();                             // the planes are free for the next tile
  }
  // the two k halves of the workgroup meet in LDS (the planes are free), one partial matrix per workgroup leaves
  float* sum = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int j = 0; j < 3; ++j) acc[j] *= inv;
  __syncthreads();
  if (kg == 1) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ST_K + (3 * ns + j) * 32 + l31] = acc[j][r];
  }
  __syncthreads();
  if (kg == 0) {
    float* part = a.scratch + (int64_t)blockIdx.x * 64 * ST_K;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ST_K + (3 * ns + j) * 32 + l31;
        part[o] = acc[j][r] + sum[o];
      }
  }
}

// dw[e] = sum over the workgroups' partial matrices; 64 elements x 4 interleaved slot ranges per block (a plain
// one-thread-per-element loop over 512 slots was latency-bound: 120 us for 19 MB)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ scratch, int slots, float* __restrict__ dw, int N) {
  __shared__ float part[4][64];
  const int el = threadIdx.x & 63, pr = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;                                  // element of dw [N][3][7][7]
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < N * 147) {
    const int n = e / 147, r = e - n * 147, c = r / 49, ky = (r % 49) / 7, kx = r % 7;
    const float* p = scratch + n * ST_K + (c * 8 + ky) * 8 + kx;
    int i = pr;
    for (; i + 60 < slots; i += 64) {          // sixteen partial matrices per trip, all requested before the first add (same order of adds)
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = p[(int64_t)(i + 4 * j) * 64 * ST_K];
#pragma unroll
      for (int j = 0; j < 16; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
    }
    for (; i + 12 < slots; i += 16) {
      s0 += p[(int64_t)i * 64 * ST_K];
      s1 += p[(int64_t)(i + 4) * 64 * ST_K];
      s2 += p[(int64_t)(i + 8) * 64 * ST_K];
      s3 += p[(int64_t)(i + 12) * 64 * ST_K];
    }
    for (; i < slots; i += 4) s0 += p[(int64_t)i * 64 * ST_K];
  }
  part[pr][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pr == 0 && e < N * 147) dw[e] = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
}

int stem_check(const void* x, const void* w, int B, int H, int W, int N) {
  if (!x || !w || B < 1 || H < 7 || W < 7 || (N != 32 && N != 64)) return FS_ERR_ARG;
  if ((int64_t)3 * H * W * 4 >= (int64_t)0x7fffffff || (int64_t)((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * N * 4 >= (int64_t)0x7fffffff) return FS_ERR_ARG;   // 32-bit byte offsets inside one image
  return FS_OK;
}

}  // namespace

extern "C" int fsraft_stem_slots(void) { return 512; }

extern "C" int fsraft_stem7x7s2_fwd(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int N,
                                    const unsigned* x_amax, hipStream_t stream) {
  const int rc = stem_check(x, w, B, H, W, N);
  if (rc || !out) return rc ? rc : FS_ERR_ARG;
  StemArgs a{};
  a.x = x; a.w = w; a.bias = bias; a.out = out; a.x_amax = x_amax;
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
                                      const unsigned* x_amax, const unsigned* dy_amax, hipStream_t stream) {
  const int rc = stem_check(x, dy, B, H, W, N);
  if (rc || !dw || !scratch || ((uintptr_t)dy & 15)) return rc ? rc : FS_ERR_ARG;
  StemArgs a{};
  a.x = x; a.dy = dy; a.scratch = scratch; a.x_amax = x_amax; a.dy_amax = dy_amax;
  a.B = B; a.H = H; a.W = W; a.N = N;
  a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  a.tx = ceil_div(a.Wo, ST_TW); a.ty = ceil_div(a.Ho, SW_TH);
  a.ntiles = B * a.tx * a.ty;
  const int grid = a.ntiles < 512 ? a.ntiles : 512;                    // two workgroups per CU (48 KB of LDS, 128 registers)
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(512), 0, stream, a);
  int rc2 = fs_launch_status();
  if (rc2) return rc2;
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(ceil_div(N * 147, 64)), dim3(256), 0, stream, scratch, grid, dw, N);
  return fs_launch_status();
}
