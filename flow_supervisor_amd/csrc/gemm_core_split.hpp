// Split ("fp16x3", split_arith.hpp) block-GEMM core for gfx950: fp32 operands are scaled by their tensor's power-of-two
// scale and split on the fly into hi = fp16(x s) and lo = fp16(x s - hi); every product is evaluated as
//     a*b ~= (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi) / (s_a s_b)            (fp32 accumulation in the MFMA)
// with three v_mfma_f32_32x32x16_f16 per 32x32x16 block: ~2^-22 relative per product, and the matrix pipe runs 16x
// faster per instruction than v_mfma_f32_32x32x2_f32, i.e. 5.3x faster per useful FLOP.  (Rounds 1-5 split into bf16
// pieces: no scales, 2^-17 per product.)
//
// Operand images in LDS are ROW-major here ([row][k], k contiguous), which is the natural order of
// channels-last activations and of the packed weights, so staging is a straight 16-byte load ->
// convert -> two 8-byte LDS stores, and an MFMA operand (8 consecutive k of one row per lane) is one
// ds_read_b128.  Row pitch = 32 hi + 32 lo bf16 + 16 B pad = 144 B: 9 sixteen-byte slots, odd, so the
// 16 rows of every ds_read_b128 lane group land on 16 different slots (conflict free).
//
// Loaders (duck-typed): fetch_chunk(kt, regs, c) issues one 16-byte load whose address is clamped to
// a zero page when the chunk is outside the matrix (no select on the data, nothing consumes the load
// early); stage_chunk(tile, regs4, c) writes it to the LDS image -- converting fp32 on the fly
// (activations) or copying records that were split when the weights were packed.
// Chunk c of a 256-thread tile is e = tid + 256 c: row e / 8, sixteen-byte column e % 8.
#pragma once
#include "common.hpp"
#include <type_traits>


// SWZ_ = true: rows are stored unpadded (128 B) and the eight 16-byte slots of row r are permuted by
// slot ^ ((r >> 1) & 7) instead -- the 16 rows of a ds_read_b128 lane group then still cover all 64 banks once,
// and a 64x128 tile pair needs 48 KB instead of 54 KB, so THREE workgroups fit the 160 KB of a CU instead of two.
// Loaders may provide fetch_tile(kt, regs) -- all chunks of a k-tile in one call, so that per-tile wave-uniform work
// (table lookup, descriptor, SGPR offset) is done once instead of once per 16-byte chunk; otherwise the main loops
// call fetch_chunk(kt, regs, c) for every chunk.
template <class L, class = void>
struct has_fetch_tile : std::false_type {};
template <class L>
struct has_fetch_tile<L, std::void_t<decltype(&L::fetch_tile)>> : std::true_type {};
template <class L, int N>
__device__ __forceinline__ void fetch_all(const L& l, int kt, float (&r)[N]) {
  if constexpr (has_fetch_tile<L>::value) {
    l.fetch_tile(kt, r);
  } else {
#pragma unroll
    for (int c = 0; c < L::NCH; ++c) l.fetch_chunk(kt, r, c);
  }
}

// NT_ = threads per workgroup (256 or 512).  128x128 tiles with EIGHT waves (each still owning 64x32) need 440 workgroups
// for an N = 256 layer at M = 28160 -- one round on 512 resident slots -- where the 64x128 / four-wave tiles need 880 on
// 768 slots, i.e. a second round that is 15 % full; and every operand byte staged is shared by twice as many MFMAs.
template <int BM_, int BN_, int WM_, int WN_, int NBUF_ = 2, bool SWZ_ = false, int NT_ = 256>
struct SplitCfg {
  static constexpr int NT = NT_;
  static constexpr int NBUF = NBUF_;                              // 2: double-buffered LDS; 1: one image, two barriers per k-tile
  static constexpr int BM = BM_, BN = BN_, BK = 32, WM = WM_, WN = WN_;
  static constexpr bool SWZ = SWZ_;
  static constexpr int PITCH = SWZ_ ? 128 : 144;                  // bytes per row
  static constexpr int TM = BM_ / WM_ / 32, TN = BN_ / WN_ / 32;
  static constexpr int A_BYTES = BM_ * PITCH, B_BYTES = BN_ * PITCH;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = NBUF_ * STAGE;
  static constexpr int EPI_BYTES = BM_ * (BN_ + 4) * 4;                         // fp32 tile parked for the row-wise epilogue
  // (a parked tile that does not fit the CU's 160 KB is not parked: the kernels then take the register epilogue)
  static constexpr int LDS_ALLOC = (LDS_BYTES > EPI_BYTES || EPI_BYTES > 160 * 1024) ? LDS_BYTES : EPI_BYTES;
  static constexpr int NCH_A = BM_ * 8 / NT_, NCH_B = BN_ * 8 / NT_;
  static_assert(WM_ * WN_ * 64 == NT_, "one 64-lane wave per (WM, WN) cell");
};

// 16 bytes of zeros that out-of-range chunks are loaded from instead of being masked afterwards
__device__ __attribute__((aligned(16))) float g_fsraft_zero16[4];

// fp32 x4 -> (hi, lo) bf16 x4.  hi = RNE bf16(x); lo = RNE bf16(x - hi): 3 VALU ops per element
// (v_cvt_pk_bf16_f32 packs two conversions; the back-conversion of hi is a shift / mask).
__device__ __forceinline__ void split4(const float* r, uint2& hi, uint2& lo, float s = 1.0f) { fs_split4(r, s, hi, lo); }

// byte offset of 16-byte slot `slot` (0..7: hi k 0-7, 8-15, 16-23, 24-31, then the same for lo) of row `row`
template <int PITCH>
__device__ __forceinline__ int slot_offset(int row, int slot) {
  return PITCH == 128 ? row * 128 + ((slot ^ ((row >> 1) & 7)) << 4) : row * PITCH + (slot << 4);
}
// staging of one 16-byte chunk e (row = e / 8, k = 4 * (e % 8)) of fp32 data into a [row][hi 32 | lo 32] image
template <int PITCH>
__device__ __forceinline__ void stage_convert(char* tile, int e, const float* r, float s = 1.0f) {
  uint2 hi, lo;
  split4(r, hi, lo, s);
  const int row = e >> 3, kq = e & 7;                 // 8 bytes of hi at slot kq / 2, half kq % 2; lo four slots further
  *reinterpret_cast<uint2*>(tile + slot_offset<PITCH>(row, kq >> 1) + (kq & 1) * 8) = hi;
  *reinterpret_cast<uint2*>(tile + slot_offset<PITCH>(row, 4 + (kq >> 1)) + (kq & 1) * 8) = lo;
}
// staging of pre-split data: chunk e is 16 bytes number (e % 8) of row e / 8's 128-byte [hi | lo] record
template <int PITCH>
__device__ __forceinline__ void stage_copy(char* tile, int e, const float* r) {
  *reinterpret_cast<f32x4*>(tile + slot_offset<PITCH>(e >> 3, e & 7)) = f32x4{r[0], r[1], r[2], r[3]};
}


// STRAIGHT selects the shape of the k-loop (both compute the same thing):
//   false: tail conditions inside the body.  The compiler re-shapes that loop and waits for vmcnt(0) before each
//          staging step -- effectively one tile of look-ahead -- which measures FASTER on the 128x128 tiles
//          (zr 1x5 384->256 at M=28160: 131 us vs 157 us) and on short-K layers (no padded tile, shorter prologue);
//   true:  branch-free body, two tiles per trip, an odd tile count runs one all-zero tile; the waits are the
//          intended vmcnt(15..8) (two tiles in flight).  Faster on the 64x128 tiles (q 1x5 384->128: 76 -> 64 us).
// Experiment hook (-DFSRAFT_SCHED_PATTERN): ask the scheduler for "fragment reads, then one MFMA followed by a few
// VALU / LDS-store / VMEM instructions, repeated" inside each half of the k-loop body.
#ifdef FSRAFT_SCHED_PATTERN
#define FSRAFT_SCHED_GROUPS()                                                            \
  do {                                                                                   \
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                   \
    _Pragma("unroll") for (int sg = 0; sg < 6; ++sg) {                                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                 \
    }                                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                   \
    _Pragma("unroll") for (int sg = 0; sg < 6; ++sg) {                                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                 \
    }                                                                                    \
  } while (0)
#else
#define FSRAFT_SCHED_GROUPS() do {} while (0)
#endif

template <class Cfg, class LA, class LB, bool STRAIGHT = false>
__device__ __forceinline__ void split_mainloop(char* __restrict__ lds, int KT, const LA& la, const LB& lb,
                                               f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int abl = 0;
  static_assert(LA::NCH == Cfg::NCH_A && LB::NCH == Cfg::NCH_B, "loader tile shape must match the config");
  static_assert(Cfg::NBUF == 2, "the two-deep prefetch schedule needs two LDS images");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int l31 = lane & 31, lh = lane >> 5;

  // Two register sets: while tile t is multiplied out of LDS, tile t+1 sits converted-on-arrival in one
  // set and the loads of tile t+2 are in flight into the other.  With the bf16 matrix pipe a k-tile is
  // only ~800 cycles of MFMA, less than one L2/HBM round trip, so a single tile of look-ahead (what the
  // fp32 core uses) leaves the loads exposed; two tiles in flight per wave keep ~16 KB per wave moving.
  float ra0[LA::NREG], rb0[LB::NREG], ra1[LA::NREG], rb1[LB::NREG];

  auto fetch = [&](int kt, float (&ra)[LA::NREG], float (&rb)[LB::NREG]) {
    if (abl & 8) return;
    // past the end: A re-reads the last tile, B is asked for tile -1, which every B loader answers with zeros --
    // so a tile beyond K contributes nothing and the k-loop below needs no tail conditions
    const int ka = kt < KT ? kt : KT - 1, kb = kt < KT ? kt : (STRAIGHT ? -1 : KT - 1);
    fetch_all(la, ka, ra);
    fetch_all(lb, kb, rb);
  };
  auto stage = [&](char* dst, const float (&ra)[LA::NREG], const float (&rb)[LB::NREG]) {
    if (abl & 1) return;
#pragma unroll
    for (int c = 0; c < LA::NCH; ++c) la.stage_chunk(dst, ra + 4 * c, c);
#pragma unroll
    for (int c = 0; c < LB::NCH; ++c) lb.stage_chunk(dst + Cfg::A_BYTES, rb + 4 * c, c);
  };
  auto compute = [&](const char* cur) {
    if (abl & 4) return;
    // this lane's row in the A / B images; slot of (k-step s, hi/lo) = 2 s + lh (+ 4 for lo).  Rows mt*32 apart share
    // (row >> 1) & 7, so the swizzled slot offsets are per-lane constants
    const int ra = wm * (Cfg::TM * 32) + l31, rb = wn * (Cfg::TN * 32) + l31;
    const char* As = cur + ra * Cfg::PITCH;
    const char* Bs = cur + Cfg::A_BYTES + rb * Cfg::PITCH;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      p16x8 ah[Cfg::TM], al[Cfg::TM], bh[Cfg::TN], bl[Cfg::TN];
      const int oah = slot_offset<Cfg::PITCH>(ra, 2 * s + lh) - ra * Cfg::PITCH, oal = slot_offset<Cfg::PITCH>(ra, 4 + 2 * s + lh) - ra * Cfg::PITCH;
      const int obh = slot_offset<Cfg::PITCH>(rb, 2 * s + lh) - rb * Cfg::PITCH, obl = slot_offset<Cfg::PITCH>(rb, 4 + 2 * s + lh) - rb * Cfg::PITCH;
#pragma unroll
      for (int mt = 0; mt < Cfg::TM; ++mt) {
        ah[mt] = *reinterpret_cast<const p16x8*>(As + mt * 32 * Cfg::PITCH + oah);
        al[mt] = *reinterpret_cast<const p16x8*>(As + mt * 32 * Cfg::PITCH + oal);
      }
#pragma unroll
      for (int nt = 0; nt < Cfg::TN; ++nt) {
        bh[nt] = *reinterpret_cast<const p16x8*>(Bs + nt * 32 * Cfg::PITCH + obh);
        bl[nt] = *reinterpret_cast<const p16x8*>(Bs + nt * 32 * Cfg::PITCH + obl);
      }
      if (abl & 2) {      // keep the fragment reads alive without issuing MFMAs
#pragma unroll
        for (int mt = 0; mt < Cfg::TM; ++mt) asm volatile("" ::"v"(ah[mt]), "v"(al[mt]));
#pragma unroll
        for (int nt = 0; nt < Cfg::TN; ++nt) asm volatile("" ::"v"(bh[nt]), "v"(bl[nt]));
        continue;
      }
#pragma unroll
      for (int mt = 0; mt < Cfg::TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::TN; ++nt) {
          acc[mt][nt] = fs_mfma_32x32x16(al[mt], bh[nt], acc[mt][nt]);
          acc[mt][nt] = fs_mfma_32x32x16(ah[mt], bl[nt], acc[mt][nt]);
          acc[mt][nt] = fs_mfma_32x32x16(ah[mt], bh[nt], acc[mt][nt]);
        }
    }
  };

  if (KT <= 0) return;
  char* buf0 = lds;
  char* buf1 = lds + Cfg::STAGE;
  fetch(0, ra0, rb0);
  fetch(1, ra1, rb1);
  stage(buf0, ra0, rb0);           // tile 0 -> LDS
  fetch(2, ra0, rb0);              // tile 2 in flight, tile 1 waiting in set 1
  __syncthreads();
  if constexpr (!STRAIGHT) {
    for (int kt = 0; kt < KT; kt += 2) {
      compute(buf0);                                   // tile kt
      if (kt + 1 < KT) stage(buf1, ra1, rb1);          // tile kt+1 -> other image
      fetch(kt + 3, ra1, rb1);
      __syncthreads();
      if (kt + 1 >= KT) break;
      compute(buf1);                                   // tile kt+1
      if (kt + 2 < KT) stage(buf0, ra0, rb0);          // tile kt+2
      fetch(kt + 4, ra0, rb0);
      __syncthreads();
    }
    return;
  }
  for (int kt = 0; kt < KT; kt += 2) {
    // No scheduling barrier between the MFMA block and the staging code: left alone, the compiler overlaps part of the
    // conversion / LDS-store work with the matrix instructions (measured 2-3 % faster than pinning "MFMAs first").
    compute(buf0);                                   // tile kt
    stage(buf1, ra1, rb1);                           // tile kt+1 -> other image
    fetch(kt + 3, ra1, rb1);
    FSRAFT_SCHED_GROUPS();
    __syncthreads();
    compute(buf1);                                   // tile kt+1 (zeros when kt+1 == KT)
    stage(buf0, ra0, rb0);                           // tile kt+2
    fetch(kt + 4, ra0, rb0);
    FSRAFT_SCHED_GROUPS();
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// "TN" variant: both operands arrive k-major (element (k, m) at base[k*ld + m]), as in the weight
// gradient dW[co][ci] = sum_pixels dY[pixel][co] * X[pixel][ci].  The MFMA wants 8 consecutive k of
// one row per lane, i.e. the transpose of what a coalesced load delivers.  The tiles are staged
// UN-transposed ([k][m] bf16 planes, hi and lo, straight 8-byte stores of converted float4s) and the
// transpose happens for free in the LDS read: ds_read_b64_tr_b16 hands lane i of a 16-lane group
// column m0+i of a 4-row block.  Row pitch 320 B (256 B of data + 64) puts the 4 rows of a block on
// disjoint quarters of the 256-byte bank row, so the transposed reads are conflict free.
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int BM_, int BN_, int WM_, int WN_, int NBUF_ = 2, int NT_ = 256>
struct SplitTnCfg {
  static constexpr int BM = BM_, BN = BN_, BK = 32, WM = WM_, WN = WN_, NBUF = NBUF_, NT = NT_;
  static constexpr int TM = BM_ / WM_ / 32, TN = BN_ / WN_ / 32;
  static constexpr int PA = BM_ * 2 + 64, PB = BN_ * 2 + 64;         // row pitch in bytes
  static constexpr int A_PLANE = 32 * PA, B_PLANE = 32 * PB;
  static constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
  static constexpr int LDS_BYTES = NBUF_ * STAGE;
  static constexpr int NCH_A = 32 * (BM_ / 4) / NT_, NCH_B = 32 * (BN_ / 4) / NT_;
  static_assert(WM_ * WN_ * 64 == NT_, "one 64-lane wave per (WM, WN) cell");
};

// chunk e of a [32][COLS] fp32 tile: k = e / (COLS/4), 4 consecutive columns from 4 * (e % (COLS/4))
template <int COLS, int PITCH, int PLANE>
__device__ __forceinline__ void stage_convert_kmajor(char* tile, int e, const float* r, float s = 1.0f) {
  uint2 hi, lo;
  split4(r, hi, lo, s);
  char* p = tile + (e / (COLS / 4)) * PITCH + (e % (COLS / 4)) * 8;
  *reinterpret_cast<uint2*>(p) = hi;
  *reinterpret_cast<uint2*>(p + PLANE) = lo;
}

__device__ __forceinline__ p16x8 tr_frag(const char* p, int pitch) {
  // two transposed 4x16 blocks: k .. k+3 and k+4 .. k+7 of this lane's column
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * pitch));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(p16x8, v);
}

// COLSUM: additionally accumulate, per thread, the column sums of the A operand (its 4 columns are the same
// for every chunk: (tid + 256 c) % (BM/4) == tid % (BM/4)) -- the bias gradient falls out of the weight
// gradient's own loads.
template <class Cfg, class LA, class LB, bool COLSUM = false>
__device__ __forceinline__ void split_mainloop_tn(char* __restrict__ lds, int KT, const LA& la, const LB& lb,
                                                  f32x16 (&acc)[Cfg::TM][Cfg::TN], float* colsum = nullptr,
                                                  float sa = 1.0f, float sb = 1.0f) {     // sa, sb: scales of the two operands
  static_assert(LA::NCH == Cfg::NCH_A && LB::NCH == Cfg::NCH_B, "loader tile shape must match the config");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int lh = lane >> 5, gb = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;

  float ra[LA::NREG], rb[LB::NREG];
  auto stage = [&](char* dst, bool count) {
#pragma unroll
    for (int c = 0; c < LA::NCH; ++c) {
      stage_convert_kmajor<Cfg::BM, Cfg::PA, Cfg::A_PLANE>(dst, threadIdx.x + Cfg::NT * c, ra + 4 * c, sa);
      if (COLSUM && count) {
#pragma unroll
        for (int q = 0; q < 4; ++q) colsum[q] += ra[4 * c + q];
      }
    }
#pragma unroll
    for (int c = 0; c < LB::NCH; ++c)
      stage_convert_kmajor<Cfg::BN, Cfg::PB, Cfg::B_PLANE>(dst + 2 * Cfg::A_PLANE, threadIdx.x + Cfg::NT * c, rb + 4 * c, sb);
  };
  if (KT > 0) {
    fetch_all(la, 0, ra);
    fetch_all(lb, 0, rb);
    stage(lds, true);
  }
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    char* cur = lds + (Cfg::NBUF == 2 ? (kt & 1) * Cfg::STAGE : 0);
    char* nxt = lds + (Cfg::NBUF == 2 ? ((kt + 1) & 1) * Cfg::STAGE : 0);
    const int ktn = (kt + 1 < KT) ? kt + 1 : kt;
    fetch_all(la, ktn, ra);
    fetch_all(lb, ktn, rb);
    // this lane's address inside a plane: row (8*lh + q) of the k-step, columns base + 16*gb + 4*p4
    const char* As = cur + (8 * lh + q) * Cfg::PA + (wm * (Cfg::TM * 32) + 16 * gb + 4 * p4) * 2;
    const char* Bs = cur + 2 * Cfg::A_PLANE + (8 * lh + q) * Cfg::PB + (wn * (Cfg::TN * 32) + 16 * gb + 4 * p4) * 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      p16x8 ah[Cfg::TM], al[Cfg::TM], bh[Cfg::TN], bl[Cfg::TN];
#pragma unroll
      for (int mt = 0; mt < Cfg::TM; ++mt) {
        ah[mt] = tr_frag(As + s * 16 * Cfg::PA + mt * 64, Cfg::PA);
        al[mt] = tr_frag(As + s * 16 * Cfg::PA + mt * 64 + Cfg::A_PLANE, Cfg::PA);
      }
#pragma unroll
      for (int nt = 0; nt < Cfg::TN; ++nt) {
        bh[nt] = tr_frag(Bs + s * 16 * Cfg::PB + nt * 64, Cfg::PB);
        bl[nt] = tr_frag(Bs + s * 16 * Cfg::PB + nt * 64 + Cfg::B_PLANE, Cfg::PB);
      }
#pragma unroll
      for (int mt = 0; mt < Cfg::TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::TN; ++nt) {
          acc[mt][nt] = fs_mfma_32x32x16(al[mt], bh[nt], acc[mt][nt]);
          acc[mt][nt] = fs_mfma_32x32x16(ah[mt], bl[nt], acc[mt][nt]);
          acc[mt][nt] = fs_mfma_32x32x16(ah[mt], bh[nt], acc[mt][nt]);
        }
    }
    if (Cfg::NBUF == 1) __syncthreads();
    stage(nxt, kt + 1 < KT);        // the last iteration re-stages its own tile: do not count it twice
    __syncthreads();
  }
}
