// Block-GEMM core on PRE-SPLIT operands ("records"), gfx950.
//
// Arithmetic is the bf16x3 scheme of gemm_core_split.hpp -- every fp32 value x is hi = bf16(x), lo = bf16(x - hi) and a
// product is a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- but the split is
// done ONCE by whoever produces the operand, not by every consumer while staging.  Storage unit: the RECORD, 32
// consecutive k of one row as [32 x bf16 hi | 32 x bf16 lo] = 128 bytes, i.e. exactly the bytes the 32 floats took.
//
// That makes staging a pure copy, so it is done by LDS-DMA (`buffer_load_dwordx4 ... lds`): no VGPR round trip, no
// conversion VALU work, no ds_write.  One wave instruction moves 1 KB = the records of 8 rows.  The LDS side of the
// instruction is fixed at (M0 base + lane * 16), so the bank swizzle of the image (16-byte slot s of row r lives at
// slot s ^ ((r >> 1) & 7), conflict-free for the ds_read_b128 fragment reads) is applied by permuting which 16 bytes a
// lane FETCHES.  A lane's source offset is constant over the k-loop (row * pitch + slot * 16); the k-tile advances
// through the instruction's scalar offset.
//
// Pipeline: a ring of NSLOT k-tile images, NSLOT - 1 tiles in flight.  The DMAs are issued from inline asm -- the
// compiler would wait vmcnt(0) before every LDS read that follows an LDS-DMA it knows about -- and retired by a counted
// s_waitcnt vmcnt(N) followed by ONE raw s_barrier per k-tile:
// (schedule: see rec_mainloop).
#pragma once
#include "common.hpp"
#include <type_traits>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor (wave-uniform inputs only): base, 0 stride, `bytes` records of 1 byte, default data format
__device__ __forceinline__ u32x4 rec_desc(const void* base, unsigned bytes) {
  const uint64_t a = (uint64_t)base;
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}

// one LDS-DMA piece: lane i fetches 16 bytes at desc.base + voff + soff and they land at LDS address lds + 16 i
__device__ __forceinline__ void rec_dma16(unsigned voff, u32x4 desc, unsigned soff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(desc), "s"(soff), "s"(lds)
               : "memory");
}
template <int N>
__device__ __forceinline__ void rec_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 8 waves, WM x WN of them, each computing (BM / WM) x (BN / WN) of the BM x BN tile out of [row][128-byte record] images
// KM_ = false: both operands are rows of records along k ("NT": C = A . B^T).  KM_ = true: both operands are k-major ("TN":
// C[m][n] = sum_k A[k][m] B[k][n]; row k of A holds records along m) -- the LDS images are then [32 k][BM * 4 bytes] and
// the MFMA fragments are gathered with ds_read_b64_tr_b16 (the hardware transpose read).
template <int BM_, int BN_, int WM_, int WN_, int NSLOT_ = 3, bool KM_ = false>
struct RecCfg {
  static constexpr bool KM = KM_;
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NSLOT = NSLOT_, NT = WM_ * WN_ * 64;
  static constexpr int TM = BM_ / WM_ / 32, TN = BN_ / WN_ / 32;
  static constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128, SLOT = A_BYTES + B_BYTES;
  static constexpr int NWAVE = WM_ * WN_;
  static constexpr int NPA = BM_ / 8 / NWAVE, NPB = BN_ / 8 / NWAVE;      // 1 KB pieces (8 rows) per wave per k-tile
  static constexpr int LDS_BYTES = NSLOT_ * SLOT;
  static_assert(BM_ % (8 * NWAVE) == 0 && BN_ % (8 * NWAVE) == 0, "whole pieces per wave");
};

// Source offset (bytes, relative to the descriptor base) of the 16 bytes lane `lane` fetches for piece `piece` (rows
// 8 piece .. 8 piece + 7 of the tile) under the slot swizzle; rows at or beyond rows_valid fetch from beyond the
// descriptor's range (-> zeros).  pitch = bytes between consecutive rows of the operand.
__device__ __forceinline__ unsigned rec_piece_voff(int piece, int lane, int rows_valid, unsigned pitch) {
  const int row = piece * 8 + (lane >> 3), ps = lane & 7;
  const int ls = ps ^ ((row >> 1) & 7);
  return row < rows_valid ? (unsigned)row * pitch + (unsigned)ls * 16u : 0x80000000u;
}

template <class Cfg>
struct RecOperands {
  u32x4 da, db;                          // descriptors of the A / B row blocks of this workgroup
  unsigned va[Cfg::NPA], vb[Cfg::NPB];   // per-lane source offsets of this wave's pieces
  unsigned b_step;                       // bytes one k-tile advances in B (128 for rows of records, 32 * pitch for k-major)
};

template <class Cfg>
__device__ __forceinline__ void rec_setup(RecOperands<Cfg>& o, const void* A, unsigned a_bytes, unsigned a_pitch, int a_rows,
                                          const void* Bm, unsigned b_bytes, unsigned b_pitch, int b_rows) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  o.da = rec_desc(A, a_bytes);
  o.db = rec_desc(Bm, b_bytes);
  o.b_step = 128u;
#pragma unroll
  for (int j = 0; j < Cfg::NPA; ++j) o.va[j] = rec_piece_voff(wave + Cfg::NWAVE * j, lane, a_rows, a_pitch);
#pragma unroll
  for (int j = 0; j < Cfg::NPB; ++j) o.vb[j] = rec_piece_voff(wave + Cfg::NWAVE * j, lane, b_rows, b_pitch);
}

// k-major operand: the tile image is [32 k][ROWB bytes] (ROWB = 4 * tile width: the records of one k-row), piece p holds
// 1024 / ROWB consecutive k-rows, 16-byte chunk c of k-row r lives at chunk c ^ ((r & 3) << 2) -- the four rows of a
// transposed-read block then sit on four different 64-byte bank groups.  Rows at or beyond K fall outside the
// descriptor (its size is K * pitch minus the column offset of the tile) and read as zeros.
template <int ROWB>
__device__ __forceinline__ unsigned rec_piece_voff_km(int piece, int lane, unsigned pitch) {
  constexpr int RPP = 1024 / ROWB;
  const int row = piece * RPP + (lane * 16) / ROWB, pc = (lane * 16 % ROWB) / 16;
  return (unsigned)row * pitch + (unsigned)(pc ^ ((row & 3) << 2)) * 16u;
}
template <class Cfg>
__device__ __forceinline__ void rec_setup_km(RecOperands<Cfg>& o, const void* A, unsigned a_bytes, unsigned a_pitch, const void* Bm,
                                             unsigned b_bytes, unsigned b_pitch) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  o.da = rec_desc(A, a_bytes);
  o.db = rec_desc(Bm, b_bytes);
  o.b_step = 32u * b_pitch;
#pragma unroll
  for (int j = 0; j < Cfg::NPA; ++j) o.va[j] = rec_piece_voff_km<Cfg::BM * 4>(wave + Cfg::NWAVE * j, lane, a_pitch);
#pragma unroll
  for (int j = 0; j < Cfg::NPB; ++j) o.vb[j] = rec_piece_voff_km<Cfg::BN * 4>(wave + Cfg::NWAVE * j, lane, b_pitch);
}

typedef short rec_s16x4 __attribute__((ext_vector_type(4)));
// 8 consecutive k of this lane's column out of a k-major image: two transposed 4 x 16 blocks, rows k .. k+3 and k+4 .. k+7
__device__ __forceinline__ p16x8 rec_tr_frag(const char* p, int pitch) {
  const rec_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rec_s16x4 __attribute__((address_space(3)))*)(p));
  const rec_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rec_s16x4 __attribute__((address_space(3)))*)(p + 4 * pitch));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(p16x8, v);
}

// Where the A pieces of k-tile t come from.  Plain GEMM: a fixed per-lane offset and the k-tile in the scalar offset.
// (The implicit-GEMM convolution supplies its own: per-tile (source, tap, chunk) from a table, per-lane tap validity.)
template <class Cfg>
struct RecPlainA {
  unsigned va[Cfg::NPA];
  int kt0;
  unsigned step;                         // bytes per k-tile (128, or 32 * pitch for a k-major operand)
  struct Tile { unsigned soff; };
  __device__ __forceinline__ Tile tile(int t) const { return Tile{(unsigned)(kt0 + t) * step}; }
  __device__ __forceinline__ unsigned voff(const Tile&, int j) const { return va[j]; }
};

// Sparse k: the k-tiles a workgroup visits come from a LIST (ascending k-tile numbers, `cnt` of them) instead of a range --
// the volume-backward GEMMs contract over gradient rows of which the step's lookups touched a fraction, and a pre-pass
// (fsraft_corr_bwd_ktiles) lists, per tile of the sparse operand, the k-tiles that can hold anything but zero records.
// Both operands take their k-tile from the list (`kidx`), so A and B stay aligned; skipped tiles would have added +-0.
template <class Cfg>
struct RecListA {
  unsigned va[Cfg::NPA];
  // The list sits in LDS (the kernel copies its slice there first): LDS reads return in order, so the compiler keeps its counted
  // lgkmcnt waits in the k-loop.  (Scalar loads from global memory return out of order -- with one in flight every wait for a
  // fragment read becomes lgkmcnt(0) -- and a vector load would join the in-order vmcnt queue of the DMA ring.)
  const int __attribute__((address_space(3))) * list;
  int cnt;                               // >= 1
  unsigned step;
  struct Tile { unsigned soff; };
  __device__ __forceinline__ unsigned kidx(int t) const { return (unsigned)list[t < cnt ? t : cnt - 1]; }
  __device__ __forceinline__ Tile tile_k(unsigned k) const { return Tile{(unsigned)__builtin_amdgcn_readfirstlane((int)k) * step}; }
  __device__ __forceinline__ Tile tile(int t) const { return tile_k(kidx(t)); }
  __device__ __forceinline__ unsigned voff(const Tile&, int j) const { return va[j]; }
};
constexpr int REC_LIST_MAX = 2048;       // k-tiles per list the kernels hold in LDS
template <class T, class = void>
struct rec_has_kidx : std::false_type {};
template <class T>
struct rec_has_kidx<T, std::void_t<decltype(std::declval<const T&>().kidx(0))>> : std::true_type {};

// acc += A(k-tiles 0 .. KT-1 as described by asrc) . B[rows][kt0 .. kt0+KT)^T
//
// Schedule: three ring slots, two k-tiles in flight, ONE barrier per k-tile placed in the MIDDLE of a tile's MFMAs:
//   read k-half 1 of tile t | 12 MFMAs on k-half 0 (the B pieces of tile t+2 issued in between) | own reads done, own
//   pieces of tile t+1 landed | barrier (tile t+1 complete, nobody reads tile t any more) | read k-half 0 of tile t+1 |
//   12 MFMAs on k-half 1 of tile t (the A pieces of tile t+3 issued in between, into tile t's slot).
// Every group of eight fragment reads has twelve MFMAs (~400 cycles) to land behind, and the DMA pieces -- each costs its
// wave 60-180 issue cycles -- are spread between MFMAs instead of piling up behind the barrier where both waves of a SIMD
// would issue them at the same time.  The loop is unrolled three times so that ring slots, LDS offsets and wait counts
// are compile-time constants: the two waves of a SIMD share its issue slots (~144 non-MFMA instructions per wave and
// k-tile at full matrix-pipe rate), so every scalar modulo / address add taken out of the loop counts.
template <class Cfg, class ASrc = RecPlainA<Cfg>>
__device__ __forceinline__ void rec_mainloop(char* __restrict__ lds, const RecOperands<Cfg>& o, const ASrc& asrc, int kt0, int KT,
                                             f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int NPA = Cfg::NPA, NPB = Cfg::NPB;
  static_assert(Cfg::NSLOT == 3, "the wait counts below are written for two tiles in flight");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const unsigned wbase = (unsigned)(uintptr_t)lds + (unsigned)wave * 1024u;      // this wave's first piece of slot 0

  // A pieces of k-tile t into ring slot SL (t >= KT: a scalar offset beyond the descriptor's range = zero fill)
  auto issue_a = [&](int t, int SL, int j, const typename ASrc::Tile& ts) {
    rec_dma16(asrc.voff(ts, j), o.da, t < KT ? ts.soff : 0x80000000u, wbase + (unsigned)(SL * Cfg::SLOT + Cfg::NWAVE * j * 1024));
  };
  // k-tile number of step t for the B operand: the range kt0 + t, or the ASrc's list entry
  auto kof = [&](int t) -> unsigned {
    if constexpr (rec_has_kidx<ASrc>::value) return asrc.kidx(t);
    else return (unsigned)(kt0 + t);
  };
  auto tile_of = [&](int t, unsigned k) -> typename ASrc::Tile {
    if constexpr (rec_has_kidx<ASrc>::value) return asrc.tile_k(k);
    else return asrc.tile(t);
  };
  auto issue_b = [&](int t, int SL, int j, unsigned k) {
    rec_dma16(o.vb[j], o.db, t < KT ? (unsigned)__builtin_amdgcn_readfirstlane((int)k) * o.b_step : 0x80000000u,
              wbase + (unsigned)(SL * Cfg::SLOT + Cfg::A_BYTES + Cfg::NWAVE * j * 1024));
  };
  auto issue_all = [&](int t, int SL) {
    const unsigned k = kof(t);
    const typename ASrc::Tile ts = tile_of(t, k);
#pragma unroll
    for (int j = 0; j < NPA; ++j) issue_a(t, SL, j, ts);
#pragma unroll
    for (int j = 0; j < NPB; ++j) issue_b(t, SL, j, k);
  };

  struct Frag { p16x8 ah[Cfg::TM], al[Cfg::TM], bh[Cfg::TN], bl[Cfg::TN]; };
  // fragment addressing.  Rows of records: this lane's row of the wave's sub-tile; slot of (k-step s, hi / lo) = 2 s + lh
  // (+ 4); rows 32 apart share (row >> 1) & 7, so the swizzled slot offsets are per-lane constants.  k-major: lane 4 q + p
  // of a 16-lane group supplies k-row q (+ 8 lh + 16 s), columns 16 gb + 4 p .. + 3 of the 32-column block; the chunk
  // swizzle depends on q only, and the lo half of a record is the hi address ^ 64.
  const int ra = wm * (Cfg::TM * 32) + l31, rb = wn * (Cfg::TN * 32) + l31;
  const int sa = (ra >> 1) & 7, sb = (rb >> 1) & 7;
  const int gb = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;
  constexpr int PA = Cfg::BM * 4, PB = Cfg::BN * 4;
  auto km_off = [&](int col) {             // byte offset of columns col .. col+3 (hi) inside k-row (8 lh + q) of an image
    const int byte = (col >> 5) * 128 + (col & 31) * 2;
    return (((byte >> 4) ^ (q << 2)) << 4) + (byte & 15);
  };
  const char* fa = Cfg::KM ? lds + (8 * lh + q) * PA : lds + ra * 128;
  const char* fb = Cfg::KM ? lds + Cfg::A_BYTES + (8 * lh + q) * PB : lds + Cfg::A_BYTES + rb * 128;
  int ka[Cfg::TM], kb[Cfg::TN];
#pragma unroll
  for (int mt = 0; mt < Cfg::TM; ++mt) ka[mt] = km_off(wm * (Cfg::TM * 32) + mt * 32 + 16 * gb + 4 * p4);
#pragma unroll
  for (int nt = 0; nt < Cfg::TN; ++nt) kb[nt] = km_off(wn * (Cfg::TN * 32) + nt * 32 + 16 * gb + 4 * p4);

  auto read_frag = [&](int SL, int s, Frag& f) {
    if constexpr (Cfg::KM) {
#pragma unroll
      for (int mt = 0; mt < Cfg::TM; ++mt) {
        f.ah[mt] = rec_tr_frag(fa + SL * Cfg::SLOT + s * 16 * PA + ka[mt], PA);
        f.al[mt] = rec_tr_frag(fa + SL * Cfg::SLOT + s * 16 * PA + (ka[mt] ^ 64), PA);
      }
#pragma unroll
      for (int nt = 0; nt < Cfg::TN; ++nt) {
        f.bh[nt] = rec_tr_frag(fb + SL * Cfg::SLOT + s * 16 * PB + kb[nt], PB);
        f.bl[nt] = rec_tr_frag(fb + SL * Cfg::SLOT + s * 16 * PB + (kb[nt] ^ 64), PB);
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < Cfg::TM; ++mt) {
        f.ah[mt] = *reinterpret_cast<const p16x8*>(fa + SL * Cfg::SLOT + mt * 32 * 128 + (((2 * s + lh) ^ sa) << 4));
        f.al[mt] = *reinterpret_cast<const p16x8*>(fa + SL * Cfg::SLOT + mt * 32 * 128 + (((4 + 2 * s + lh) ^ sa) << 4));
      }
#pragma unroll
      for (int nt = 0; nt < Cfg::TN; ++nt) {
        f.bh[nt] = *reinterpret_cast<const p16x8*>(fb + SL * Cfg::SLOT + nt * 32 * 128 + (((2 * s + lh) ^ sb) << 4));
        f.bl[nt] = *reinterpret_cast<const p16x8*>(fb + SL * Cfg::SLOT + nt * 32 * 128 + (((4 + 2 * s + lh) ^ sb) << 4));
      }
    }
  };
  // the 3 TM TN MFMAs of one k-half with NPIECE DMA pieces spaced evenly between the accumulator groups
  auto mfmas = [&](const Frag& f, auto&& piece, int npiece) {
    constexpr int NM = Cfg::TM * Cfg::TN;
#pragma unroll
    for (int mt = 0; mt < Cfg::TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < Cfg::TN; ++nt) {
        acc[mt][nt] = fs_mfma_32x32x16(f.al[mt], f.bh[nt], acc[mt][nt]);
        acc[mt][nt] = fs_mfma_32x32x16(f.ah[mt], f.bl[nt], acc[mt][nt]);
        acc[mt][nt] = fs_mfma_32x32x16(f.ah[mt], f.bh[nt], acc[mt][nt]);
        const int i = mt * Cfg::TN + nt;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < npiece && (k * NM) / npiece == i) {
            __builtin_amdgcn_sched_barrier(0);
            piece(k);
            __builtin_amdgcn_sched_barrier(0);
          }
      }
  };

  if (KT <= 0) return;
  issue_all(0, 0);
  issue_all(1, 1);
  unsigned kc = kof(2);                          // k-tile number of tile t + 2 at the top of step t
  {                                              // tile 2: A pieces now, B pieces inside the first MFMA group (as in steady state)
    const typename ASrc::Tile ts = tile_of(2, kc);
#pragma unroll
    for (int j = 0; j < NPA; ++j) issue_a(2, 2, j, ts);
  }
  rec_wait_vm<NPA + NPA + NPB>();                // tile 0 landed (tile 1 and the A pieces of tile 2 may be in flight)
  __builtin_amdgcn_s_barrier();
  Frag f0, f1;
  read_frag(0, 0, f0);

  // one k-tile: tile t lives in slot SL, tile t+1 in slot (SL + 1) % 3
  auto step = [&](int t, auto SLc) {
    constexpr int SL = decltype(SLc)::value, SN = (SL + 1) % 3, SP = (SL + 2) % 3;
    const unsigned k3 = kof(t + 3);              // (list variant: an LDS read with the first MFMA group to land behind)
    read_frag(SL, 1, f1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f0, [&](int k) { issue_b(t + 2, SP, k, kc); }, NPB);        // B pieces of tile t+2 (its A pieces went last step)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    rec_wait_vm<NPA + NPB>();                    // outstanding: tile t+1, tile t+2 -> leave tile t+2 in flight
    __builtin_amdgcn_s_barrier();
    read_frag(SN, 0, f0);
    __builtin_amdgcn_sched_barrier(0);
    const typename ASrc::Tile ts = tile_of(t + 3, k3);
    mfmas(f1, [&](int k) { issue_a(t + 3, SL, k, ts); }, NPA);        // A pieces of tile t+3 into the slot tile t leaves
    __builtin_amdgcn_sched_barrier(0);
    kc = k3;
  };
  for (int t = 0; t < KT; t += 3) {
    step(t, std::integral_constant<int, 0>{});
    if (t + 1 >= KT) break;
    step(t + 1, std::integral_constant<int, 1>{});
    if (t + 2 >= KT) break;
    step(t + 2, std::integral_constant<int, 2>{});
  }
  rec_wait_vm<0>();                              // zero-fill pieces issued past the end
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// ---- the same pipeline on v_mfma_f32_16x16x32_bf16 ------------------------------------------------------------------
// Same LDS images, same DMA ring and barrier placement; a k-tile (one record, 32 k) is ONE k-step of the 16x16x32 shape
// instead of two of the 32x32x16 shape.  Cycles per FLOP are equal; the chip holds a higher clock on this shape under a
// power-limited load (MI355X_MICROARCH.md, DVFS give-back item 7), which is where these kernels sit.
// Row-major records only.  Fragment of a 16-row block: lane = (row & 15) + 16 g fetches k-group g (8 k) = slot g (hi) and
// slot 4 + g (lo) of its row's record, under the same slot swizzle ((row >> 1) & 7 is a per-lane constant: blocks are 16
// rows apart).  A k-tile's 48 TM TN MFMAs run as two phases of 24: row blocks of mt = 0 first, of mt = 1 second, so the
// schedule of rec_mainloop carries over: [read A(mt=1) of tile t | phase 0 + B pieces of t+2 | waits, barrier | read A(mt=0)
// and all of B of tile t+1 (B double-buffered in registers by tile parity) | phase 1 + A pieces of t+3].
// Accumulators: acc[mt][nt] holds four 16x16 blocks, block (mi, ni) in registers 4 (2 mi + ni) .. + 3
// (row = 16 mi + 4 (lane >> 4) + r, col = 16 ni + (lane & 15)): rec16_row / rec16_col.
typedef float rec_f32x4 __attribute__((ext_vector_type(4)));

template <class Cfg, class ASrc = RecPlainA<Cfg>>
__device__ __forceinline__ void rec_mainloop16(char* __restrict__ lds, const RecOperands<Cfg>& o, const ASrc& asrc, int kt0, int KT,
                                               f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int NPA = Cfg::NPA, NPB = Cfg::NPB, TM = Cfg::TM, TN = Cfg::TN;
  static_assert(Cfg::NSLOT == 3 && !Cfg::KM && TM == 2, "written for the 64-row wave tile, three slots, rows of records");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int l15 = lane & 15, g4 = lane >> 4;
  const unsigned wbase = (unsigned)(uintptr_t)lds + (unsigned)wave * 1024u;
  auto issue_a = [&](int t, int SL, int j, const typename ASrc::Tile& ts) {
    rec_dma16(asrc.voff(ts, j), o.da, t < KT ? ts.soff : 0x80000000u, wbase + (unsigned)(SL * Cfg::SLOT + Cfg::NWAVE * j * 1024));
  };
  auto issue_b = [&](int t, int SL, int j) {
    rec_dma16(o.vb[j], o.db, t < KT ? (unsigned)(kt0 + t) * o.b_step : 0x80000000u,
              wbase + (unsigned)(SL * Cfg::SLOT + Cfg::A_BYTES + Cfg::NWAVE * j * 1024));
  };
  auto issue_all = [&](int t, int SL) {
    const typename ASrc::Tile ts = asrc.tile(t);
#pragma unroll
    for (int j = 0; j < NPA; ++j) issue_a(t, SL, j, ts);
#pragma unroll
    for (int j = 0; j < NPB; ++j) issue_b(t, SL, j);
  };
  const int ra = wm * (TM * 32) + l15, rb = wn * (TN * 32) + l15;
  const int sa = (ra >> 1) & 7, sb = (rb >> 1) & 7;
  const char* fa = lds + ra * 128;
  const char* fb = lds + Cfg::A_BYTES + rb * 128;
  const int a_hi = (g4 ^ sa) << 4, a_lo = ((4 + g4) ^ sa) << 4, b_hi = (g4 ^ sb) << 4, b_lo = ((4 + g4) ^ sb) << 4;

  struct FragA { p16x8 h[2], l[2]; };                  // the two 16-row blocks of one mt
  struct FragB { p16x8 h[2 * TN], l[2 * TN]; };        // all 16-column blocks of the wave's columns
  auto read_a = [&](int SL, int mt, FragA& f) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      f.h[mi] = *reinterpret_cast<const p16x8*>(fa + SL * Cfg::SLOT + (mt * 32 + mi * 16) * 128 + a_hi);
      f.l[mi] = *reinterpret_cast<const p16x8*>(fa + SL * Cfg::SLOT + (mt * 32 + mi * 16) * 128 + a_lo);
    }
  };
  auto read_b = [&](int SL, FragB& f) {
#pragma unroll
    for (int nb = 0; nb < 2 * TN; ++nb) {
      f.h[nb] = *reinterpret_cast<const p16x8*>(fb + SL * Cfg::SLOT + nb * 16 * 128 + b_hi);
      f.l[nb] = *reinterpret_cast<const p16x8*>(fb + SL * Cfg::SLOT + nb * 16 * 128 + b_lo);
    }
  };
  rec_f32x4 c[TM][2][2 * TN];
#pragma unroll
  for (int mt = 0; mt < TM; ++mt)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nb = 0; nb < 2 * TN; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[mt][mi][nb][r] = acc[mt][nb >> 1][4 * (2 * mi + (nb & 1)) + r];
  // the 12 TN MFMAs of one phase (row blocks of `mt`) with NPIECE DMA pieces spaced between the column blocks
  auto mfmas = [&](int mt, const FragA& a, const FragB& b, auto&& piece, int npiece) {
    constexpr int NG = 2 * TN;
#pragma unroll
    for (int nb = 0; nb < NG; ++nb) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        c[mt][mi][nb] = fs_mfma_16x16x32(a.l[mi], b.h[nb], c[mt][mi][nb]);
        c[mt][mi][nb] = fs_mfma_16x16x32(a.h[mi], b.l[nb], c[mt][mi][nb]);
        c[mt][mi][nb] = fs_mfma_16x16x32(a.h[mi], b.h[nb], c[mt][mi][nb]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < npiece && (k * NG) / npiece == nb) {
          __builtin_amdgcn_sched_barrier(0);
          piece(k);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };

  if (KT <= 0) return;
  issue_all(0, 0);
  issue_all(1, 1);
  {
    const typename ASrc::Tile ts = asrc.tile(2);
#pragma unroll
    for (int j = 0; j < NPA; ++j) issue_a(2, 2, j, ts);
  }
  rec_wait_vm<NPA + NPA + NPB>();
  __builtin_amdgcn_s_barrier();
  FragA a0, a1;
  FragB b0, b1;
  read_a(0, 0, a0);
  read_b(0, b0);

  auto step = [&](int t, auto SLc, FragB& bc, FragB& bn) {
    constexpr int SL = decltype(SLc)::value, SN = (SL + 1) % 3, SP = (SL + 2) % 3;
    read_a(SL, 1, a1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(0, a0, bc, [&](int k) { issue_b(t + 2, SP, k); }, NPB);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    rec_wait_vm<NPA + NPB>();
    __builtin_amdgcn_s_barrier();
    read_a(SN, 0, a0);
    read_b(SN, bn);
    __builtin_amdgcn_sched_barrier(0);
    const typename ASrc::Tile ts = asrc.tile(t + 3);
    mfmas(1, a1, bc, [&](int k) { issue_a(t + 3, SL, k, ts); }, NPA);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < KT; t += 6) {                    // ring slot period 3 x register-buffer period 2
    step(t, std::integral_constant<int, 0>{}, b0, b1);
    if (t + 1 >= KT) break;
    step(t + 1, std::integral_constant<int, 1>{}, b1, b0);
    if (t + 2 >= KT) break;
    step(t + 2, std::integral_constant<int, 2>{}, b0, b1);
    if (t + 3 >= KT) break;
    step(t + 3, std::integral_constant<int, 0>{}, b1, b0);
    if (t + 4 >= KT) break;
    step(t + 4, std::integral_constant<int, 1>{}, b0, b1);
    if (t + 5 >= KT) break;
    step(t + 5, std::integral_constant<int, 2>{}, b1, b0);
  }
  rec_wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int mt = 0; mt < TM; ++mt)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nb = 0; nb < 2 * TN; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][nb >> 1][4 * (2 * mi + (nb & 1)) + r] = c[mt][mi][nb][r];
}
// accumulator tile (mt, nt), register r of lane -> (row, col) inside the BM x BN tile, 16x16x32 layout
template <class Cfg>
__device__ __forceinline__ int rec16_row(int mt, int r) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave / Cfg::WN) * (Cfg::TM * 32) + mt * 32 + 16 * ((r >> 2) >> 1) + 4 * (lane >> 4) + (r & 3);
}
template <class Cfg>
__device__ __forceinline__ int rec16_col(int nt, int r) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave % Cfg::WN) * (Cfg::TN * 32) + nt * 32 + 16 * ((r >> 2) & 1) + (lane & 15);
}

// accumulator tile (mt, nt), register r of lane -> (row, col) inside the BM x BN tile
template <class Cfg>
__device__ __forceinline__ int rec_row(int mt, int r) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave / Cfg::WN) * (Cfg::TM * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
template <class Cfg>
__device__ __forceinline__ int rec_col(int nt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave % Cfg::WN) * (Cfg::TN * 32) + nt * 32 + (lane & 31);
}

// fp32 x4 -> hi / lo bf16 x4 (same rounding as gemm_core_split.hpp::split4)
__device__ __forceinline__ void rec_split4(const float* r, uint2& hi, uint2& lo, float s = 1.0f) { fs_split4(r, s, hi, lo); }
