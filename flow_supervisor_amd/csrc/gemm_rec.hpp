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
//     wait(own pieces of tile t landed) ; barrier (=> tile t complete, slot of tile t-1 free) ; issue tile t+NSLOT-1 ;
//     fragment reads + 3 x MFMA on tile t.
#pragma once
#include "common.hpp"

typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
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
template <int BM_, int BN_, int WM_, int WN_, int NSLOT_ = 3>
struct RecCfg {
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
};

template <class Cfg>
__device__ __forceinline__ void rec_setup(RecOperands<Cfg>& o, const void* A, unsigned a_bytes, unsigned a_pitch, int a_rows,
                                          const void* Bm, unsigned b_bytes, unsigned b_pitch, int b_rows) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  o.da = rec_desc(A, a_bytes);
  o.db = rec_desc(Bm, b_bytes);
#pragma unroll
  for (int j = 0; j < Cfg::NPA; ++j) o.va[j] = rec_piece_voff(wave + Cfg::NWAVE * j, lane, a_rows, a_pitch);
#pragma unroll
  for (int j = 0; j < Cfg::NPB; ++j) o.vb[j] = rec_piece_voff(wave + Cfg::NWAVE * j, lane, b_rows, b_pitch);
}

// acc += A[rows][kt0 .. kt0+KT) . B[rows][same]^T over KT k-tiles (records) of both operands.
template <class Cfg>
__device__ __forceinline__ void rec_mainloop(char* __restrict__ lds, const RecOperands<Cfg>& o, int kt0, int KT,
                                             f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int NP = Cfg::NPA + Cfg::NPB;
  static_assert(Cfg::NSLOT == 3, "the wait counts below are written for two tiles in flight");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;

  auto issue = [&](int t) {                      // k-tile t (relative) -> ring slot t % 3; t >= KT: nothing to fetch (the
    const unsigned slot = lds0 + (unsigned)(t % 3) * Cfg::SLOT;   // descriptor range check turns the pieces into zero fills)
    const unsigned soff = t < KT ? (unsigned)(kt0 + t) * 128u : 0x80000000u;
#pragma unroll
    for (int j = 0; j < Cfg::NPA; ++j) rec_dma16(o.va[j], o.da, soff, slot + (unsigned)(wave + Cfg::NWAVE * j) * 1024u);
#pragma unroll
    for (int j = 0; j < Cfg::NPB; ++j)
      rec_dma16(o.vb[j], o.db, soff, slot + Cfg::A_BYTES + (unsigned)(wave + Cfg::NWAVE * j) * 1024u);
  };

  // fragment addressing: this lane's row of the wave's sub-tile; slot of (k-step s, hi / lo) = 2 s + lh (+ 4); rows 32 apart
  // share (row >> 1) & 7, so the swizzled slot offsets are per-lane constants
  const int ra = wm * (Cfg::TM * 32) + l31, rb = wn * (Cfg::TN * 32) + l31;
  const int sa = (ra >> 1) & 7, sb = (rb >> 1) & 7;

  if (KT <= 0) return;
  issue(0);
  issue(1);
  for (int t = 0; t < KT; ++t) {
    rec_wait_vm<NP>();                           // all but the youngest tile's pieces of this wave have landed
    __builtin_amdgcn_s_barrier();                // ... of every wave: tile t complete, and everyone is done reading tile t - 1
    issue(t + 2);                                // into the slot tile t - 1 occupied
    const char* cur = lds + (t % 3) * Cfg::SLOT;
    const char* As = cur + ra * 128;
    const char* Bs = cur + Cfg::A_BYTES + rb * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8r ah[Cfg::TM], al[Cfg::TM], bh[Cfg::TN], bl[Cfg::TN];
#pragma unroll
      for (int mt = 0; mt < Cfg::TM; ++mt) {
        ah[mt] = *reinterpret_cast<const bf16x8r*>(As + mt * 32 * 128 + (((2 * s + lh) ^ sa) << 4));
        al[mt] = *reinterpret_cast<const bf16x8r*>(As + mt * 32 * 128 + (((4 + 2 * s + lh) ^ sa) << 4));
      }
#pragma unroll
      for (int nt = 0; nt < Cfg::TN; ++nt) {
        bh[nt] = *reinterpret_cast<const bf16x8r*>(Bs + nt * 32 * 128 + (((2 * s + lh) ^ sb) << 4));
        bl[nt] = *reinterpret_cast<const bf16x8r*>(Bs + nt * 32 * 128 + (((4 + 2 * s + lh) ^ sb) << 4));
      }
#pragma unroll
      for (int mt = 0; mt < Cfg::TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::TN; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
    }
  }
  rec_wait_vm<0>();                              // the two zero-fill tiles issued past the end
  __builtin_amdgcn_s_barrier();
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
__device__ __forceinline__ void rec_split4(const float* r, uint2& hi, uint2& lo) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  unsigned h[2], l[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f2 x = {r[2 * i], r[2 * i + 1]};
    h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf2));
    const float h0 = __builtin_bit_cast(float, h[i] << 16), h1 = __builtin_bit_cast(float, h[i] & 0xffff0000u);
    const f2 d = {x[0] - h0, x[1] - h1};
    l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf2));
  }
  hi = make_uint2(h[0], h[1]);
  lo = make_uint2(l[0], l[1]);
}
