// fp32 MFMA block-GEMM core for gfx950.
//
// One workgroup = 256 threads = 4 wavefronts (64 lanes each) arranged WM x WN.
// Each wave owns a (TM*32) x (TN*32) sub-tile built from v_mfma_f32_32x32x2_f32
// (exact fp32: a k-ordered fmaf chain, see cdna_hip_programming.md section 3).
//
// LDS images are always k-major:  As[k][m], Bs[k][n].  The MFMA operand maps are
//   A: lane l holds A[i = l&31][k = l>>5]     B: lane l holds B[k = l>>5][j = l&31]
// so a fragment read is one ds_read_b32 per lane with the 32 lanes of a half-wave
// on 32 consecutive dwords (conflict free) and the two half-waves on rows k, k+1.
// fp32 MFMA issues one instruction per 64 cycles per SIMD, so LDS bandwidth is
// never the limiter here; what matters is keeping global loads in flight, which
// the register-staged double buffer below does (fetch tile t+1 -> VGPRs, MFMA on
// tile t from LDS, then write the VGPRs to the other LDS buffer, one barrier).
//
// Loader concept (duck-typed):
//   struct L { static constexpr int NREG; __device__ void fetch(int kt, float (&r)[NREG]) const;
//              __device__ void store(float* lds_tile, const float (&r)[NREG]) const; };
// `lds_tile` is the [BK][LD] image for that operand.
#pragma once
#include "common.hpp"

template <int BM_, int BN_, int BK_, int WM_, int WN_, int PADA_, int PADB_>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
  static constexpr int LDA = BM_ + PADA_, LDB = BN_ + PADB_;
  static constexpr int TM = BM_ / WM_ / 32, TN = BN_ / WN_ / 32;
  static constexpr int A_TILE = BK_ * LDA, B_TILE = BK_ * LDB;
  static constexpr int STAGE = A_TILE + B_TILE;          // floats per stage
  static constexpr int LDS_FLOATS = 2 * STAGE;
  static_assert(WM_ * WN_ == 4, "4 waves per workgroup");
  static_assert(BK_ % 2 == 0, "BK must be even");
};

// acc[mt][nt] += A_tile * B_tile over KT k-tiles.
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop(float* __restrict__ lds, int KT, const LA& la, const LB& lb,
                                              f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int l31 = lane & 31, lh = lane >> 5;

  float ra[LA::NREG];
  float rb[LB::NREG];
  if (KT > 0) {
    la.fetch(0, ra);
    lb.fetch(0, rb);
    la.store(lds, ra);
    lb.store(lds + Cfg::A_TILE, rb);
  }
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    float* cur = lds + (kt & 1) * Cfg::STAGE;
    float* nxt = lds + ((kt + 1) & 1) * Cfg::STAGE;
    const bool more = (kt + 1) < KT;
    if (more) {
      la.fetch(kt + 1, ra);
      lb.fetch(kt + 1, rb);
    }
    const float* As = cur + lh * Cfg::LDA + wm * (Cfg::TM * 32) + l31;
    const float* Bs = cur + Cfg::A_TILE + lh * Cfg::LDB + wn * (Cfg::TN * 32) + l31;
    // fragment reads run one k-pair ahead of the MFMAs that consume them, so the LDS latency
    // of step ks+1 hides under the 4 x 64-cycle MFMAs of step ks
    float a[2][Cfg::TM], b[2][Cfg::TN];
#pragma unroll
    for (int mt = 0; mt < Cfg::TM; ++mt) a[0][mt] = As[mt * 32];
#pragma unroll
    for (int nt = 0; nt < Cfg::TN; ++nt) b[0][nt] = Bs[nt * 32];
#pragma unroll
    for (int ks = 0; ks < Cfg::BK / 2; ++ks) {
      const int c = ks & 1, n = c ^ 1;
      if (ks + 1 < Cfg::BK / 2) {
#pragma unroll
        for (int mt = 0; mt < Cfg::TM; ++mt) a[n][mt] = As[(2 * ks + 2) * Cfg::LDA + mt * 32];
#pragma unroll
        for (int nt = 0; nt < Cfg::TN; ++nt) b[n][nt] = Bs[(2 * ks + 2) * Cfg::LDB + nt * 32];
      }
      // pin the order: hipcc otherwise sinks the prefetch reads back next to their use and
      // waits lgkmcnt(0) in front of every MFMA group
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < Cfg::TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][mt], b[c][nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
      la.store(nxt, ra);
      lb.store(nxt + Cfg::A_TILE, rb);
    }
    __syncthreads();
  }
}

// Accumulator element (mt, nt, reg r) of this lane sits at
//   row = wm*TM*32 + mt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),  col = wn*TN*32 + nt*32 + (lane&31).
template <class Cfg>
__device__ __forceinline__ int acc_row(int mt, int r) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave / Cfg::WN) * (Cfg::TM * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
template <class Cfg>
__device__ __forceinline__ int acc_col(int nt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave % Cfg::WN) * (Cfg::TN * 32) + nt * 32 + (lane & 31);
}

// ---- generic tile loaders ---------------------------------------------------
// Source is k-contiguous: element (row, k) at base[row*ld + k].  Each thread
// moves float4s along k and transposes on the LDS write (As[k][row]).
// ROWS x BK tile, 256 threads.  Requires ld % 4 == 0 and 16-byte aligned base.
template <int ROWS, int BK, int LD>
struct RowMajorTileLoader {
  static constexpr int F4_PER_ROW = BK / 4;
  static constexpr int NF4 = ROWS * F4_PER_ROW / 256;
  static constexpr int NREG = NF4 * 4;
  static_assert((ROWS * F4_PER_ROW) % 256 == 0, "tile must divide over 256 threads");
  const float* base;   // already offset to the tile's first row / first k
  int64_t ld;
  int rows_valid;      // rows >= rows_valid read as zero
  int k_valid_total;   // k >= k_valid_total reads as zero
  bool vec = true;     // false: ld or k_valid_total not a multiple of 4 -> guarded scalar loads
  __device__ __forceinline__ void fetch(int kt, float (&r)[NREG]) const {
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      const int e = threadIdx.x + 256 * j;
      const int row = e / F4_PER_ROW, kq = e % F4_PER_ROW;
      const int k = kt * BK + kq * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        if (row < rows_valid && k < k_valid_total) v = *reinterpret_cast<const f32x4*>(base + (int64_t)row * ld + k);
      } else if (row < rows_valid) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (k + c < k_valid_total) v[c] = base[(int64_t)row * ld + k + c];
      }
      r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
    }
  }
  __device__ __forceinline__ void store(float* t, const float (&r)[NREG]) const {
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      const int e = threadIdx.x + 256 * j;
      const int row = e / F4_PER_ROW, kq = e % F4_PER_ROW;
#pragma unroll
      for (int c = 0; c < 4; ++c) t[(kq * 4 + c) * LD + row] = r[4 * j + c];
    }
  }
};

// Source is row(k)-major with the tile dimension contiguous: element (k, col) at
// base[k*ld + col]  (LDS image is the same orientation: straight float4 copies).
template <int COLS, int BK, int LD>
struct KMajorTileLoader {
  static constexpr int F4_PER_K = COLS / 4;
  static constexpr int NF4 = BK * F4_PER_K / 256;
  static constexpr int NREG = NF4 * 4;
  static_assert((BK * F4_PER_K) % 256 == 0, "tile must divide over 256 threads");
  static_assert(LD % 4 == 0, "LDS row pitch must keep float4 alignment");
  const float* base;
  int64_t ld;
  int cols_valid;
  int k_valid_total;
  bool vec = true;     // false: ld or cols_valid not a multiple of 4 -> guarded scalar loads
  __device__ __forceinline__ void fetch(int kt, float (&r)[NREG]) const {
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      const int e = threadIdx.x + 256 * j;
      const int k = e / F4_PER_K, c4 = e % F4_PER_K;
      const int kk = kt * BK + k;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        if (kk < k_valid_total && c4 * 4 < cols_valid) v = *reinterpret_cast<const f32x4*>(base + (int64_t)kk * ld + c4 * 4);
      } else if (kk < k_valid_total) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c4 * 4 + c < cols_valid) v[c] = base[(int64_t)kk * ld + c4 * 4 + c];
      }
      r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
    }
  }
  __device__ __forceinline__ void store(float* t, const float (&r)[NREG]) const {
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      const int e = threadIdx.x + 256 * j;
      const int k = e / F4_PER_K, c4 = e % F4_PER_K;
      f32x4 v = {r[4 * j + 0], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]};
      *reinterpret_cast<f32x4*>(t + k * LD + c4 * 4) = v;
    }
  }
};
