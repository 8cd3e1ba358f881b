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
// Loader concept (duck-typed): the staging of one k-tile is cut into NCH chunks (typically one
// 16-byte load each) so the mainloop can interleave them with MFMAs:
//   struct L { static constexpr int NREG, NCH;
//              __device__ void fetch_chunk(int kt, float (&r)[NREG], int c) const;   // global -> VGPR
//              __device__ void store_chunk(float* lds_tile, const float (&r)[NREG], int c) const; };
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
//
// Schedule of one k-tile (measured with scripts/probe/mfma_probe.hip, 512 workgroups, 2 per CU):
// global loads of the next tile first (branch-free, nothing consumes them yet), then G = BK/2 MFMA
// groups whose LDS fragment reads run one group ahead (+6 % over reading right before use), then
// the LDS stores of the staged tile and one barrier.  Spreading the loads/stores between the MFMA
// groups, or prefetching two tiles deep, measured no better (the load latency is already hidden;
// the remaining gap to the bare-MFMA rate is issue overhead at 2 waves per SIMD).
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop(float* __restrict__ lds, int KT, const LA& la, const LB& lb,
                                              f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int l31 = lane & 31, lh = lane >> 5;
  constexpr int G = Cfg::BK / 2;

  float ra[LA::NREG];
  float rb[LB::NREG];
  // validity of each staged chunk (zero-select is applied when the chunk is written to LDS, so
  // nothing consumes a load result right behind the load)
  bool oka[LA::NCH], okb[LB::NCH];
  if (KT > 0) {
#pragma unroll
    for (int c = 0; c < LA::NCH; ++c) oka[c] = la.fetch_chunk(0, ra, c);
#pragma unroll
    for (int c = 0; c < LB::NCH; ++c) okb[c] = lb.fetch_chunk(0, rb, c);
#pragma unroll
    for (int c = 0; c < LA::NCH; ++c) la.store_chunk(lds, ra, c, oka[c]);
#pragma unroll
    for (int c = 0; c < LB::NCH; ++c) lb.store_chunk(lds + Cfg::A_TILE, rb, c, okb[c]);
  }
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    float* cur = lds + (kt & 1) * Cfg::STAGE;
    float* nxt = lds + ((kt + 1) & 1) * Cfg::STAGE;
    const int ktn = (kt + 1 < KT) ? kt + 1 : kt;   // last iteration re-stages its own tile: keeps the body branch-free
#pragma unroll
    for (int c = 0; c < LA::NCH; ++c) oka[c] = la.fetch_chunk(ktn, ra, c);
#pragma unroll
    for (int c = 0; c < LB::NCH; ++c) okb[c] = lb.fetch_chunk(ktn, rb, c);
    const float* As = cur + lh * Cfg::LDA + wm * (Cfg::TM * 32) + l31;
    const float* Bs = cur + Cfg::A_TILE + lh * Cfg::LDB + wn * (Cfg::TN * 32) + l31;
    float a[2][Cfg::TM], b[2][Cfg::TN];
#pragma unroll
    for (int mt = 0; mt < Cfg::TM; ++mt) a[0][mt] = As[mt * 32];
#pragma unroll
    for (int nt = 0; nt < Cfg::TN; ++nt) b[0][nt] = Bs[nt * 32];
#pragma unroll
    for (int ks = 0; ks < G; ++ks) {
      const int c = ks & 1, n = c ^ 1;
      if (ks + 1 < G) {
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
#pragma unroll
    for (int c = 0; c < LA::NCH; ++c) la.store_chunk(nxt, ra, c, oka[c]);
#pragma unroll
    for (int c = 0; c < LB::NCH; ++c) lb.store_chunk(nxt + Cfg::A_TILE, rb, c, okb[c]);
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
  static constexpr int NCH = NF4;
  static_assert((ROWS * F4_PER_ROW) % 256 == 0, "tile must divide over 256 threads");
  const float* base;   // already offset to the tile's first row / first k
  int64_t ld;
  int rows_valid;      // rows >= rows_valid read as zero
  int k_valid_total;   // k >= k_valid_total reads as zero
  bool vec = true;     // false: ld or k_valid_total not a multiple of 4 -> guarded scalar loads
  // Loads are unconditional (address clamped into the tile, result zeroed by a select): a
  // branch around a load makes hipcc wait vmcnt(0) at the join and serialises the prefetch.
  __device__ __forceinline__ bool fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int row = e / F4_PER_ROW, kq = e % F4_PER_ROW;
    const int k = kt * BK + kq * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    bool ok = true;
    if (vec) {
      ok = row < rows_valid && k < k_valid_total;
      v = gload4(base + (ok ? (int64_t)row * ld + k : 0));
    } else if (row < rows_valid) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (k + c < k_valid_total) v[c] = base[(int64_t)row * ld + k + c];
    }
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
    return ok;
  }
  __device__ __forceinline__ void store_chunk(float* t, const float (&r)[NREG], int j, bool ok) const {
    const int e = threadIdx.x + 256 * j;
    const int row = e / F4_PER_ROW, kq = e % F4_PER_ROW;
#pragma unroll
    for (int c = 0; c < 4; ++c) t[(kq * 4 + c) * LD + row] = ok ? r[4 * j + c] : 0.f;
  }
};

// Source is row(k)-major with the tile dimension contiguous: element (k, col) at
// base[k*ld + col]  (LDS image is the same orientation: straight float4 copies).
template <int COLS, int BK, int LD>
struct KMajorTileLoader {
  static constexpr int F4_PER_K = COLS / 4;
  static constexpr int NF4 = BK * F4_PER_K / 256;
  static constexpr int NREG = NF4 * 4;
  static constexpr int NCH = NF4;
  static_assert((BK * F4_PER_K) % 256 == 0, "tile must divide over 256 threads");
  static_assert(LD % 4 == 0, "LDS row pitch must keep float4 alignment");
  const float* base;
  int64_t ld;
  int cols_valid;
  int k_valid_total;
  bool vec = true;     // false: ld or cols_valid not a multiple of 4 -> guarded scalar loads
  __device__ __forceinline__ bool fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int k = e / F4_PER_K, c4 = e % F4_PER_K;
    const int kk = kt * BK + k;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    bool ok = true;
    if (vec) {
      ok = kk < k_valid_total && c4 * 4 < cols_valid;
      v = gload4(base + (ok ? (int64_t)kk * ld + c4 * 4 : 0));
    } else if (kk < k_valid_total) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c4 * 4 + c < cols_valid) v[c] = base[(int64_t)kk * ld + c4 * 4 + c];
    }
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
    return ok;
  }
  __device__ __forceinline__ void store_chunk(float* t, const float (&r)[NREG], int j, bool ok) const {
    const int e = threadIdx.x + 256 * j;
    const int k = e / F4_PER_K, c4 = e % F4_PER_K;
    f32x4 v = {r[4 * j + 0], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]};
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(t + k * LD + c4 * 4) = ok ? v : z;
  }
};
