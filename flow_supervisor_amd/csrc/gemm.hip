// Batched fp32 GEMMs on the MFMA core, used by the backward of the all-pairs volume
// (row a1 backward, K5 in SURVEY.md 2a; reference: autograd of torch.matmul in
// pytorch/core/corr.py:52-60):
//   NT:  C[b][m][n] = alpha * sum_k A[b][m][k] * Bm[b][n][k]     (dF1 = s * f2 . dV^T  laid out [c][i])
//   NN:  C[b][m][n] = alpha * sum_k A[b][m][k] * Bm[b][k][n]     (dF2 = s * f1 . dV    laid out [c][j])
// With A = the NCHW feature map ([C][N] per sample) both results come out directly in
// NCHW, so no transposes are needed on either side.
#include "gemm_core.hpp"
#include "gemm_core_split.hpp"

namespace {

using CfgNT = GemmCfg<128, 128, 32, 2, 2, 2, 2>;
using CfgNN = GemmCfg<128, 128, 32, 2, 2, 2, 0>;

struct GemmArgs {
  const float* A; int64_t lda, sA;
  const float* Bm; int64_t ldb, sB;
  float* C; int64_t ldc, sC;
  int M, N, K; float alpha; int accumulate;
  const unsigned* a_amax; const unsigned* b_amax;     // split kernels: amax words of A and B (NULL: scale 1)
};

template <class Cfg, bool BT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  const int n0 = blockIdx.x * Cfg::BN, m0 = blockIdx.y * Cfg::BM, b = blockIdx.z;
  const float* A = g.A + b * g.sA + (int64_t)m0 * g.lda;
  RowMajorTileLoader<Cfg::BM, Cfg::BK, Cfg::LDA> la{A, g.lda, g.M - m0, g.K, (g.lda % 4 == 0) && (g.K % 4 == 0)};
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int KT = (g.K + Cfg::BK - 1) / Cfg::BK;
  if constexpr (BT) {
    RowMajorTileLoader<Cfg::BN, Cfg::BK, Cfg::LDB> lb{g.Bm + b * g.sB + (int64_t)n0 * g.ldb, g.ldb, g.N - n0, g.K,
                                                      (g.ldb % 4 == 0) && (g.K % 4 == 0)};
    gemm_mainloop<Cfg>(lds, KT, la, lb, acc);
  } else {
    KMajorTileLoader<Cfg::BN, Cfg::BK, Cfg::LDB> lb{g.Bm + b * g.sB + n0, g.ldb, g.N - n0, g.K,
                                                    (g.ldb % 4 == 0) && (g.N % 4 == 0)};
    gemm_mainloop<Cfg>(lds, KT, la, lb, acc);
  }
  float* C = g.C + b * g.sC;
#pragma unroll
  for (int nt = 0; nt < Cfg::TN; ++nt) {
    const int n = n0 + acc_col<Cfg>(nt);
    if (n >= g.N) continue;
#pragma unroll
    for (int mt = 0; mt < Cfg::TM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + acc_row<Cfg>(mt, r);
        if (m >= g.M) continue;
        float* o = C + (int64_t)m * g.ldc + n;
        const float v = g.alpha * acc[mt][nt][r];
        *o = g.accumulate ? *o + v : v;
      }
  }
}

// ---- split-bf16 variants (same results to ~2^-17 per product, see gemm_core_split.hpp) ----------
using SNT = SplitCfg<128, 128, 2, 2>;          // NT: A [M][K], B [N][K], both k-contiguous
using STN = SplitTnCfg<128, 128, 2, 2, 1>;     // TN: A [K][M], B [K][N], both k-major (transposed LDS reads)

template <class Cfg, int NCH_>
struct SplitRowLoader {          // chunk e: row e / 8, k = kt*32 + 4*(e % 8)
  static constexpr int NCH = NCH_, NREG = NCH_ * 4;
  const float* base; int64_t ld; int rows_valid, K; float s;
  __device__ __forceinline__ void fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int row = e >> 3, k = kt * 32 + 4 * (e & 7);
    const bool ok = row < rows_valid && k < K && kt >= 0;          // kt < 0: zero tile (split_mainloop's padding)
    const f32x4 v = gload4(ok ? base + row * ld + k : g_fsraft_zero16);
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
  }
  __device__ __forceinline__ void stage_chunk(char* tile, const float* r4, int j) const {
    stage_convert<Cfg::PITCH>(tile, threadIdx.x + 256 * j, r4, s);
  }
};
template <int COLS, int NCH_>
struct SplitKmajorLoader {       // chunk e: k = kt*32 + e / (COLS/4), columns 4*(e % (COLS/4)) .. +3
  static constexpr int NCH = NCH_, NREG = NCH_ * 4;
  const float* base; int64_t ld; int cols_valid, K;
  __device__ __forceinline__ void fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int k = kt * 32 + e / (COLS / 4), c = 4 * (e % (COLS / 4));
    const bool ok = k < K && c < cols_valid;
    const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? base + k * ld + c : g_fsraft_zero16);
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
  }
};

template <class Cfg>
__device__ __forceinline__ void store_tile(const GemmArgs& g, float* C, f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0, int n0, float inv) {
  const float alpha = g.alpha * inv;
#pragma unroll
  for (int nt = 0; nt < Cfg::TN; ++nt) {
    const int n = n0 + acc_col<Cfg>(nt);
    if (n >= g.N) continue;
#pragma unroll
    for (int mt = 0; mt < Cfg::TM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + acc_row<Cfg>(mt, r);
        if (m >= g.M) continue;
        float* o = C + (int64_t)m * g.ldc + n;
        const float v = alpha * acc[mt][nt][r];
        *o = g.accumulate ? *o + v : v;
      }
  }
}

__global__ __launch_bounds__(256) void gemm_split_nt_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[SNT::LDS_BYTES];
  const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128, b = blockIdx.z;
  const float sa = fs_scale_of_amax(fs_amax_load(g.a_amax)), sb = fs_scale_of_amax(fs_amax_load(g.b_amax));
  SplitRowLoader<SNT, SNT::NCH_A> la{g.A + b * g.sA + (int64_t)m0 * g.lda, g.lda, g.M - m0, g.K, sa};
  SplitRowLoader<SNT, SNT::NCH_B> lb{g.Bm + b * g.sB + (int64_t)n0 * g.ldb, g.ldb, g.N - n0, g.K, sb};
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  split_mainloop<SNT>(lds, (g.K + 31) / 32, la, lb, acc);
  store_tile<SNT>(g, g.C + b * g.sC, acc, m0, n0, fs_inv_scale(sa) * fs_inv_scale(sb));
}

// C[b][m][n] = alpha * sum_k A[b][k][m] * B[b][k][n]
__global__ __launch_bounds__(256) void gemm_split_tn_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[STN::LDS_BYTES];
  const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128, b = blockIdx.z;
  const float sa = fs_scale_of_amax(fs_amax_load(g.a_amax)), sb = fs_scale_of_amax(fs_amax_load(g.b_amax));
  SplitKmajorLoader<128, STN::NCH_A> la{g.A + b * g.sA + m0, g.lda, g.M - m0, g.K};
  SplitKmajorLoader<128, STN::NCH_B> lb{g.Bm + b * g.sB + n0, g.ldb, g.N - n0, g.K};
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  split_mainloop_tn<STN>(lds, (g.K + 31) / 32, la, lb, acc, nullptr, sa, sb);
  store_tile<STN>(g, g.C + b * g.sC, acc, m0, n0, fs_inv_scale(sa) * fs_inv_scale(sb));
}

int g_gemm_split = 1;

}  // namespace

// trans_b != 0: Bm is [N][K] (k contiguous); else Bm is [K][N].  Strides sA/sB/sC are per-batch element counts.
extern "C" int fsraft_gemm_f32(const float* A, int64_t lda, int64_t sA, const float* Bm, int64_t ldb, int64_t sB,
                               float* C, int64_t ldc, int64_t sC, int batch, int M, int N, int K, int trans_b,
                               float alpha, int accumulate, const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream) {
  if (!A || !Bm || !C || batch < 1 || M < 1 || N < 1 || K < 1) return FS_ERR_ARG;
  GemmArgs g{A, lda, sA, Bm, ldb, sB, C, ldc, sC, M, N, K, alpha, accumulate, a_amax, b_amax};
  dim3 grid(ceil_div(N, 128), ceil_div(M, 128), batch);
  const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && (K % 4 == 0) && (sA % 4 == 0) && (sB % 4 == 0) &&
                       ((uintptr_t)A % 16 == 0) && ((uintptr_t)Bm % 16 == 0);
  if (trans_b && g_gemm_split && aligned) hipLaunchKernelGGL(gemm_split_nt_kernel, grid, dim3(256), 0, stream, g);
  else if (trans_b) hipLaunchKernelGGL((gemm_kernel<CfgNT, true>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm_kernel<CfgNN, false>), grid, dim3(256), 0, stream, g);
  return fs_launch_status();
}

// C[b][m][n] = alpha * sum_k A[b][k][m] * Bm[b][k][n]  (both operands k-major), split-bf16 only.
// Requires lda, ldb, M, N multiples of 4 and 16-byte aligned bases.
extern "C" int fsraft_gemm_tn_split(const float* A, int64_t lda, int64_t sA, const float* Bm, int64_t ldb, int64_t sB,
                                    float* C, int64_t ldc, int64_t sC, int batch, int M, int N, int K, float alpha,
                                    int accumulate, const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream) {
  if (!A || !Bm || !C || batch < 1 || M < 1 || N < 1 || K < 1) return FS_ERR_ARG;
  if ((lda % 4) || (ldb % 4) || (M % 4) || (N % 4) || (sA % 4) || (sB % 4) || ((uintptr_t)A % 16) || ((uintptr_t)Bm % 16)) return FS_ERR_ARG;
  GemmArgs g{A, lda, sA, Bm, ldb, sB, C, ldc, sC, M, N, K, alpha, accumulate, a_amax, b_amax};
  dim3 grid(ceil_div(N, 128), ceil_div(M, 128), batch);
  hipLaunchKernelGGL(gemm_split_tn_kernel, grid, dim3(256), 0, stream, g);
  return fs_launch_status();
}

extern "C" int fsraft_set_gemm_split(int on) {
  g_gemm_split = on ? 1 : 0;
  return FS_OK;
}
