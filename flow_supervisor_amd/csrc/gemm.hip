// Batched fp32 GEMMs on the MFMA core, used by the backward of the all-pairs volume
// (row a1 backward, K5 in SURVEY.md 2a; reference: autograd of torch.matmul in
// pytorch/core/corr.py:52-60):
//   NT:  C[b][m][n] = alpha * sum_k A[b][m][k] * Bm[b][n][k]     (dF1 = s * f2 . dV^T  laid out [c][i])
//   NN:  C[b][m][n] = alpha * sum_k A[b][m][k] * Bm[b][k][n]     (dF2 = s * f1 . dV    laid out [c][j])
// With A = the NCHW feature map ([C][N] per sample) both results come out directly in
// NCHW, so no transposes are needed on either side.
#include "gemm_core.hpp"

namespace {

using CfgNT = GemmCfg<128, 128, 32, 2, 2, 2, 2>;
using CfgNN = GemmCfg<128, 128, 32, 2, 2, 2, 0>;

struct GemmArgs {
  const float* A; int64_t lda, sA;
  const float* Bm; int64_t ldb, sB;
  float* C; int64_t ldc, sC;
  int M, N, K; float alpha; int accumulate;
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

}  // namespace

// trans_b != 0: Bm is [N][K] (k contiguous); else Bm is [K][N].  Strides sA/sB/sC are per-batch element counts.
extern "C" int fsraft_gemm_f32(const float* A, int64_t lda, int64_t sA, const float* Bm, int64_t ldb, int64_t sB,
                               float* C, int64_t ldc, int64_t sC, int batch, int M, int N, int K, int trans_b,
                               float alpha, int accumulate, hipStream_t stream) {
  if (!A || !Bm || !C || batch < 1 || M < 1 || N < 1 || K < 1) return FS_ERR_ARG;
  GemmArgs g{A, lda, sA, Bm, ldb, sB, C, ldc, sC, M, N, K, alpha, accumulate};
  dim3 grid(ceil_div(N, 128), ceil_div(M, 128), batch);
  if (trans_b) hipLaunchKernelGGL((gemm_kernel<CfgNT, true>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm_kernel<CfgNN, false>), grid, dim3(256), 0, stream, g);
  return fs_launch_status();
}
