// Batched GEMMs on pre-split operands (gemm_rec.hpp) and the kernels that produce records from fp32 data.
//   NT:  C[b][m][n] (+)= alpha * sum_k A[b][m][k] * B[b][n][k]        A, B: rows of K / 32 records (128 bytes each)
// Used by the volume backward (dF1 = s * F2cat . dV^T with K = P, pytorch/core/corr.py:52-60's autograd).
#include "gemm_rec.hpp"

namespace {

using G = RecCfg<256, 128, 4, 2>;

struct RecGemmArgs {
  const char* A; int64_t sA; unsigned pitchA;      // bytes
  const char* Bm; int64_t sB; unsigned pitchB;
  float* C; int64_t ldc, sC;
  int M, N, KT, ksplit;
  float alpha; int atomic;
  // sparse k (fsraft_gemm_rec_*_list): per (batch entry, tile of the sparse operand) `kcount` k-tile numbers in `klist`
  // (kl_stride ints apart); kl_by_n: the lists belong to the N tiles (B operand rows), else to the M tiles
  const int* klist; const int* kcount; int kl_stride, kl_by_n;
  const unsigned* a_amax; const unsigned* b_amax;   // amax words the records of A / B were split with (NULL: scale 1)
};

// alpha times the factor that takes the accumulators of scaled records back (exact: powers of two)
__device__ __forceinline__ float rec_alpha(const RecGemmArgs& g) {
  return g.alpha * fs_inv_scale(fs_scale_of_amax(fs_amax_load(g.a_amax))) * fs_inv_scale(fs_scale_of_amax(fs_amax_load(g.b_amax)));
}

int g_rec_mfma16 = 0;      // fsraft_set_tuning-style switch (fsraft_set_rec_mfma16): the NT kernel on v_mfma_f32_16x16x32_bf16

template <bool M16, bool LIST = false>
__global__ __launch_bounds__(512) void gemm_rec_nt_kernel(RecGemmArgs g) {
  __shared__ __attribute__((aligned(1024))) char lds[G::LDS_BYTES];
  const int ntn = (g.N + G::BN - 1) / G::BN;
  const int tile = blockIdx.x, ks = blockIdx.y, b = blockIdx.z;
  const int n0 = (tile % ntn) * G::BN, m0 = (tile / ntn) * G::BM;
  int KTall = g.KT;
  const int* list = nullptr;
  if constexpr (LIST) {
    const int ntm = (g.M + G::BM - 1) / G::BM;
    const int lt = g.kl_by_n ? b * ntn + tile % ntn : b * ntm + tile / ntn;
    KTall = g.kcount[lt];
    list = g.klist + (int64_t)lt * g.kl_stride;
  }
  const int per = (KTall + g.ksplit - 1) / g.ksplit;
  const int kt0 = ks * per, kt = min(per, KTall - kt0);
  const char* A = g.A + b * g.sA + (int64_t)m0 * g.pitchA;
  const char* Bm = g.Bm + b * g.sB + (int64_t)n0 * g.pitchB;
  const int ar = min(G::BM, g.M - m0), br = min(G::BN, g.N - n0);
  RecOperands<G> o;
  rec_setup<G>(o, A, (unsigned)min((int64_t)ar * g.pitchA, (int64_t)0x7fffffff), g.pitchA, ar, Bm,
               (unsigned)min((int64_t)br * g.pitchB, (int64_t)0x7fffffff), g.pitchB, br);
  f32x16 acc[G::TM][G::TN];
#pragma unroll
  for (int i = 0; i < G::TM; ++i)
#pragma unroll
    for (int j = 0; j < G::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if constexpr (LIST) {
    __shared__ int klds[REC_LIST_MAX];
    for (int i = threadIdx.x; i < kt; i += G::NT) klds[i] = list[kt0 + i];
    __syncthreads();
    RecListA<G> pa;
#pragma unroll
    for (int j = 0; j < G::NPA; ++j) pa.va[j] = o.va[j];
    pa.list = (const int __attribute__((address_space(3)))*)klds; pa.cnt = kt > 0 ? kt : 1; pa.step = 128u;
    rec_mainloop<G>(lds, o, pa, 0, kt, acc);
  } else {
    RecPlainA<G> pa;
#pragma unroll
    for (int j = 0; j < G::NPA; ++j) pa.va[j] = o.va[j];
    pa.kt0 = kt0; pa.step = 128u;
    if constexpr (M16) rec_mainloop16<G>(lds, o, pa, kt0, kt, acc);
    else rec_mainloop<G>(lds, o, pa, kt0, kt, acc);
  }
  float* C = g.C + b * g.sC;
  const float alpha = rec_alpha(g);
#pragma unroll
  for (int nt = 0; nt < G::TN; ++nt) {
#pragma unroll
    for (int mt = 0; mt < G::TM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (M16 ? rec16_col<G>(nt, r) : rec_col<G>(nt));
        const int m = m0 + (M16 ? rec16_row<G>(mt, r) : rec_row<G>(mt, r));
        if (m >= g.M || n >= g.N) continue;
        float* p = C + (int64_t)m * g.ldc + n;
        const float v = alpha * acc[mt][nt][r];
        if (g.atomic) atomicAdd(p, v);
        else gstore1(p, v);
      }
  }
}

// C[b][m][n] (+)= alpha * sum_k A[b][k][m] * B[b][k][n], both operands k-major rows of records (along m resp. n)
using GT = RecCfg<256, 128, 4, 2, 3, true>;

template <bool LIST = false>
__global__ __launch_bounds__(512) void gemm_rec_tn_kernel(RecGemmArgs g, int K) {
  __shared__ __attribute__((aligned(1024))) char lds[GT::LDS_BYTES];
  const int ntn = (g.N + GT::BN - 1) / GT::BN;
  const int tile = blockIdx.x, ks = blockIdx.y, b = blockIdx.z;
  const int n0 = (tile % ntn) * GT::BN, m0 = (tile / ntn) * GT::BM;
  int KTall = g.KT;
  const int* list = nullptr;
  if constexpr (LIST) {
    const int ntm = (g.M + GT::BM - 1) / GT::BM;
    const int lt = g.kl_by_n ? b * ntn + tile % ntn : b * ntm + tile / ntn;
    KTall = g.kcount[lt];
    list = g.klist + (int64_t)lt * g.kl_stride;
  }
  const int per = (KTall + g.ksplit - 1) / g.ksplit;
  const int kt0 = ks * per, kt = min(per, KTall - kt0);
  const char* A = g.A + b * g.sA + (int64_t)m0 * 4;
  const char* Bm = g.Bm + b * g.sB + (int64_t)n0 * 4;
  RecOperands<GT> o;
  rec_setup_km<GT>(o, A, (unsigned)min((int64_t)K * g.pitchA - (int64_t)m0 * 4, (int64_t)0x7fffffff), g.pitchA, Bm,
                   (unsigned)min((int64_t)K * g.pitchB - (int64_t)n0 * 4, (int64_t)0x7fffffff), g.pitchB);
  f32x16 acc[GT::TM][GT::TN];
#pragma unroll
  for (int i = 0; i < GT::TM; ++i)
#pragma unroll
    for (int j = 0; j < GT::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if constexpr (LIST) {
    __shared__ int klds[REC_LIST_MAX];
    for (int i = threadIdx.x; i < kt; i += GT::NT) klds[i] = list[kt0 + i];
    __syncthreads();
    RecListA<GT> pa;
#pragma unroll
    for (int j = 0; j < GT::NPA; ++j) pa.va[j] = o.va[j];
    pa.list = (const int __attribute__((address_space(3)))*)klds; pa.cnt = kt > 0 ? kt : 1; pa.step = 32u * g.pitchA;
    rec_mainloop<GT>(lds, o, pa, 0, kt, acc);
  } else {
    RecPlainA<GT> pa;
#pragma unroll
    for (int j = 0; j < GT::NPA; ++j) pa.va[j] = o.va[j];
    pa.kt0 = kt0; pa.step = 32u * g.pitchA;
    rec_mainloop<GT>(lds, o, pa, kt0, kt, acc);
  }
  float* C = g.C + b * g.sC;
  const float alpha = rec_alpha(g);
#pragma unroll
  for (int nt = 0; nt < GT::TN; ++nt) {
    const int n = n0 + rec_col<GT>(nt);
    if (n >= g.N) continue;
#pragma unroll
    for (int mt = 0; mt < GT::TM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + rec_row<GT>(mt, r);
        if (m >= g.M) continue;
        float* p = C + (int64_t)m * g.ldc + n;
        const float v = alpha * acc[mt][nt][r];
        if (g.atomic) atomicAdd(p, v);
        else gstore1(p, v);
      }
  }
}

// fp32 rows -> records: dst row = ceil(K / 32) records, the tail of the last one zero
__global__ __launch_bounds__(256) void to_records_kernel(const float* __restrict__ src, int64_t ld, char* __restrict__ dst, int64_t dpitch,
                                                         int64_t rows, int K, int KR, const unsigned* __restrict__ amax) {
  const int64_t total = rows * KR * 4;                       // 8-float units
  const float sc = fs_scale_of_amax(fs_amax_load(amax));
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int u = (int)(e % (KR * 4));
    const int64_t row = e / (KR * 4);
    const int k = u * 8;
    float v[8];
    const float* s = src + row * ld + k;
    if (k + 8 <= K && ((ld & 3) == 0)) {
      const f32x4 a = gload4(s), b = gload4(s + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = k + i < K ? gload1(s + i) : 0.f;
    }
    uint2 h0, l0, h1, l1;
    rec_split4(v, h0, l0, sc);
    rec_split4(v + 4, h1, l1, sc);
    char* d = dst + row * dpitch + (u >> 2) * 128 + (u & 3) * 16;
    gstore4(d, __builtin_bit_cast(f32x4, u32x4{h0.x, h0.y, h1.x, h1.y}));
    gstore4(d + 64, __builtin_bit_cast(f32x4, u32x4{l0.x, l0.y, l1.x, l1.y}));
  }
}

}  // namespace

// src [rows][K] fp32 (row pitch ld floats) -> dst [rows][ceil(K/32)] records of [32 hi | 32 lo] bf16 (128 bytes), row pitch
// dst_ld floats (>= 32 * ceil(K/32), a multiple of 32; 0 = dense)
namespace {
// C[b][m][0..N) = 0 for the split-K / accumulate path.  A kernel, not hipMemsetAsync: inside a captured hipGraph the memset
// node of these calls left the matrix untouched on every replay after the first (the partial tiles were then added to the
// previous replay's result -- found by the round-2 graph-replay checker, docs/history); a fill kernel is an ordinary graph node.
__global__ __launch_bounds__(256) void zero_matrices_kernel(float* __restrict__ C, int64_t ldc, int64_t sC, int M, int N) {
  float* row = C + (int64_t)blockIdx.z * sC + (int64_t)blockIdx.y * ldc;
  for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) row[n] = 0.f;
}
int zero_matrices(float* C, int64_t ldc, int64_t sC, int batch, int M, int N, hipStream_t stream) {
  if (ldc == N && sC == (int64_t)M * N && (int64_t)batch * M * N < (int64_t)0x7fffffff) {      // dense: one long row
    const int total = batch * M * N;
    int gx = (total + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(zero_matrices_kernel, dim3(gx), dim3(256), 0, stream, C, (int64_t)0, (int64_t)0, 1, total);
  } else {
    if (M > 65535 || batch > 65535) return FS_ERR_ARG;
    int gx = (N + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(zero_matrices_kernel, dim3(gx, M, batch), dim3(256), 0, stream, C, ldc, sC, M, N);
  }
  return fs_launch_status();
}
}  // namespace

// amax: the word of src (fsraft_amax*, or a producer's dst_amax); the records hold fp16 pieces of x * scale(amax).  NULL: scale 1.
extern "C" int fsraft_to_records(const float* src, int64_t ld, void* dst, int64_t dst_ld, int64_t rows, int K, const unsigned* amax,
                                 hipStream_t stream) {
  if (!src || !dst || rows < 1 || K < 1 || ((uintptr_t)dst % 16) || ((uintptr_t)src % 16) || ((uintptr_t)amax & 3)) return FS_ERR_ARG;
  const int KR = (K + 31) / 32;
  if (dst_ld == 0) dst_ld = (int64_t)KR * 32;
  if (dst_ld < (int64_t)KR * 32 || dst_ld % 32) return FS_ERR_ARG;
  const int64_t total = rows * KR * 4;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(to_records_kernel, dim3(blocks), dim3(256), 0, stream, src, ld, (char*)dst, dst_ld * 4, rows, K, KR, amax);
  return fs_launch_status();
}

// C[b][m][n] = alpha * sum_k A[b][m][k] B[b][n][k] on record operands: A [batch][M] rows of K / 32 records with row pitch lda
// floats, B [batch][N] rows with pitch ldb (lda, ldb multiples of 32, >= K; 0 = K: dense), K % 32 == 0; sA / sB: batch
// strides in BYTES.  ksplit > 1 splits K over workgroups that add their partial tiles with fp32 atomics (C is zeroed first
// unless accumulate != 0); ksplit == 1 and accumulate == 0: plain stores.
namespace {
int gemm_rec_nt_impl(const void* A, int64_t lda, int64_t sA, const void* Bm, int64_t ldb, int64_t sB, float* C, int64_t ldc, int64_t sC,
                     int batch, int M, int N, int K, float alpha, int ksplit, int accumulate, const int* klist, const int* kcount,
                     int kl_stride, int kl_by_n, const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream) {
  if (!A || !Bm || !C || batch < 1 || M < 1 || N < 1 || K < 32 || (K % 32) || ksplit < 1 || ((uintptr_t)A % 16) || ((uintptr_t)Bm % 16))
    return FS_ERR_ARG;
  if (lda == 0) lda = K;
  if (ldb == 0) ldb = K;
  if (lda < K || ldb < K || (lda % 32) || (ldb % 32)) return FS_ERR_ARG;
  const int KT = K / 32;
  if (ksplit > KT) ksplit = KT;
  const bool atomic = ksplit > 1 || accumulate;
  if (atomic && !accumulate) {
    const int rc = zero_matrices(C, ldc, sC, batch, M, N, stream);
    if (rc) return rc;
  }
  RecGemmArgs g{(const char*)A, sA, (unsigned)lda * 4u, (const char*)Bm, sB, (unsigned)ldb * 4u, C, ldc, sC, M, N, KT, ksplit, alpha, atomic ? 1 : 0,
                klist, kcount, kl_stride, kl_by_n, a_amax, b_amax};
  dim3 grid(ceil_div(N, G::BN) * ceil_div(M, G::BM), ksplit, batch);
  if (klist) hipLaunchKernelGGL((gemm_rec_nt_kernel<false, true>), grid, dim3(512), 0, stream, g);
  else if (g_rec_mfma16) hipLaunchKernelGGL((gemm_rec_nt_kernel<true, false>), grid, dim3(512), 0, stream, g);
  else hipLaunchKernelGGL((gemm_rec_nt_kernel<false, false>), grid, dim3(512), 0, stream, g);
  return fs_launch_status();
}
}  // namespace

extern "C" int fsraft_gemm_rec_nt(const void* A, int64_t lda, int64_t sA, const void* Bm, int64_t ldb, int64_t sB, float* C,
                                  int64_t ldc, int64_t sC, int batch, int M, int N, int K, float alpha, int ksplit, int accumulate,
                                  const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream) {
  return gemm_rec_nt_impl(A, lda, sA, Bm, ldb, sB, C, ldc, sC, batch, M, N, K, alpha, ksplit, accumulate, nullptr, nullptr, 0, 0, a_amax, b_amax, stream);
}
// The same contraction over LISTED k-tiles only: klist[(b * tiles + tile) * kl_stride + i], i < kcount[b * tiles + tile], ascending
// k-tile numbers (32 k each) for every tile of the sparse operand -- the 128-row tiles of B (kl_by_n != 0) or the 256-row tiles
// of A; everything the lists leave out must be zero records (fsraft_corr_bwd_ktiles builds such lists for the gradient volume).
extern "C" int fsraft_gemm_rec_nt_list(const void* A, int64_t lda, int64_t sA, const void* Bm, int64_t ldb, int64_t sB, float* C,
                                       int64_t ldc, int64_t sC, int batch, int M, int N, int K, float alpha, int ksplit,
                                       int accumulate, const int* klist, const int* kcount, int kl_stride, int kl_by_n,
                                       const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream) {
  if (!klist || !kcount || kl_stride < 1 || K / 32 > REC_LIST_MAX) return FS_ERR_ARG;
  return gemm_rec_nt_impl(A, lda, sA, Bm, ldb, sB, C, ldc, sC, batch, M, N, K, alpha, ksplit, accumulate, klist, kcount, kl_stride,
                          kl_by_n, a_amax, b_amax, stream);
}

// C[b][m][n] = alpha * sum_k A[b][k][m] B[b][k][n]: A [batch][K][lda floats] with the records along m, B [batch][K][ldb
// floats] with the records along n (lda, ldb multiples of 32 covering M resp. N; sA / sB batch strides in BYTES).  K need not
// be a multiple of 32 (rows beyond K read as zeros).  ksplit / accumulate as fsraft_gemm_rec_nt.
namespace {
int gemm_rec_tn_impl(const void* A, int64_t lda, int64_t sA, const void* Bm, int64_t ldb, int64_t sB, float* C, int64_t ldc, int64_t sC,
                     int batch, int M, int N, int K, float alpha, int ksplit, int accumulate, const int* klist, const int* kcount,
                     int kl_stride, int kl_by_n, const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream) {
  if (!A || !Bm || !C || batch < 1 || M < 1 || N < 1 || K < 1 || ksplit < 1 || (lda % 32) || (ldb % 32) || lda < M || ldb < N ||
      ((uintptr_t)A % 16) || ((uintptr_t)Bm % 16))
    return FS_ERR_ARG;
  const int KT = (K + 31) / 32;
  if (ksplit > KT) ksplit = KT;
  const bool atomic = ksplit > 1 || accumulate;
  if (atomic && !accumulate) {
    const int rc = zero_matrices(C, ldc, sC, batch, M, N, stream);
    if (rc) return rc;
  }
  RecGemmArgs g{(const char*)A, sA, (unsigned)lda * 4u, (const char*)Bm, sB, (unsigned)ldb * 4u, C, ldc, sC, M, N, KT, ksplit, alpha, atomic ? 1 : 0,
                klist, kcount, kl_stride, kl_by_n, a_amax, b_amax};
  dim3 grid(ceil_div(N, GT::BN) * ceil_div(M, GT::BM), ksplit, batch);
  if (klist) hipLaunchKernelGGL(gemm_rec_tn_kernel<true>, grid, dim3(512), 0, stream, g, K);
  else hipLaunchKernelGGL(gemm_rec_tn_kernel<false>, grid, dim3(512), 0, stream, g, K);
  return fs_launch_status();
}
}  // namespace

extern "C" int fsraft_gemm_rec_tn(const void* A, int64_t lda, int64_t sA, const void* Bm, int64_t ldb, int64_t sB, float* C,
                                  int64_t ldc, int64_t sC, int batch, int M, int N, int K, float alpha, int ksplit, int accumulate,
                                  const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream) {
  return gemm_rec_tn_impl(A, lda, sA, Bm, ldb, sB, C, ldc, sC, batch, M, N, K, alpha, ksplit, accumulate, nullptr, nullptr, 0, 0, a_amax, b_amax, stream);
}
// k-major twin of fsraft_gemm_rec_nt_list: the lists belong to the 256-column tiles of A (M tiles; kl_by_n == 0) or the
// 128-column tiles of B, a k-tile is 32 consecutive k-rows.
extern "C" int fsraft_gemm_rec_tn_list(const void* A, int64_t lda, int64_t sA, const void* Bm, int64_t ldb, int64_t sB, float* C,
                                       int64_t ldc, int64_t sC, int batch, int M, int N, int K, float alpha, int ksplit,
                                       int accumulate, const int* klist, const int* kcount, int kl_stride, int kl_by_n,
                                       const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream) {
  if (!klist || !kcount || kl_stride < 1 || (K + 31) / 32 > REC_LIST_MAX) return FS_ERR_ARG;
  return gemm_rec_tn_impl(A, lda, sA, Bm, ldb, sB, C, ldc, sC, batch, M, N, K, alpha, ksplit, accumulate, klist, kcount, kl_stride,
                          kl_by_n, a_amax, b_amax, stream);
}

extern "C" int fsraft_set_rec_mfma16(int on) {
  g_rec_mfma16 = on ? 1 : 0;
  return FS_OK;
}
