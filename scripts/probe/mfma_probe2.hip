// Probe 2: same GEMM-shaped loop as mfma_probe V5, parametrised on BK (LDS footprint -> workgroups per CU)
// and on the wave tile (TM x TN 32x32 MFMA tiles per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BK, int TM, int TN>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* __restrict__ out, int KT) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, LDA = BM + 2, LDB = BN + 2, STAGE = BK * (LDA + LDB);
  __shared__ float lds[2 * STAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  for (int i = threadIdx.x; i < 2 * STAGE; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x16 acc[TM][TN];
  for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  constexpr int NF4 = (BM + BN) * BK / 4 / 256;
  const float* gp = g + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  float st[NF4 * 4];
  for (int i = 0; i < NF4 * 4; ++i) st[i] = 0.f;
  for (int kt = 0; kt < KT; ++kt) {
    float* cur = lds + (kt & 1) * STAGE;
    float* nxt = lds + ((kt + 1) & 1) * STAGE;
    const float* As = cur + lh * LDA + (wave >> 1) * (TM * 32) + l31;
    const float* Bs = cur + BK * LDA + lh * LDB + (wave & 1) * (TN * 32) + l31;
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      f32x4 v = *reinterpret_cast<const f32x4*>(gp + ((size_t)(kt * NF4 + j) % 512) * (size_t)gridDim.x * 1024);
      st[4 * j] = v[0]; st[4 * j + 1] = v[1]; st[4 * j + 2] = v[2]; st[4 * j + 3] = v[3];
    }
    float fa[2][TM], fb[2][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) fa[0][m] = As[m * 32];
#pragma unroll
    for (int n = 0; n < TN; ++n) fb[0][n] = Bs[n * 32];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int c = ks & 1, nx = c ^ 1;
      if (ks + 1 < BK / 2) {
#pragma unroll
        for (int m = 0; m < TM; ++m) fa[nx][m] = As[(2 * ks + 2) * LDA + m * 32];
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[nx][n] = Bs[(2 * ks + 2) * LDB + n * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][m], fb[c][n], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      const int e = threadIdx.x + 256 * j;
      const int row = (e / (BK / 4)) % BM, kq = e % (BK / 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) nxt[(kq * 4 + c) * LDA + row] = st[4 * j + c] + 0.001f;
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int BK, int TM, int TN>
void run(int blocks, int Ktotal, const float* g, float* out) {
  int KT = Ktotal / BK;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<BK, TM, TN>), dim3(blocks), dim3(256), 0, 0, g, out, KT);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<BK, TM, TN>), dim3(blocks), dim3(256), 0, 0, g, out, KT);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double fl = (double)blocks * 4 * KT * (BK / 2) * TM * TN * 4096.0;
  int occ = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<BK, TM, TN>, 256, 0);
  printf("BK %2d tile %3dx%3d  blocks %5d (%d/CU resident)  %8.1f us  %6.1f TF (%4.1f%%)\n", BK, 2 * TM * 32, 2 * TN * 32, blocks, occ, ms * 1e3,
         fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}

int main() {
  float *g, *out;
  size_t gbytes = (size_t)4096 * 1024 * 4 * 512;
  (void)hipMalloc(&g, gbytes); (void)hipMemset(g, 0, gbytes);
  (void)hipMalloc(&out, 8192 * 256 * 4);
  // same total work per CU in every row: 128x128 tiles x 512 blocks x K=1920  ==  64x64 tiles x 2048 blocks ...
  run<32, 2, 2>(512, 1920, g, out);
  run<16, 2, 2>(512, 1920, g, out);
  run<16, 2, 2>(1024, 1920, g, out);
  run<8, 2, 2>(1024, 1920, g, out);
  run<32, 1, 2>(1024, 1920, g, out);
  run<16, 1, 2>(1024, 1920, g, out);
  run<32, 1, 1>(2048, 1920, g, out);
  run<16, 1, 1>(2048, 1920, g, out);
  run<32, 2, 2>(440, 1920, g, out);
  run<16, 2, 2>(440, 1920, g, out);
  run<16, 1, 2>(880, 1920, g, out);
  run<16, 1, 1>(1760, 1920, g, out);
  run<32, 1, 1>(1760, 1920, g, out);
  return 0;
}
