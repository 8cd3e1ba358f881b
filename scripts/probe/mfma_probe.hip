// Calibration probe: how fast does v_mfma_f32_32x32x2_f32 really run on this MI355X under the
// structure of our GEMM mainloop?  Variants add one ingredient at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* __restrict__ out, int KT) {
  __shared__ float lds[2 * (32 * 130 + 32 * 130)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  for (int i = threadIdx.x; i < 2 * 2 * 32 * 130; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float a0 = 1.0f + lane * 0.001f, a1 = 0.5f, b0 = 0.25f, b1 = 2.0f;
  const float* gp = g + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  float st[32];
  for (int i = 0; i < 32; ++i) st[i] = 0.f;
  for (int kt = 0; kt < KT; ++kt) {
    float* cur = lds + (kt & 1) * (2 * 32 * 130);
    float* nxt = lds + ((kt + 1) & 1) * (2 * 32 * 130);
    const float* As = cur + lh * 130 + (wave >> 1) * 64 + l31;
    const float* Bs = cur + 32 * 130 + lh * 130 + (wave & 1) * 64 + l31;
    float st2[32];
    if (V == 6) {
      // consume (store) the set loaded one iteration ago, load a new set now: distance = 2 tiles
#pragma unroll
      for (int i = 0; i < 32; ++i) st2[i] = st[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f32x4 v = *reinterpret_cast<const f32x4*>(gp + ((size_t)kt * 8 + j) * (size_t)gridDim.x * 1024);
        st[4 * j] = v[0]; st[4 * j + 1] = v[1]; st[4 * j + 2] = v[2]; st[4 * j + 3] = v[3];
      }
    } else if (V >= 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f32x4 v = *reinterpret_cast<const f32x4*>(gp + ((size_t)kt * 8 + j) * (size_t)gridDim.x * 1024);
        st[4 * j] = v[0]; st[4 * j + 1] = v[1]; st[4 * j + 2] = v[2]; st[4 * j + 3] = v[3];
      }
    }
    if (V >= 5) {
      float fa[2][2], fb[2][2];
      fa[0][0] = As[0]; fa[0][1] = As[32]; fb[0][0] = Bs[0]; fb[0][1] = Bs[32];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int c = ks & 1, n = c ^ 1;
        if (ks < 15) {
          fa[n][0] = As[(2 * ks + 2) * 130]; fa[n][1] = As[(2 * ks + 2) * 130 + 32];
          fb[n][0] = Bs[(2 * ks + 2) * 130]; fb[n][1] = Bs[(2 * ks + 2) * 130 + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (V >= 1) {
        a0 = As[(2 * ks) * 130]; a1 = As[(2 * ks) * 130 + 32];
        b0 = Bs[(2 * ks) * 130]; b1 = Bs[(2 * ks) * 130 + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    }
    if (V >= 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = threadIdx.x + 256 * j;
        const int row = (e / 8) & 127, kq = e % 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) nxt[(j / 4) * 32 * 130 + (kq * 4 + c) * 130 + row] = (V == 6 ? st2[4 * j + c] : st[4 * j + c]) + 0.001f;
      }
    }
    if (V >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(const char* name, int blocks, int KT, const float* g, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, g, out, KT);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, g, out, KT);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double fl = (double)blocks * 4 * KT * 64 * 4096.0;
  printf("%-34s blocks %5d KT %3d  %8.1f us  %6.1f TF (%4.1f%% of 157.3)\n", name, blocks, KT, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}

int main() {
  float *g, *out;
  size_t gbytes = (size_t)2048 * 1024 * 4 * 8 * 64;   // blocks*1024 floats * 8 loads * KT
  hipMalloc(&g, gbytes); hipMemset(g, 0, gbytes);
  hipMalloc(&out, 2048 * 256 * 4);
  for (int blocks : {440, 512, 1024}) {
    run<0>("V0 mfma only", blocks, 60, g, out);
    run<1>("V1 + LDS fragment reads", blocks, 60, g, out);
    run<2>("V2 + barrier per k-tile", blocks, 60, g, out);
    run<3>("V3 + transposed LDS stores", blocks, 60, g, out);
    run<4>("V4 + 8 global float4 loads/tile", blocks, 60, g, out);
    run<5>("V5 = V4 + fragment prefetch", blocks, 60, g, out);
    run<6>("V6 = V5 + loads 2 tiles ahead", blocks, 60, g, out);
  }
  return 0;
}
