// HBM-bound kernels of the GMA variant:
//   * row softmax of the N x N similarity map and its backward  (Attention.forward, pytorch/core/gma.py:71-74)
//   * motion_global = motion + gamma * (attn @ v) and its backward (Aggregate.forward, gma.py:113)
// The GEMMs around them (q k^T, attn @ v and their transposes) run on fsraft_gemm_f32 / fsraft_gemm_tn_split.
#include "common.hpp"
#include "gemm_rec.hpp"      // rec_split4: the [32 hi | 32 lo] bf16 records the GEMMs on the attention map read

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 256 threads = 4 waves; reduce across the workgroup through 4 LDS slots
template <bool MAX>
__device__ __forceinline__ float block_reduce(float v, float* red) {
  v = MAX ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return MAX ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup per row, in place.  The row is read once from HBM (float4), held in LDS (n <= 16384)
// and written once: 8 bytes per element.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, int n) {
  extern __shared__ float row[];
  __shared__ float red[4];
  float* p = S + (int64_t)blockIdx.x * n;
  const int n4 = (n & 3) ? 0 : (n >> 2);      // rows are 16-byte aligned only when n % 4 == 0
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
    reinterpret_cast<f32x4*>(row)[i] = v;
    m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) { row[i] = p[i]; m = fmaxf(m, p[i]); }
  m = block_reduce<true>(m, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float e = __expf(row[i] - m); row[i] = e; s += e; }
  s = block_reduce<false>(s, red);
  const float inv = 1.0f / s;
  for (int i = threadIdx.x; i < n4; i += 256) {
    f32x4 v = reinterpret_cast<f32x4*>(row)[i];
    v *= inv;
    reinterpret_cast<f32x4*>(p)[i] = v;
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) p[i] = row[i] * inv;
}

// dS = A * (dA - sum_j dA_j A_j), written over dA.  A and dA are read once (LDS keeps the row of dA*A inputs).
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ A, float* __restrict__ dA, int n) {
  extern __shared__ float row[];          // [2][n]: A row, dA row
  __shared__ float red[4];
  const float* a = A + (int64_t)blockIdx.x * n;
  float* d = dA + (int64_t)blockIdx.x * n;
  float* ra = row;
  float* rd = row + ((n + 3) & ~3);
  const int n4 = (n & 3) ? 0 : (n >> 2);
  float dot = 0.f;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 av = reinterpret_cast<const f32x4*>(a)[i];
    const f32x4 dv = reinterpret_cast<const f32x4*>(d)[i];
    reinterpret_cast<f32x4*>(ra)[i] = av;
    reinterpret_cast<f32x4*>(rd)[i] = dv;
    dot += av[0] * dv[0] + av[1] * dv[1] + av[2] * dv[2] + av[3] * dv[3];
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) { ra[i] = a[i]; rd[i] = d[i]; dot += a[i] * d[i]; }
  dot = block_reduce<false>(dot, red);
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 av = reinterpret_cast<f32x4*>(ra)[i];
    f32x4 dv = reinterpret_cast<f32x4*>(rd)[i];
    dv = av * (dv - dot);
    reinterpret_cast<f32x4*>(d)[i] = dv;
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) d[i] = ra[i] * (rd[i] - dot);
}

// The same softmax, but the probabilities leave as RECORDS ([32 bf16 hi | 32 bf16 lo] per 32 columns, gemm_rec.hpp) written over
// the logits: for n % 32 == 0 a row of records is exactly as long as the fp32 row, so the map exists ONCE -- in the form every
// later reader takes it in (attn @ v and attn^T @ dagg of all iterations, the softmax backward below) -- and the separate
// fp32 -> records pass (read + write of the whole map) is gone.
// Probabilities are bounded by 1: their records are split with the fixed scale 2^13 -- the scale of an amax word holding 1.0,
// which is the word the GEMMs reading them are given.
constexpr float GMA_P_SCALE = 8192.0f;
__global__ __launch_bounds__(256) void softmax_rows_rec_kernel(float* __restrict__ S, int n) {
  extern __shared__ float row[];
  __shared__ float red[4];
  float* p = S + (int64_t)blockIdx.x * n;
  const int n4 = n >> 2;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
    reinterpret_cast<f32x4*>(row)[i] = v;
    m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  m = block_reduce<true>(m, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float e = __expf(row[i] - m); row[i] = e; s += e; }
  s = block_reduce<false>(s, red);
  const float inv = 1.0f / s;
  char* out = reinterpret_cast<char*>(p);
  for (int u = threadIdx.x; u < (n >> 3); u += 256) {          // 8-float units: 16 bytes of hi and 16 of lo
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {      // (the product rounded as the dense kernel stores it: the empty asm keeps hipcc from
      v[i] = row[u * 8 + i] * inv;       //  contracting it into the split's x - hi as an fma on the unrounded product)
      asm volatile("" : "+v"(v[i]));
    }
    uint2 h0, l0, h1, l1;
    rec_split4(v, h0, l0, GMA_P_SCALE);
    rec_split4(v + 4, h1, l1, GMA_P_SCALE);
    char* d = out + (u >> 2) * 128 + (u & 3) * 16;
    *reinterpret_cast<u32x4*>(d) = u32x4{h0.x, h0.y, h1.x, h1.y};
    *reinterpret_cast<u32x4*>(d + 64) = u32x4{l0.x, l0.y, l1.x, l1.y};
  }
}

// Backward of the above: A as records (a = hi + lo), dA fp32 in, dS = A * (dA - sum_j dA_j A_j) out AS RECORDS over dA -- the
// two GEMMs that consume dS read records, so neither the fp32 dS nor a conversion pass of it exists.
// ds_amax: a word bounding |dS| (|dS| <= 2 max |dA|), the one the GEMMs reading dS are given
__global__ __launch_bounds__(256) void softmax_rows_bwd_rec_kernel(const char* __restrict__ A, float* __restrict__ dA, int n,
                                                                   const unsigned* __restrict__ ds_amax) {
  const float ds_scale = fs_scale_of_amax(fs_amax_load(ds_amax));
  extern __shared__ float row[];          // [2][n]: A row, dA row
  __shared__ float red[4];
  const char* a = A + (int64_t)blockIdx.x * n * 4;
  float* d = dA + (int64_t)blockIdx.x * n;
  float* ra = row;
  float* rd = row + n;
  float dot = 0.f;
  for (int u = threadIdx.x; u < (n >> 3); u += 256) {
    const char* s = a + (u >> 2) * 128 + (u & 3) * 16;
    const u32x4 h = *reinterpret_cast<const u32x4*>(s), l = *reinterpret_cast<const u32x4*>(s + 64);
    const f32x4 d0 = reinterpret_cast<const f32x4*>(d)[u * 2], d1 = reinterpret_cast<const f32x4*>(d)[u * 2 + 1];
    float av[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) fs_unsplit2(h[i], l[i], 1.0f / GMA_P_SCALE, av[2 * i], av[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 8; ++i) ra[u * 8 + i] = av[i];
    reinterpret_cast<f32x4*>(rd)[u * 2] = d0;
    reinterpret_cast<f32x4*>(rd)[u * 2 + 1] = d1;
#pragma unroll
    for (int i = 0; i < 4; ++i) dot += av[i] * d0[i] + av[4 + i] * d1[i];
  }
  dot = block_reduce<false>(dot, red);
  char* out = reinterpret_cast<char*>(d);
  for (int u = threadIdx.x; u < (n >> 3); u += 256) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = ra[u * 8 + i] * (rd[u * 8 + i] - dot);
      asm volatile("" : "+v"(v[i]));
    }
    uint2 h0, l0, h1, l1;
    rec_split4(v, h0, l0, ds_scale);
    rec_split4(v + 4, h1, l1, ds_scale);
    char* o = out + (u >> 2) * 128 + (u & 3) * 16;
    *reinterpret_cast<u32x4*>(o) = u32x4{h0.x, h0.y, h1.x, h1.y};
    *reinterpret_cast<u32x4*>(o + 64) = u32x4{l0.x, l0.y, l1.x, l1.y};
  }
}

// Rows too long for LDS: same arithmetic, the row is re-read from global memory (L2) instead.
__global__ __launch_bounds__(256) void softmax_rows_big_kernel(float* __restrict__ S, int n) {
  __shared__ float red[4];
  float* p = S + (int64_t)blockIdx.x * n;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, p[i]);
  m = block_reduce<true>(m, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += __expf(p[i] - m);
  s = block_reduce<false>(s, red);
  const float inv = 1.0f / s;
  for (int i = threadIdx.x; i < n; i += 256) p[i] = __expf(p[i] - m) * inv;
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_big_kernel(const float* __restrict__ A, float* __restrict__ dA, int n) {
  __shared__ float red[4];
  const float* a = A + (int64_t)blockIdx.x * n;
  float* d = dA + (int64_t)blockIdx.x * n;
  float dot = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) dot += a[i] * d[i];
  dot = block_reduce<false>(dot, red);
  for (int i = threadIdx.x; i < n; i += 256) d[i] = a[i] * (d[i] - dot);
}

// dst[m][c] = x[m][c] + gamma * y[m][c]   (C % 4 == 0, 16-byte aligned rows)
__global__ __launch_bounds__(256) void gma_mix_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y,
                                                          int ldy, const float* __restrict__ gamma, float* __restrict__ dst,
                                                          int ldd, int64_t M, int C, unsigned* __restrict__ dst_amax) {
  const float g = gamma[0];
  const int c4n = C >> 2;
  const int64_t total = M * c4n;
  unsigned amx = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / c4n; const int c = (int)(e % c4n) * 4;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + m * ldx + c);
    const f32x4 yv = *reinterpret_cast<const f32x4*>(y + m * ldy + c);
    const f32x4 r = xv + g * yv;
    *reinterpret_cast<f32x4*>(dst + m * ldd + c) = r;
    amx = fs_umax(amx, fs_abs_bits4(r));
    if (dst_amax && e < (int64_t)gridDim.x * 256) fs_amax_early(dst_amax, amx);
  }
  __shared__ unsigned ared[4];
  if (dst_amax) fs_amax_commit(dst_amax, amx, ared);      // (nullable) word of dst, raised
}

// d = dL/d dst:  dx[m][c] += d;  dy[m][c] = gamma * d;  dgamma += sum d * y
__global__ __launch_bounds__(256) void gma_mix_bwd_kernel(const float* __restrict__ d, int ldd, const float* __restrict__ y,
                                                          int ldy, const float* __restrict__ gamma, float* __restrict__ dx,
                                                          int ldx, float* __restrict__ dy, int lddy,
                                                          float* __restrict__ dgamma, int64_t M, int C,
                                                          unsigned* __restrict__ dx_amax, unsigned* __restrict__ dy_amax) {   // (nullable) words of dx / dy, raised
  __shared__ float red[4];
  const float g = gamma[0];
  const int c4n = C >> 2;
  const int64_t total = M * c4n;
  float acc = 0.f;
  unsigned ax = 0u, ay = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / c4n; const int c = (int)(e % c4n) * 4;
    const f32x4 dv = *reinterpret_cast<const f32x4*>(d + m * ldd + c);
    const f32x4 yv = *reinterpret_cast<const f32x4*>(y + m * ldy + c);
    f32x4* px = reinterpret_cast<f32x4*>(dx + m * ldx + c);
    const f32x4 nx = *px + dv, ny = g * dv;
    *px = nx;
    *reinterpret_cast<f32x4*>(dy + m * lddy + c) = ny;
    ax = fs_umax(ax, fs_abs_bits4(nx)); ay = fs_umax(ay, fs_abs_bits4(ny));
    acc += dv[0] * yv[0] + dv[1] * yv[1] + dv[2] * yv[2] + dv[3] * yv[3];
  }
  acc = block_reduce<false>(acc, red);
  if (threadIdx.x == 0) atomicAdd(dgamma, acc);
  __shared__ unsigned ared[4];
  if (dx_amax) fs_amax_commit(dx_amax, ax, ared);
  if (dy_amax) fs_amax_commit(dy_amax, ay, ared);
}

inline int grid_for(int64_t work) { int64_t g = (work + 255) / 256; return (int)(g < 1 ? 1 : g > 4096 ? 4096 : g); }

}  // namespace

extern "C" int fsraft_softmax_rows(float* S, int64_t rows, int n, hipStream_t s) {
  if (!S || rows < 1 || n < 1 || rows > 0x7fffffff) return FS_ERR_ARG;
  if (n > 16384) hipLaunchKernelGGL(softmax_rows_big_kernel, dim3((unsigned)rows), dim3(256), 0, s, S, n);
  else hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), (size_t)((n + 3) & ~3) * 4, s, S, n);
  return fs_launch_status();
}
extern "C" int fsraft_softmax_rows_bwd(const float* A, float* dA, int64_t rows, int n, hipStream_t s) {
  if (!A || !dA || rows < 1 || n < 1 || rows > 0x7fffffff) return FS_ERR_ARG;
  if (n > 8192) hipLaunchKernelGGL(softmax_rows_bwd_big_kernel, dim3((unsigned)rows), dim3(256), 0, s, A, dA, n);
  else hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), (size_t)((n + 3) & ~3) * 8, s, A, dA, n);
  return fs_launch_status();
}
// In place: fp32 logits [rows][n] -> softmax probabilities as records [rows][n / 32][32 hi | 32 lo] (n % 32 == 0, n <= 16352:
// the row in dynamic LDS plus the kernel's static reduction words stay within 64 KB, the launch limit without an opt-in -- there
// is no second copy of the logits to fall back to if a launch were refused).
extern "C" int fsraft_softmax_rows_rec(float* S, int64_t rows, int n, hipStream_t s) {
  if (!S || rows < 1 || n < 32 || (n % 32) || n > 16352 || rows > 0x7fffffff || ((uintptr_t)S % 16)) return FS_ERR_ARG;
  hipLaunchKernelGGL(softmax_rows_rec_kernel, dim3((unsigned)rows), dim3(256), (size_t)n * 4, s, S, n);
  return fs_launch_status();
}
// A: records of fsraft_softmax_rows_rec; dA: fp32 gradient in, records of dS out (in place).  n % 32 == 0, n <= 8160 (two rows
// in dynamic LDS + the static reduction words within 64 KB).
extern "C" int fsraft_softmax_rows_bwd_rec(const void* A, float* dA, int64_t rows, int n, const unsigned* ds_amax, hipStream_t s) {
  if (!A || !dA || rows < 1 || n < 32 || (n % 32) || n > 8160 || rows > 0x7fffffff || ((uintptr_t)A % 16) || ((uintptr_t)dA % 16))
    return FS_ERR_ARG;
  hipLaunchKernelGGL(softmax_rows_bwd_rec_kernel, dim3((unsigned)rows), dim3(256), (size_t)n * 8, s, (const char*)A, dA, n, ds_amax);
  return fs_launch_status();
}
extern "C" int fsraft_gma_mix_fwd(const float* x, int ldx, const float* y, int ldy, const float* gamma, float* dst, int ldd,
                                  int64_t M, int C, unsigned* dst_amax, hipStream_t s) {
  if (!x || !y || !gamma || !dst || C % 4 || ldx % 4 || ldy % 4 || ldd % 4 || ((uintptr_t)dst_amax & 3)) return FS_ERR_ARG;
  int gf = grid_for(M * (C / 4));
  if (dst_amax && gf > 1024) gf = 1024;
  hipLaunchKernelGGL(gma_mix_fwd_kernel, dim3(gf), dim3(256), 0, s, x, ldx, y, ldy, gamma, dst, ldd, M, C, dst_amax);
  return fs_launch_status();
}
extern "C" int fsraft_gma_mix_bwd(const float* d, int ldd, const float* y, int ldy, const float* gamma, float* dx, int ldx,
                                  float* dy, int lddy, float* dgamma, int64_t M, int C, unsigned* dx_amax, unsigned* dy_amax, hipStream_t s) {
  if (!d || !y || !gamma || !dx || !dy || !dgamma || C % 4 || ldd % 4 || ldy % 4 || ldx % 4 || lddy % 4 ||
      (((uintptr_t)dx_amax | (uintptr_t)dy_amax) & 3)) return FS_ERR_ARG;
  // one atomic per workgroup on the ONE dgamma address: 3520 of them took ~40 of the kernel's 51 us (they serialise at the
  // memory side); 512 grid-striding workgroups leave 512
  const int grid = grid_for(M * (C / 4));
  hipLaunchKernelGGL(gma_mix_bwd_kernel, dim3(grid > 512 ? 512 : grid), dim3(256), 0, s, d, ldd, y, ldy, gamma, dx, ldx, dy,
                     lddy, dgamma, M, C, dx_amax, dy_amax);
  return fs_launch_status();
}
