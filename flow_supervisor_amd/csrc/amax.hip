// amax words (split_arith.hpp): a tensor's largest magnitude -- to within [1/2, 4] of it -- as one unsigned word in device memory, from which every
// kernel that feeds the tensor to the matrix pipe derives its power-of-two scale.  Kernels that PRODUCE a tensor raise the
// word in their epilogue (fsraft_conv_desc.dst_amax, the gradient-stage kernels, the lookups); these entry points compute it
// for tensors that come from outside the library (images, feature maps of a PyTorch encoder, parameters, gradients handed in
// by autograd).  The reference has no counterpart: its fp32 GEMMs (pytorch/core/corr.py:52-60, every nn.Conv2d of
// update.py:6-136) need no scales; here they are what lets three fp16 products stand in for one fp32 product.
#include "common.hpp"

namespace {

constexpr int AMAX_JOBS = 32;
struct AmaxJob {
  const float* p;        // element (r, c) at p[r * ld + c]
  int64_t rows, C, ld;   // contiguous array of n floats: rows = 1, C = n
  unsigned* word;        // raised (atomicMax) to the largest |x| bit pattern; several jobs may share a word
};
struct AmaxJobs { AmaxJob j[AMAX_JOBS]; };

__global__ __launch_bounds__(256) void amax_jobs_kernel(const AmaxJobs tab) {
  const AmaxJob& j = tab.j[blockIdx.y];
  __shared__ unsigned red[4];
  unsigned m = 0u;
  const bool v4 = (j.C & 3) == 0 && (j.ld & 3) == 0 && ((uintptr_t)j.p & 15) == 0;
  if (v4) {
    const int64_t c4 = j.C >> 2, total = j.rows * c4;
    // two loads in flight per trip; the first trip of the first workgroups publishes an early sample (split_arith.hpp)
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += 2 * stride) {
      const int64_t r = e / c4, c = (e - r * c4) << 2;
      const int64_t e2 = e + stride < total ? e + stride : e, r2 = e2 / c4, c2 = (e2 - r2 * c4) << 2;
      const f32x4 v = gload4(j.p + r * j.ld + c), w = gload4(j.p + r2 * j.ld + c2);
      m = fs_umax(m, fs_umax(fs_abs_bits4(v), fs_abs_bits4(w)));
      if (e < stride && blockIdx.x < 8u && threadIdx.x < 64u) {
        const unsigned s = fs_wave_umax(m);
        if (threadIdx.x == 0) fs_amax_raise(j.word, s, 2u);
      }
    }
  } else {
    const int64_t total = j.rows * j.C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
      const int64_t r = e / j.C, c = e - r * j.C;
      m = fs_umax(m, fs_abs_bits(gload1(j.p + r * j.ld + c)));
    }
  }
  fs_amax_commit(j.word, m, red);
}

// dst = max(dst, bit pattern of factor * float(src)): a bound that follows from another tensor's word (the gradient volume
// collects at most one unit of bilinear weight per lookup and cell: |dV| <= lookups x max |dout|)
__global__ void amax_scaled_kernel(const unsigned* __restrict__ src, float factor, unsigned* __restrict__ dst) {
  const float v = __builtin_bit_cast(float, *src) * factor;
  const unsigned b = fs_abs_bits(v);
  if (b > *dst) *dst = b;
}

}  // namespace

extern "C" int fsraft_abi_version(void) { return 6; }     // FSRAFT_ABI_VERSION of include/fsraft.h

extern "C" int fsraft_amax_scaled(const unsigned* src, float factor, unsigned* dst, hipStream_t stream) {
  if (!src || !dst || ((uintptr_t)src & 3) || ((uintptr_t)dst & 3) || !(factor >= 0.f)) return FS_ERR_ARG;
  hipLaunchKernelGGL(amax_scaled_kernel, dim3(1), dim3(1), 0, stream, src, factor, dst);
  return fs_launch_status();
}

// One launch per 32 jobs: word[i] = max(word[i], max |x| over job i).  ptrs / rows / C / ld / words: host arrays of n entries.
// Words must be 4-byte aligned device addresses, zero (or holding an earlier bound) on entry.
extern "C" int fsraft_amax_jobs(const float* const* ptrs, const int64_t* rows, const int64_t* C, const int64_t* ld,
                                unsigned* const* words, int n, hipStream_t stream) {
  if (n < 0 || (n && (!ptrs || !rows || !C || !ld || !words))) return FS_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (!ptrs[i] || !words[i] || ((uintptr_t)words[i] & 3) || rows[i] < 0 || C[i] < 0 || ld[i] < C[i] * (rows[i] > 1)) return FS_ERR_ARG;
  for (int i0 = 0; i0 < n; i0 += AMAX_JOBS) {
    AmaxJobs tab{};
    const int k = n - i0 < AMAX_JOBS ? n - i0 : AMAX_JOBS;
    int64_t most = 0;
    for (int i = 0; i < k; ++i) {
      tab.j[i] = AmaxJob{ptrs[i0 + i], rows[i0 + i], C[i0 + i], ld[i0 + i], words[i0 + i]};
      const int64_t tot = rows[i0 + i] * C[i0 + i];
      most = tot > most ? tot : most;
    }
    // ~16 float4 per thread; at most 512 workgroups per job (each may cost one atomic on the job's word)
    int64_t blocks = (most / 4 + 256 * 16 - 1) / (256 * 16);
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(amax_jobs_kernel, dim3((unsigned)blocks, k), dim3(256), 0, stream, tab);
    const int rc = fs_launch_status();
    if (rc) return rc;
  }
  return FS_OK;
}

extern "C" int fsraft_amax(const float* x, int64_t rows, int64_t C, int64_t ld, unsigned* word, hipStream_t stream) {
  return fsraft_amax_jobs(&x, &rows, &C, &ld, &word, 1, stream);
}
