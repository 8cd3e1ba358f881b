// The "split" arithmetic of every GEMM-shaped kernel of the library (gfx950): fp16x3.
//
// An fp32 operand x is carried as two fp16 pieces of the SCALED value y = x * s (s a power of two, one per tensor):
//     hi = fp16(y),   lo = fp16(y - hi)            (round to nearest; |y - hi - lo| <= max(2^-22 |y|, 2^-25))
// and a product is evaluated as   a*b ~= (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi) / (s_a * s_b)   on v_mfma_f32_*_f16 with fp32
// accumulation: three matrix instructions per product block -- the cost of the bf16x3 split of rounds 1-5 (fp16 and bf16
// MFMA issue at the same rate) -- but 22 significant bits per operand instead of 16: the dropped a_lo*b_lo term is 2^-22
// relative, products of fp16 pieces are exact in fp32 (11 x 11 bits), so what is left is fp32 accumulation, as in the
// reference's fp32 GEMMs.  fp16 has 5 exponent bits where bf16 had fp32's 8, hence the scale: s maps the tensor's amax word
// into [2^13, 2^14) (fs_scale_of_amax) -- the word may sit up to a factor 2 below the true maximum (fs_amax_commit), so every
// element lands below 2^15 and nothing overflows; every element within 2^-16 of the largest keeps all 22 bits, and smaller ones
// degrade gracefully through fp16's subnormals to an absolute floor of 2^-38 of the largest magnitude (scripts/probes/mfma_f16_denorm.hip: the matrix pipe keeps subnormal inputs).  Scales are exact powers of two and
// are divided out of the fp32 accumulators in the epilogue: no rounding is added by them.
#pragma once

typedef _Float16 p16x8 __attribute__((ext_vector_type(8)));     // eight 16-bit pieces: one MFMA operand fragment
typedef _Float16 p16x4 __attribute__((ext_vector_type(4)));
typedef float fs_f32x16 __attribute__((ext_vector_type(16)));
typedef float fs_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ fs_f32x16 fs_mfma_32x32x16(p16x8 a, p16x8 b, fs_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ fs_f32x4 fs_mfma_16x16x32(p16x8 a, p16x8 b, fs_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// two fp32 -> packed (hi, hi) and (lo, lo) pieces of x * s.  hi: v_pk_mul_f32 + v_cvt_pk_f16_f32 (round to nearest even);
// lo = fp16(x * s - hi) in ONE instruction per element: v_fma_mixlo/mixhi_f16 take the fp32 operands x and s and the fp16
// operand hi as they are, evaluate the fused multiply-add unrounded and round once, to fp16, into the lower / upper half of
// the destination.  Two VALU instructions per element (the bf16 split of rounds 1-5 took 2.5; the compiler's own sequence
// for the same expression -- convert hi back, v_pk_fma_f32, convert -- takes 3: FS_SPLIT_MIX=0).
#ifndef FS_SPLIT_MIX
#define FS_SPLIT_MIX 1
#endif
__device__ __forceinline__ void fs_split2(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 x = {x0, x1};
  const h2 h = __builtin_convertvector(x * s, h2);
  hi = __builtin_bit_cast(unsigned, h);
#if FS_SPLIT_MIX
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(hi));
  lo = l;
#else
  const f2 d = __builtin_elementwise_fma(x, (f2){s, s}, -__builtin_convertvector(h, f2));
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(d, h2));
#endif
}
__device__ __forceinline__ void fs_split4(const float* r, float s, uint2& hi, uint2& lo) {
  fs_split2(r[0], r[1], s, hi.x, lo.x);
  fs_split2(r[2], r[3], s, hi.y, lo.y);
}
// two packed (hi, hi) / (lo, lo) piece pairs back to fp32: (hi + lo) * inv   (inv = 1 / scale)
__device__ __forceinline__ void fs_unsplit2(unsigned hi, unsigned lo, float inv, float& x0, float& x1) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 h = __builtin_bit_cast(h2, hi), l = __builtin_bit_cast(h2, lo);
  x0 = ((float)h[0] + (float)l[0]) * inv;
  x1 = ((float)h[1] + (float)l[1]) * inv;
}
__device__ __forceinline__ void fs_split1(float x, float s, _Float16& hi, _Float16& lo) {
  hi = (_Float16)(x * s);
  lo = (_Float16)__builtin_fmaf(x, s, -(float)hi);
}

// ---- amax words and scales -------------------------------------------------------------------------------------------
// An "amax word" is one unsigned in device memory holding the bit pattern of a magnitude w with max |x| < 2 w over a tensor:
// the exact maximum (fsraft_amax), an upper bound, or what producers raised it to with atomicMax (bit patterns of
// non-negative floats order like the floats; NaN sorts above infinity, so a NaN anywhere makes the scale NaN and the result
// NaN, as fp32 arithmetic would).  Consumers derive the tensor's scale from it.  A NULL word means "the caller vouches
// |x| < 2^15": scale 1.
#define FS_AMAX_UNIT 0x46000000u        // bit pattern of 2^13: the amax that maps to scale 1

__device__ __forceinline__ unsigned fs_abs_bits(float x) { return __builtin_bit_cast(unsigned, x) & 0x7fffffffu; }
__device__ __forceinline__ unsigned fs_amax_load(const unsigned* w) {      // wave-uniform pointer -> scalar load (the word was
  return w ? *w : FS_AMAX_UNIT;                                            // written by an earlier kernel: caches are clean)
}
// scale of a tensor whose word has the bit pattern `amax_bits`: the power of two that maps the word into [2^13, 2^14) and
// hence every element below 2^15; 1 for an all-zero tensor; never above 2^62 (tensors whose largest magnitude is below 2^-49
// keep fewer bits), so that the product of two scales and its reciprocal stay finite normal numbers
__device__ __forceinline__ float fs_scale_of_amax(unsigned amax_bits) {
  if (amax_bits == 0) return 1.0f;
  const int e = (int)(amax_bits >> 23);                 // biased exponent (0: an fp32 subnormal)
  if (e == 255) return __builtin_bit_cast(float, 0x7fc00000u);
  int se = 127 + 13 - (e - 127);                        // biased exponent of the scale
  se = se > 127 + 62 ? 127 + 62 : se;
  return __builtin_bit_cast(float, (unsigned)se << 23);
}
__device__ __forceinline__ unsigned fs_umax(unsigned a, unsigned b) { return a > b ? a : b; }
// exact reciprocal of a power-of-two scale
__device__ __forceinline__ float fs_inv_scale(float s) {
  const unsigned b = __builtin_bit_cast(unsigned, s);
  if ((b & 0x7f800000u) == 0x7f800000u) return s;       // NaN stays NaN
  return __builtin_bit_cast(float, (254u << 23) - b);
}
// ---- raising a word from a producer ----
// Atomics on ONE address serialise at the memory side (~11 ns each: 3520 of them took 40 us of the GMA gamma-gradient kernel, and a
// first version of this file that let every wave of gru_bwd1 issue three turned 19 us into 212).  So: (1) a word is only
// touched when the value's EXPONENT exceeds the word's -- a scale needs the word to within a factor 2, which this keeps (same
// exponent => value < 2 x word); (2) what is written is the value ONE binade up (early samples: two), so that the thousands of
// workgroups behind the first few find a word they do not exceed and skip -- the word may then sit up to 4x above the true
// maximum, which costs two of the ~10 binades of slack the fp16 pieces have; (3) streaming kernels let a handful of waves
// publish a sample after their first loop trip (fs_amax_early), long before the crowd arrives.  The word only grows inside a
// launch, so a stale read can only cause a spare atomic.
__device__ __forceinline__ bool fs_amax_worth(unsigned m, unsigned cur) { return (m >> 23) > (cur >> 23) || (cur == 0u && m != 0u); }
__device__ __forceinline__ unsigned fs_amax_up(unsigned m, unsigned binades) {      // m x 2^binades, never beyond the largest finite exponent
  const unsigned e = m >> 23;
  return (m == 0u || e >= 255u) ? m : (e + binades > 254u ? (254u << 23) | (m & 0x7fffffu) : m + (binades << 23));
}
__device__ __forceinline__ unsigned fs_wave_umax(unsigned m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fs_umax(m, (unsigned)__shfl_xor((int)m, o, 64));
  return m;
}
__device__ __forceinline__ void fs_amax_raise(unsigned* word, unsigned m, unsigned up) {     // one lane
  // (device-scope relaxed load: served by L2, where the atomics land -- not by this CU's L1, which would keep showing the
  //  value it saw first)
  if (fs_amax_worth(m, __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicMax(word, fs_amax_up(m, up));
}
// the workgroup's per-thread maxima `m` (bit patterns) -> the word.  `red`: LDS scratch of at least (threads / 64) words, free
// to use; every thread of the workgroup must call.  At most one atomic per workgroup.
__device__ __forceinline__ void fs_amax_commit(unsigned* word, unsigned m, unsigned* red) {
  m = fs_wave_umax(m);
  const int nw = (int)(blockDim.x * blockDim.y * blockDim.z + 63) >> 6;
  const int tid = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < nw; ++i) m = fs_umax(m, red[i]);
    fs_amax_raise(word, m, 1u);
  }
}
// the same with the look at the word taken EARLIER (fs_amax_peek before the epilogue's stores): the L2 round trip of the look
// -- ~1 us, measured as +1.8 % on the convolution launches when it sat at the very end of the kernel -- then hides behind the
// stores; a look that is a few microseconds old can only cause a spare atomic.
__device__ __forceinline__ unsigned fs_amax_peek(const unsigned* word) {
  const int tid = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
  return (word && tid == 0) ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
__device__ __forceinline__ void fs_amax_commit_peeked(unsigned* word, unsigned m, unsigned* red, unsigned cur) {
  m = fs_wave_umax(m);
  const int nw = (int)(blockDim.x * blockDim.y * blockDim.z + 63) >> 6;
  const int tid = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < nw; ++i) m = fs_umax(m, red[i]);
    if (fs_amax_worth(m, cur)) atomicMax(word, fs_amax_up(m, 1u));
  }
}
// the same without barriers, at most one atomic per WAVE: for kernels whose waves leave early
__device__ __forceinline__ void fs_amax_commit_wave(unsigned* word, unsigned m) {
  m = fs_wave_umax(m);
  if ((threadIdx.x & 63) == 0) fs_amax_raise(word, m, 1u);
}
// an early sample, two binades up, from wave 0 of the first eight workgroups (call once, after the first loop trip; wave-uniform
// control flow around the call)
__device__ __forceinline__ void fs_amax_early(unsigned* word, unsigned m) {
  if (blockIdx.x < 8u && threadIdx.x < 64u && blockIdx.y == 0u && blockIdx.z == 0u) {
    m = fs_wave_umax(m);
    if (threadIdx.x == 0) fs_amax_raise(word, m, 2u);
  }
}
__device__ __forceinline__ unsigned fs_abs_bits4(fs_f32x4 v) {
  return fs_umax(fs_umax(fs_abs_bits(v[0]), fs_abs_bits(v[1])), fs_umax(fs_abs_bits(v[2]), fs_abs_bits(v[3])));
}
