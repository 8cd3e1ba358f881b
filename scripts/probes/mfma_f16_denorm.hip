// Probe: does v_mfma_f32_32x32x16_f16 on gfx950 keep fp16 subnormal INPUTS (the lo pieces of the fp16x3 split are
// subnormal whenever |x * scale| < 2^-3)?  Prints the product of a subnormal A element with B = 2^10.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_f16_denorm.hip -o /tmp/denorm && /tmp/denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  // lane l holds row (l & 31) of A / column (l & 31) of B, k = 8 * (l >> 5) .. +7: put one non-zero at k = 0 of every row / column
  if (threadIdx.x < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  const float avals[] = {1.0f, 6.103515625e-05f /*2^-14 min normal*/, 3.0517578125e-05f /*2^-15*/, 9.5367431640625e-07f /*2^-20*/, 5.9604644775390625e-08f /*2^-24 min subnormal*/};
  for (float av : avals) {
    k<<<1, 64>>>(av, 1024.f, d);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a = 2^%6.1f  b = 2^10  mfma product = %.10g  expected %.10g  %s\n", log2f(av), h, av * 1024.f, h == av * 1024.f ? "kept" : "FLUSHED/other");
  }
  // both subnormal-ish: product 2^-20 * 2^-4
  k<<<1, 64>>>(9.5367431640625e-07f, 0.0625f, d);
  float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("a = 2^-20 b = 2^-4  product = %.10g expected %.10g\n", h, 9.5367431640625e-07f * 0.0625f);
  return 0;
}
