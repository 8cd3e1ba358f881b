// What does ds_read_b64_tr_b16 return?  LDS image img[k][m] = 100*k + m (16-bit), 8 rows x 64 cols.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short img[8 * 64];
  for (int i = threadIdx.x; i < 8 * 64; i += 64) img[i] = (short)(100 * (i / 64) + (i % 64));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const short* a = img + (0 + q) * 64 + 16 * g + 4 * p;     // row k0+q, columns m0+4p.. ; m0 = 16*g
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)a);
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = v[i];
}
int main() {
  short* d; (void)hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int i = 0; i < 4; ++i) printf(" %4d", h[l * 4 + i]); printf("\n"); }
}
