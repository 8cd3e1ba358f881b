// LDS read rates on one CU under 8 waves: ds_read_b64_tr_b16 with the fragment addressing of the TN cores (row pitch varied),
// plain ds_read_b64 / ds_read_b128 with linear addresses.  Prints SIMD cycles per wave instruction, CU wide.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 256, NR = 8;
template <int KIND>
__global__ __launch_bounds__(512) void k(int pitch, long long* cyc, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 32768; i += 512) ((int*)lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lh = lane >> 5, gb = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;
  int base;
  if (KIND == 0) base = (8 * lh + q) * pitch + (16 * gb + 4 * p4) * 2 + (wave & 1) * 64;
  else if (KIND == 1) base = lane * 8 + wave * 512;
  else base = lane * 16 + wave * 1024;
  i4 acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
    const char* p = lds + base + (it & 3) * 16 * pitch;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (KIND == 0) {
        s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(p + (r & 3) * 4 * pitch + (r >> 2) * 16 * pitch));
        i2 w = __builtin_bit_cast(i2, v); acc[0] ^= w[0]; acc[1] ^= w[1];
      } else if (KIND == 1) {
        i2 w = *(const i2*)(p + r * 4096); acc[0] ^= w[0]; acc[1] ^= w[1];
      } else {
        i4 w = *(const i4*)(p + r * 8192); acc ^= w;
      }
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc[0] == 0x12345 && acc[1] == 77 && acc[2] == 1) sink[0] = acc[3];
}
template <int KIND> void run(const char* name, int pitch) {
  long long* d; int* s; (void)hipMalloc(&d, 256 * 8); (void)hipMalloc(&s, 4);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 131072, 0, pitch, d, s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 131072, 0, pitch, d, s);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
  // clock64 = s_memtime at 100 MHz on this part; report both the counter and the event time
  printf("%-28s pitch %4d: %8.0f ticks, %7.1f us -> %6.2f ns per wave instruction (CU wide, 8 waves)\n", name, pitch, c, ms * 1e3,
         ms * 1e6 / (ITERS * NR * 8.0));
  (void)hipFree(d); (void)hipFree(s);
}
int main() {
  for (int p : {192, 320, 144, 160, 136, 132, 256, 128, 208, 576, 64, 96}) run<0>("ds_read_b64_tr_b16", p);
  run<1>("ds_read_b64 linear", 64);
  run<2>("ds_read_b128 linear", 64);
}
