// Warm start of the next frame's flow (a step next to the path):
// pytorch/core/utils/utils.py:26-54 forward_interpolate -- every pixel's flow vector is carried to where it points
// (x + dx, y + dy); vectors landing outside the open image rectangle are dropped; every grid node then takes the flow
// of the NEAREST landed point (scipy.interpolate.griddata(method="nearest"), an exact nearest-neighbour query in
// float64).  The reference does this on the CPU between two frames of an evaluation sequence (evaluate.py:43: a device
// -> host -> KD-tree -> device round trip per frame); here it stays on the device: one thread per grid node scans all
// landed points through LDS tiles, in float64 like the reference, lowest point index winning exact ties.
// At the 1/8-resolution grid the flow lives on (55 x 128 = 7040 nodes) that is 5e7 distance evaluations.
#include "common.hpp"

namespace {

constexpr int WS_TILE = 256;

__global__ __launch_bounds__(WS_TILE) void forward_interpolate_kernel(const float* __restrict__ flow, float* __restrict__ out,
                                                                     int H, int W) {
  __shared__ double px[WS_TILE], py[WS_TILE];
  const int N = H * W;
  const int q = blockIdx.x * WS_TILE + threadIdx.x;
  const double xq = (double)(q % W), yq = (double)(q / W);
  double best = 1.0e300;
  int besti = -1;
  for (int t0 = 0; t0 < N; t0 += WS_TILE) {
    const int i = t0 + threadIdx.x;
    double x1 = 1.0e200, y1 = 1.0e200;            // dropped points: farther than anything real
    if (i < N) {
      const double x = (double)(i % W) + (double)flow[i], y = (double)(i / W) + (double)flow[N + i];
      if (x > 0.0 && x < (double)W && y > 0.0 && y < (double)H) { x1 = x; y1 = y; }
    }
    __syncthreads();
    px[threadIdx.x] = x1; py[threadIdx.x] = y1;
    __syncthreads();
    const int n = N - t0 < WS_TILE ? N - t0 : WS_TILE;
    for (int j = 0; j < n; ++j) {
      const double ddx = px[j] - xq, ddy = py[j] - yq;
      const double d = ddx * ddx + ddy * ddy;
      if (d < best) { best = d; besti = t0 + j; }
    }
  }
  if (q < N) {
    const bool hit = besti >= 0 && best < 1.0e100;
    out[q] = hit ? flow[besti] : 0.f;
    out[N + q] = hit ? flow[N + besti] : 0.f;
  }
}

}  // namespace

// flow, out: [2][H][W] fp32 (dx plane, dy plane).  out may not alias flow.
extern "C" int fsraft_forward_interpolate(const float* flow, float* out, int H, int W, hipStream_t s) {
  if (!flow || !out || flow == out || H < 1 || W < 1 || (int64_t)H * W > (1 << 24)) return FS_ERR_ARG;
  hipLaunchKernelGGL(forward_interpolate_kernel, dim3(ceil_div(H * W, WS_TILE)), dim3(WS_TILE), 0, s, flow, out, H, W);
  return fs_launch_status();
}
