// Small HBM-bound kernels around the update-block GEMMs: layout changes at the API
// boundary (NCHW <-> channels-last), the 7x7 flow-conv im2col / col2im pair, activation
// and ConvGRU gate derivatives, and column sums for bias gradients.
// Reference ops they stand in for: torch.cat / permute / relu / sigmoid / tanh backward
// in pytorch/core/update.py:16-136 as executed by autograd.
#include "common.hpp"

namespace {

// dst[(b*HW + p)*ld + coff + c] (=|+=) src[(b*C + c)*HW + p]
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int C, int HW, int ld, int coff, int accumulate) {
  __shared__ float t[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, p = p0 + tx;
    const float v = src[((int64_t)b * C + (c < C ? c : C - 1)) * HW + (p < HW ? p : HW - 1)];    // (clamped: the four loads go out together)
    t[ty + 8 * j][tx] = (c < C && p < HW) ? v : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = p0 + ty + 8 * j, c = c0 + tx;
    if (c < C && p < HW) {
      float* d = dst + ((int64_t)b * HW + p) * ld + coff + c;
      const float v = t[tx][ty + 8 * j];
      *d = accumulate ? *d + v : v;
    }
  }
}

// dst[(b*C + c)*HW + p] (=|+=) src[(b*HW + p)*ld + coff + c]
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int C, int HW, int ld, int coff, int accumulate) {
  __shared__ float t[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = p0 + ty + 8 * j, c = c0 + tx;
    const float v = src[((int64_t)b * HW + (p < HW ? p : HW - 1)) * ld + coff + (c < C ? c : C - 1)];
    t[ty + 8 * j][tx] = (c < C && p < HW) ? v : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, p = p0 + tx;
    if (c < C && p < HW) {
      float* d = dst + ((int64_t)b * C + c) * HW + p;
      const float v = t[tx][ty + 8 * j];
      *d = accumulate ? *d + v : v;
    }
  }
}

// Space-to-depth by 2 of a channels-last tensor, as a pixel permutation: dst[b][y/2][x/2][(y%2)*2 + x%2][c] = src[b][y][x][c]
// (INVERSE: the other way).  A stride-2 convolution over src is a stride-1 convolution with 2x2 taps over dst viewed as
// [B][H/2][W/2][4C].  One 16-byte chunk per thread and step; both sides move whole pixels (C*4 contiguous bytes).
template <bool INVERSE>
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int C) {
  const int c4n = C >> 2;
  const int64_t total = (int64_t)B * H * W * c4n;
  const int h2 = H >> 1, w2 = W >> 1;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(e % c4n);
    const int64_t q = e / c4n;                       // pixel index in the block-major (space-to-depth) order
    const int sub = (int)(q & 3);
    const int64_t blk = q >> 2;
    const int x2 = (int)(blk % w2), y2 = (int)((blk / w2) % h2);
    const int64_t b = blk / ((int64_t)w2 * h2);
    const int64_t p = (b * H + 2 * y2 + (sub >> 1)) * W + 2 * x2 + (sub & 1);      // raster pixel index
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    if (INVERSE) d4[p * c4n + c4] = s4[q * c4n + c4];
    else d4[q * c4n + c4] = s4[p * c4n + c4];
  }
}

// cols[m][ci*49 + ky*7 + kx] = flow[b, ci, y+ky-3, x+kx-3] (zero outside); cols pitch = ld (>= 100), pad cols zeroed
// (loads are unconditional with a clamped index and a select: `ok ? p[i] : 0` compiles to a branch around the load, one
//  round trip per tap -- col2im7 was bound by exactly that, 22 us per launch for 14 MB)
__global__ __launch_bounds__(256) void im2col7_kernel(const float* __restrict__ flow, int64_t bs, int64_t cs, int64_t ps,
                                                      float* __restrict__ cols, int ld, int B, int H, int W,
                                                      unsigned* __restrict__ amax) {
  const int64_t total = (int64_t)B * H * W * ld;
  unsigned mx = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int k = (int)(e % ld);
    const int64_t m = e / ld;
    float v = 0.f;
    if (k < 98) {                                  // (here the branchy form measured faster: 13.8 vs 16.8 us -- a quarter of the columns is padding)
      const int ci = k / 49, t = k % 49;
      const int x = (int)(m % W), y = (int)((m / W) % H);
      const int64_t b = m / ((int64_t)W * H);
      const int yy = y + t / 7 - 3, xx = x + t % 7 - 3;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = flow[b * bs + ci * cs + ((int64_t)yy * W + xx) * ps];
    }
    cols[e] = v;
    mx = fs_umax(mx, fs_abs_bits(v));
  }
  if (amax) fs_amax_commit_wave(amax, mx);
}

// the same, four columns per thread and one 16-byte store (ld % 4 == 0: always; the scalar kernel above wrote 4 bytes per thread:
// 13.8 us per launch for 11 MB at 4 x 55x128, on the forward chain of every iteration)
__global__ __launch_bounds__(256) void im2col7_v4_kernel(const float* __restrict__ flow, int64_t bs, int64_t cs, int64_t ps,
                                                         float* __restrict__ cols, int ld, int B, int H, int W,
                                                         unsigned* __restrict__ amax) {      // amax (nullable): word of cols, raised
  const int l4 = ld >> 2;
  const int64_t total = (int64_t)B * H * W * l4;
  unsigned mx = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int k0 = (int)(e % l4) * 4;
    const int64_t m = e / l4;
    const int x = (int)(m % W), y = (int)((m / W) % H);
    const int64_t b = m / ((int64_t)W * H);
    const float* fb = flow + b * bs;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + i;
      const int kk = k < 98 ? k : 0;
      const int ci = kk / 49, t = kk % 49;
      const int yy = y + t / 7 - 3, xx = x + t % 7 - 3;
      const bool in = k < 98 && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const float f = fb[ci * cs + (in ? ((int64_t)yy * W + xx) * ps : 0)];          // (unconditional load of a clamped address)
      v[i] = in ? f : 0.f;
    }
    gstore4(cols + m * ld + k0, v);
    mx = fs_umax(mx, fs_abs_bits4(v));
    if (amax && e < (int64_t)gridDim.x * 256) fs_amax_early(amax, mx);      // (first trip)
  }
  __shared__ unsigned red[4];
  if (amax) fs_amax_commit(amax, mx, red);
}

// adjoint: dflow[b, ci, y, x] (+)= sum_t dcols[(b, y-(ky-3), x-(kx-3))][ci*49 + t]   (dflow contiguous [B,2,H,W])
__global__ __launch_bounds__(256) void col2im7_kernel(const float* __restrict__ dcols, int ld, float* __restrict__ dflow,
                                                      int B, int H, int W, int accumulate) {
  const int64_t total = (int64_t)B * 2 * H * W;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int x = (int)(e % W), y = (int)((e / W) % H);
  const int ci = (int)((e / ((int64_t)W * H)) % 2);
  const int64_t b = e / ((int64_t)2 * W * H);
  float v[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) {                  // all 49 loads in flight, then the sum
    const int yy = y - (t / 7 - 3), xx = x - (t % 7 - 3);
    const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
    v[t] = dcols[((b * H + yc) * W + xc) * ld + ci * 49 + t];
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 49; ++t) {
    const int yy = y - (t / 7 - 3), xx = x - (t % 7 - 3);
    s += (yy >= 0 && yy < H && xx >= 0 && xx < W) ? v[t] : 0.f;
  }
  const float old = dflow[e];
  dflow[e] = accumulate ? old + s : s;
}

// 2-channel strided tensor -> channels [coff, coff+2) of a channels-last buffer, and the reverse (accumulating)
__global__ __launch_bounds__(256) void flow_to_nhwc_kernel(const float* __restrict__ flow, int64_t bs, int64_t cs,
                                                           int64_t ps, float* __restrict__ dst, int ld, int coff,
                                                           int64_t M, int HW, unsigned* __restrict__ amax) {
  unsigned mx = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < 2 * M; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e >> 1; const int c = (int)(e & 1);
    const int64_t b = m / HW, p = m % HW;
    const float v = flow[b * bs + c * cs + p * ps];
    dst[m * ld + coff + c] = v;
    mx = fs_umax(mx, fs_abs_bits(v));
  }
  __shared__ unsigned red[4];
  if (amax) fs_amax_commit(amax, mx, red);      // (nullable) word of dst, raised
}
__global__ __launch_bounds__(256) void nhwc_to_flow_kernel(const float* __restrict__ src, int ld, int coff,
                                                           float* __restrict__ dflow, int64_t M, int HW, int accumulate) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= 2 * M) return;
  const int64_t m = e >> 1; const int c = (int)(e & 1);
  const int64_t b = m / HW, p = m % HW;
  float* d = dflow + (b * 2 + c) * HW + p;
  const float v = src[m * ld + coff + c];
  *d = accumulate ? *d + v : v;
}

// g[m][c] *= (y[m][c] > 0)     (ReLU backward, in place on the incoming gradient)
__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ g, int ldg, const float* __restrict__ y,
                                                       int ldy, int64_t M, int C) {
  const int c4n = (C + 3) / 4;
  const int64_t total = M * c4n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / c4n; const int c = (int)(e % c4n) * 4;
    f32x4 gv = *reinterpret_cast<f32x4*>(g + m * ldg + c);
    const f32x4 yv = *reinterpret_cast<const f32x4*>(y + m * ldy + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) gv[i] = yv[i] > 0.f ? gv[i] : 0.f;
    *reinterpret_cast<f32x4*>(g + m * ldg + c) = gv;
  }
}

// ConvGRU gate backward, stage 1 (h' = (1-z) h + z q):
//   dzr[:, 0:hid] = dh' * (q - h) * z (1-z)     (pre-activation grad of the z conv)
//   dq_pre        = dh' * z * (1 - q^2)
//   dh            = dh' * (1 - z)                 (direct path; conv paths are added later)
__global__ __launch_bounds__(256) void gru_bwd1_kernel(const float* __restrict__ dhn, const float* __restrict__ z,
                                                       const float* __restrict__ q, const float* __restrict__ h,
                                                       float* __restrict__ dzr, int ldzr, float* __restrict__ dq,
                                                       float* __restrict__ dh, float* __restrict__ dzr_sum,
                                                       float* __restrict__ dq_sum, int64_t M, int hid,
                                                       unsigned* am_dzr, unsigned* am_dq, unsigned* am_dh) {   // (nullable) words of dzr / dq / dh, raised
  const int64_t total = M * hid;
  unsigned m0 = 0u, m1 = 0u, m2 = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / hid; const int c = (int)(e % hid);
    const float g = dhn[e], zz = z[e], qq = q[e], hh = h[e];
    const float dz = g * (qq - hh) * zz * (1.f - zz), dqv = g * zz * (1.f - qq * qq), dhv = g * (1.f - zz);
    dzr[m * ldzr + c] = dz;
    dq[e] = dqv;
    dh[e] = dhv;
    m0 = fs_umax(m0, fs_abs_bits(dz)); m1 = fs_umax(m1, fs_abs_bits(dqv)); m2 = fs_umax(m2, fs_abs_bits(dhv));
    if (dzr_sum) dzr_sum[m * ldzr + c] += dz;       // running sums over the iterations of a step (context part's backward)
    if (dq_sum) dq_sum[e] += dqv;
  }
  if (am_dzr) fs_amax_commit_wave(am_dzr, m0);
  if (am_dq) fs_amax_commit_wave(am_dq, m1);
  if (am_dh) fs_amax_commit_wave(am_dh, m2);
}

// the same on four channels per thread (hid, ldzr multiples of 4, 16-byte aligned pointers); dhn2 (nullable): a second summand of
// dh' -- the hidden-state gradient arriving from the next iteration, added here instead of by a launch of its own
__global__ __launch_bounds__(256) void gru_bwd1_v4_kernel(const float* __restrict__ dhn, const float* __restrict__ dhn2,
                                                          const float* __restrict__ z, const float* __restrict__ q,
                                                          const float* __restrict__ h, float* __restrict__ dzr, int ldzr,
                                                          float* __restrict__ dq, float* __restrict__ dh, float* __restrict__ dzr_sum,
                                                          float* __restrict__ dq_sum, int64_t M, int hid,
                                                          unsigned* am_dzr, unsigned* am_dq, unsigned* am_dh) {
  const int h4 = hid >> 2;
  const int64_t total = M * h4;
  unsigned m0 = 0u, m1 = 0u, m2 = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / h4; const int c = (int)(e % h4) * 4;
    const int64_t o = m * hid + c, oz = m * ldzr + c;
    f32x4 g = gload4(dhn + o);
    const f32x4 zz = gload4(z + o), qq = gload4(q + o), hh = gload4(h + o);
    if (dhn2) g += gload4(dhn2 + o);
    f32x4 dz, dqv, dhv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dz[i] = g[i] * (qq[i] - hh[i]) * zz[i] * (1.f - zz[i]);
      dqv[i] = g[i] * zz[i] * (1.f - qq[i] * qq[i]);
      dhv[i] = g[i] * (1.f - zz[i]);
    }
    gstore4(dzr + oz, dz);
    gstore4(dq + o, dqv);
    gstore4(dh + o, dhv);
    m0 = fs_umax(m0, fs_abs_bits4(dz)); m1 = fs_umax(m1, fs_abs_bits4(dqv)); m2 = fs_umax(m2, fs_abs_bits4(dhv));
    if (dzr_sum) gstore4(dzr_sum + oz, gload4(dzr_sum + oz) + dz);
    if (dq_sum) gstore4(dq_sum + o, gload4(dq_sum + o) + dqv);
    if (e < (int64_t)gridDim.x * 256) {          // (first trip: early samples)
      if (am_dzr) fs_amax_early(am_dzr, am_dzr == am_dq ? fs_umax(m0, m1) : m0);
      if (am_dq && am_dq != am_dzr) fs_amax_early(am_dq, m1);
      if (am_dh) fs_amax_early(am_dh, m2);
    }
  }
  // (dzr and dq may share one word -- they feed convolutions of one layer group -- and then cost one commit)
  __shared__ unsigned red[4];
  if (am_dzr) fs_amax_commit(am_dzr, am_dzr == am_dq ? fs_umax(m0, m1) : m0, red);
  if (am_dq && am_dq != am_dzr) fs_amax_commit(am_dq, m1, red);
  if (am_dh) fs_amax_commit(am_dh, m2, red);
}

__global__ __launch_bounds__(256) void gru_bwd2_v4_kernel(const float* __restrict__ drh, const float* __restrict__ r,
                                                          const float* __restrict__ h, float* __restrict__ dzr, int ldzr,
                                                          float* __restrict__ dh, float* __restrict__ dzr_sum, int64_t M, int hid,
                                                          unsigned* am_dzr, unsigned* am_dh) {      // (nullable) words of dzr / dh, raised
  const int h4 = hid >> 2;
  const int64_t total = M * h4;
  unsigned m0 = 0u, m2 = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / h4; const int c = (int)(e % h4) * 4;
    const int64_t o = m * hid + c, oz = m * ldzr + hid + c;
    const f32x4 g = gload4(drh + o), rr = gload4(r + o), hh = gload4(h + o);
    f32x4 dhv = gload4(dh + o), dr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dr[i] = g[i] * hh[i] * rr[i] * (1.f - rr[i]);
      dhv[i] += g[i] * rr[i];
    }
    gstore4(dzr + oz, dr);
    gstore4(dh + o, dhv);
    m0 = fs_umax(m0, fs_abs_bits4(dr)); m2 = fs_umax(m2, fs_abs_bits4(dhv));
    if (dzr_sum) gstore4(dzr_sum + oz, gload4(dzr_sum + oz) + dr);
    if (e < (int64_t)gridDim.x * 256) {
      if (am_dzr) fs_amax_early(am_dzr, m0);
      if (am_dh) fs_amax_early(am_dh, m2);
    }
  }
  __shared__ unsigned red[4];
  if (am_dzr) fs_amax_commit(am_dzr, m0, red);
  if (am_dh) fs_amax_commit(am_dh, m2, red);
}

// stage 2 (input of the q conv was r*h):  dzr[:, hid:2hid] = d(rh) * h * r (1-r);   dh += d(rh) * r
__global__ __launch_bounds__(256) void gru_bwd2_kernel(const float* __restrict__ drh, const float* __restrict__ r,
                                                       const float* __restrict__ h, float* __restrict__ dzr, int ldzr,
                                                       float* __restrict__ dh, float* __restrict__ dzr_sum, int64_t M, int hid,
                                                       unsigned* am_dzr, unsigned* am_dh) {
  const int64_t total = M * hid;
  unsigned m0 = 0u, m2 = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / hid; const int c = (int)(e % hid);
    const float g = drh[e], rr = r[e];
    const float dr = g * h[e] * rr * (1.f - rr), dhv = dh[e] + g * rr;
    dzr[m * ldzr + hid + c] = dr;
    dh[e] = dhv;
    m0 = fs_umax(m0, fs_abs_bits(dr)); m2 = fs_umax(m2, fs_abs_bits(dhv));
    if (dzr_sum) dzr_sum[m * ldzr + hid + c] += dr;
  }
  if (am_dzr) fs_amax_commit_wave(am_dzr, m0);
  if (am_dh) fs_amax_commit_wave(am_dh, m2);
}

// out[c] += scale * sum_m x[m][c]      (bias gradients).  grid.x covers channels in 64s, grid.y splits rows.
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, int ld, int64_t M, int C,
                                                      float* __restrict__ out, float scale) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int64_t rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  const int64_t r1 = r0 + rows_per < M ? r0 + rows_per : M;
  float s = 0.f;
  if (c < C)
    for (int64_t m = r0 + w; m < r1; m += 4) s += x[m * ld + c];
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < C) atomicAdd(out + c, scale * (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// y = a*x + b*y over n floats (n % 4 == 0 not required)
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, float a, float b,
                                                    int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    y[e] = a * x[e] + (b == 0.f ? 0.f : b * y[e]);
}

// out = (accumulate ? out : 0) + src[0] + src[1] + ... + src[N - 1], added in that order, four floats per thread; all N loads of a
// trip are requested before the first add.  (The running sums of the GRU gate gradients over the iterations of a step used to be
// read-modify-write passes inside gru_bwd1 / gru_bwd2 -- 86 MB per iteration and GRU pass at the bench shape; one pass over the
// kept gradients at the end of the step moves 40 % fewer bytes.)
constexpr int SUMN_MAX = 16;
struct SumNArgs { const float* src[SUMN_MAX]; };
template <int N>
__global__ __launch_bounds__(256) void sum_n_kernel(SumNArgs a, float* __restrict__ out, int64_t n4, int accumulate) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
    f32x4 v[N];
#pragma unroll
    for (int t = 0; t < N; ++t) v[t] = gload4(a.src[t] + 4 * e);
    f32x4 acc = accumulate ? gload4(out + 4 * e) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < N; ++t) acc += v[t];
    gstore4(out + 4 * e, acc);
  }
}

inline int grid_for(int64_t work) { int64_t g = (work + 255) / 256; return (int)(g < 1 ? 1 : g > 16384 ? 16384 : g); }

}  // namespace

// src [B][H][W][C] -> dst [B][H/2][W/2][2][2][C] (inverse != 0: the other way; H, W are the full-resolution sizes).
extern "C" int fsraft_space_to_depth2(const float* src, float* dst, int B, int H, int W, int C, int inverse, hipStream_t s) {
  if (!src || !dst || src == dst || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 4 || (C & 3)) return FS_ERR_ARG;
  int64_t blocks = ((int64_t)B * H * W * (C / 4) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (inverse) hipLaunchKernelGGL((s2d_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, src, dst, B, H, W, C);
  else hipLaunchKernelGGL((s2d_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, src, dst, B, H, W, C);
  return fs_launch_status();
}

extern "C" int fsraft_nchw_to_nhwc(const float* src, float* dst, int B, int C, int HW, int ld, int coff, int accumulate,
                                   hipStream_t s) {
  if (!src || !dst || B < 1 || C < 1 || HW < 1) return FS_ERR_ARG;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ceil_div(HW, 32), ceil_div(C, 32), B), dim3(256), 0, s, src, dst, C, HW, ld, coff, accumulate);
  return fs_launch_status();
}
extern "C" int fsraft_nhwc_to_nchw(const float* src, float* dst, int B, int C, int HW, int ld, int coff, int accumulate,
                                   hipStream_t s) {
  if (!src || !dst || B < 1 || C < 1 || HW < 1) return FS_ERR_ARG;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ceil_div(HW, 32), ceil_div(C, 32), B), dim3(256), 0, s, src, dst, C, HW, ld, coff, accumulate);
  return fs_launch_status();
}
extern "C" int fsraft_im2col7(const float* flow, int64_t bs, int64_t cs, int64_t ps, float* cols, int ld, int B, int H,
                              int W, unsigned* cols_amax, hipStream_t s) {
  if (!flow || !cols || ld < 100 || ld % 4 || ((uintptr_t)cols_amax & 3)) return FS_ERR_ARG;
  if (((uintptr_t)cols & 15) == 0) {
    int g = grid_for((int64_t)B * H * W * (ld / 4));
    if (cols_amax && g > 512) g = 512;           // (a word to raise: fewer, longer workgroups -- each may cost one atomic)
    hipLaunchKernelGGL(im2col7_v4_kernel, dim3(g), dim3(256), 0, s, flow, bs, cs, ps, cols, ld, B, H, W, cols_amax);
  }
  else
    hipLaunchKernelGGL(im2col7_kernel, dim3(grid_for((int64_t)B * H * W * ld)), dim3(256), 0, s, flow, bs, cs, ps, cols, ld, B, H, W, cols_amax);
  return fs_launch_status();
}
extern "C" int fsraft_col2im7(const float* dcols, int ld, float* dflow, int B, int H, int W, int accumulate, hipStream_t s) {
  if (!dcols || !dflow) return FS_ERR_ARG;
  const int64_t total = (int64_t)B * 2 * H * W;
  hipLaunchKernelGGL(col2im7_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dcols, ld, dflow, B, H, W, accumulate);
  return fs_launch_status();
}
extern "C" int fsraft_flow_to_nhwc(const float* flow, int64_t bs, int64_t cs, int64_t ps, float* dst, int ld, int coff,
                                   int B, int HW, unsigned* dst_amax, hipStream_t s) {
  if (!flow || !dst || ((uintptr_t)dst_amax & 3)) return FS_ERR_ARG;
  const int64_t M = (int64_t)B * HW;
  // (with a word to raise: few workgroups -- the kernel is tiny, and every workgroup may cost one atomic on the word)
  const int64_t wg = (2 * M + 255) / 256;
  hipLaunchKernelGGL(flow_to_nhwc_kernel, dim3((unsigned)(dst_amax && wg > 64 ? 64 : wg)), dim3(256), 0, s, flow, bs, cs, ps, dst, ld, coff, M, HW, dst_amax);
  return fs_launch_status();
}
extern "C" int fsraft_nhwc_to_flow(const float* src, int ld, int coff, float* dflow, int B, int HW, int accumulate, hipStream_t s) {
  if (!src || !dflow) return FS_ERR_ARG;
  const int64_t M = (int64_t)B * HW;
  hipLaunchKernelGGL(nhwc_to_flow_kernel, dim3((unsigned)((2 * M + 255) / 256)), dim3(256), 0, s, src, ld, coff, dflow, M, HW, accumulate);
  return fs_launch_status();
}
extern "C" int fsraft_relu_bwd(float* g, int ldg, const float* y, int ldy, int64_t M, int C, hipStream_t s) {
  if (!g || !y || ldg % 4 || ldy % 4) return FS_ERR_ARG;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(M * ((C + 3) / 4))), dim3(256), 0, s, g, ldg, y, ldy, M, C);
  return fs_launch_status();
}
namespace {
bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
}
extern "C" int fsraft_gru_bwd1(const float* dhn, const float* dhn2, const float* z, const float* q, const float* h, float* dzr, int ldzr,
                               float* dq, float* dh, float* dzr_sum, float* dq_sum, int64_t M, int hid, unsigned* dzr_amax, unsigned* dq_amax,
                               unsigned* dh_amax, hipStream_t s) {
  if (!dhn || !z || !q || !h || !dzr || !dq || !dh || (((uintptr_t)dzr_amax | (uintptr_t)dq_amax | (uintptr_t)dh_amax) & 3)) return FS_ERR_ARG;
  const bool v4 = hid % 4 == 0 && ldzr % 4 == 0 && al16(dhn) && al16(dhn2) && al16(z) && al16(q) && al16(h) && al16(dzr) && al16(dq) &&
                  al16(dh) && al16(dzr_sum) && al16(dq_sum);
  if (!v4 && dhn2) return FS_ERR_ARG;            // (the second summand exists for the vectorised kernel only)
  const bool words = dzr_amax || dq_amax || dh_amax;
  int g4 = grid_for(M * (hid / 4));
  if (words && g4 > 1024) g4 = 1024;
  if (v4) hipLaunchKernelGGL(gru_bwd1_v4_kernel, dim3(g4), dim3(256), 0, s, dhn, dhn2, z, q, h, dzr, ldzr, dq, dh,
                             dzr_sum, dq_sum, M, hid, dzr_amax, dq_amax, dh_amax);
  else hipLaunchKernelGGL(gru_bwd1_kernel, dim3(grid_for(M * hid)), dim3(256), 0, s, dhn, z, q, h, dzr, ldzr, dq, dh, dzr_sum, dq_sum, M, hid,
                          dzr_amax, dq_amax, dh_amax);
  return fs_launch_status();
}
extern "C" int fsraft_gru_bwd2(const float* drh, const float* r, const float* h, float* dzr, int ldzr, float* dh,
                               float* dzr_sum, int64_t M, int hid, unsigned* dzr_amax, unsigned* dh_amax, hipStream_t s) {
  if (!drh || !r || !h || !dzr || !dh || (((uintptr_t)dzr_amax | (uintptr_t)dh_amax) & 3)) return FS_ERR_ARG;
  int g4 = grid_for(M * (hid / 4));
  if ((dzr_amax || dh_amax) && g4 > 1024) g4 = 1024;
  if (hid % 4 == 0 && ldzr % 4 == 0 && al16(drh) && al16(r) && al16(h) && al16(dzr) && al16(dh) && al16(dzr_sum))
    hipLaunchKernelGGL(gru_bwd2_v4_kernel, dim3(g4), dim3(256), 0, s, drh, r, h, dzr, ldzr, dh, dzr_sum, M, hid, dzr_amax, dh_amax);
  else hipLaunchKernelGGL(gru_bwd2_kernel, dim3(grid_for(M * hid)), dim3(256), 0, s, drh, r, h, dzr, ldzr, dh, dzr_sum, M, hid, dzr_amax, dh_amax);
  return fs_launch_status();
}
extern "C" int fsraft_col_sum(const float* x, int ld, int64_t M, int C, float* out, float scale, hipStream_t s) {
  if (!x || !out) return FS_ERR_ARG;
  int ysplit = (int)(M / 128); if (ysplit < 1) ysplit = 1; if (ysplit > 2048) ysplit = 2048;
  hipLaunchKernelGGL(col_sum_kernel, dim3(ceil_div(C, 64), ysplit), dim3(256), 0, s, x, ld, M, C, out, scale);
  return fs_launch_status();
}
// out[0..count) = (accumulate ? out : 0) + sum of the n tensors src[t][0..count), added in list order (count % 4 == 0, 16-byte
// aligned pointers, n >= 1: lists longer than 16 take several launches)
extern "C" int fsraft_sum_n(const float* const* src, int n, float* out, int64_t count, int accumulate, hipStream_t s) {
  if (!src || !out || n < 1 || count < 4 || (count & 3) || ((uintptr_t)out & 15)) return FS_ERR_ARG;
  for (int t = 0; t < n; ++t)
    if (!src[t] || ((uintptr_t)src[t] & 15)) return FS_ERR_ARG;
  const int64_t n4 = count / 4;
  for (int base = 0; base < n; base += SUMN_MAX) {
    const int m = n - base < SUMN_MAX ? n - base : SUMN_MAX;
    SumNArgs a{};
    for (int t = 0; t < SUMN_MAX; ++t) a.src[t] = src[base + (t < m ? t : 0)];
    const int acc = (accumulate || base > 0) ? 1 : 0;
    const dim3 g(grid_for(n4)), b(256);
    // (exact list lengths: a padded slot would be a wasted read of a whole tensor)
    switch (m) {
#define FS_SUMN(K) case K: hipLaunchKernelGGL((sum_n_kernel<K>), g, b, 0, s, a, out, n4, acc); break;
      FS_SUMN(1) FS_SUMN(2) FS_SUMN(3) FS_SUMN(4) FS_SUMN(5) FS_SUMN(6) FS_SUMN(7) FS_SUMN(8) FS_SUMN(9) FS_SUMN(10) FS_SUMN(11) FS_SUMN(12)
      FS_SUMN(13) FS_SUMN(14) FS_SUMN(15) FS_SUMN(16)
#undef FS_SUMN
    }
    const int rc = fs_launch_status();
    if (rc) return rc;
  }
  return FS_OK;
}
extern "C" int fsraft_axpby(const float* x, float* y, float a, float b, int64_t n, hipStream_t s) {
  if (!x || !y) return FS_ERR_ARG;
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, a, b, n);
  return fs_launch_status();
}
