// Layout of the correlation volume pyramid in HBM ("tiled rows"), shared by the build, the lookups, the
// gradient-volume materialiser and the operands of the volume-backward GEMMs.
//
// The reference keeps four tensors [B*N, 1, h_l, w_l] (pytorch/core/corr.py:19-27), row-major per query.  A radius-4
// lookup reads a 10x10 window per level: in a row-major 55x128 slice that is ten 40-byte runs 512 bytes apart, i.e. ~13
// 128-byte lines (1.7 KB) for 400 useful bytes.  Here every query owns ONE row of P floats that holds all levels back
// to back, each level cut into 4x4-cell tiles of 64 contiguous bytes (tiles x-fastest, so two x-neighbours share a
// 128-byte line):
//
//     cell (y, x) of level l  ->  off[l] + ((y >> 2) * tw[l] + (x >> 2)) * 16 + (y & 3) * 4 + (x & 3)
//
// A 10x10 window now touches on average 3.25 x 3.25 tiles (~0.7-0.9 KB), one wave instruction fetches the whole
// 4x4-tile superset of a window (lane = (tile, row of the tile), 16 bytes each), and all four windows of a query come
// from one 38 KB row.  Level sizes are the reference's floor halvings; tile counts are ceilings, so levels carry pad cells:
//   * in the volume V pad cells are never read (lookups mask by the true h_l, w_l) and may hold anything;
//   * in the gradient volume dV and in the pooled target-side operand f2cat they are zero, so the volume-backward
//     GEMMs can contract over whole rows.
// P is a multiple of 32 floats (whole [32 hi | 32 lo] bf16 records for the GEMM operands).
#pragma once
#include "common.hpp"

struct VolLayout {
  int nlev, H, W, P;
  int h[4], w[4], th[4], tw[4], off[4];
};

static inline bool vol_layout_make(int H, int W, int nlev, VolLayout& L) {
  if (nlev < 1 || nlev > 4 || H < 1 || W < 1) return false;
  L.nlev = nlev; L.H = H; L.W = W;
  int h = H, w = W, o = 0;
  for (int l = 0; l < 4; ++l) {
    const bool on = l < nlev;
    if (on && (h < 1 || w < 1)) return false;
    L.h[l] = on ? h : 0; L.w[l] = on ? w : 0;
    L.th[l] = on ? (h + 3) / 4 : 0; L.tw[l] = on ? (w + 3) / 4 : 0;
    L.off[l] = o;
    o += L.th[l] * L.tw[l] * 16;
    h /= 2; w /= 2;
  }
  L.P = (o + 31) / 32 * 32;
  return true;
}

__host__ __device__ __forceinline__ int vol_cell(const VolLayout& L, int l, int y, int x) {
  return L.off[l] + ((y >> 2) * L.tw[l] + (x >> 2)) * 16 + (y & 3) * 4 + (x & 3);
}
