// Pyramid lookup and its backward on the tiled-row volume layout (corr_layout.hpp); rows a3 of SURVEY.md section 8,
// reference: pytorch/core/corr.py:29-50 + core/utils/utils.py:57-71 (forward), grid_sampler_2d_backward w.r.t. the
// volume (backward).
//
// Forward: ONE WAVE PER QUERY.  Per level a single wave instruction fetches the 4x4-tile superset of the (2r+2)^2 window
// (lane = (tile, row of the tile): 16 bytes each, four lanes = one 64-byte tile), masked down to the tiles / rows the
// window really touches; the 16x16-cell region goes to LDS and the (2r+1)^2 bilinear blends of the level are computed
// from it with the channel on the lane, so every store instruction writes 256 contiguous bytes of the channels-last
// output.  Algorithmic HBM bytes per query: L (2r+2)^2 4 read + 8 coords + L (2r+1)^2 4 written.
//
// Backward: the reference materialises a dense zero gradient of the whole volume on each of the 4 x 12 grid_sample
// backward calls; round 1 of this library kept one dense gradient pyramid and read-modify-wrote the windows of every
// lookup into it (12 passes + a zero fill over ~1 GB).  Here a lookup's backward does NOTHING but keep (coords, dOut):
// the window gradients are a pure function of those 324 + 2 floats per query.  When autograd reaches the volume build,
// fsraft_corr_dvol_build writes the gradient volume ONCE: per query, all lookups of the step are accumulated in LDS
// (zero-initialised there) and the finished row -- pad cells zero -- leaves as 16-byte stores.  HBM traffic:
// read n * 324 * 4 + write P * 4 bytes per query, no zero fill, no read-modify-write.
//
// The pooling chain has no backward pass of its own any more: with V_l[i, cell] = f1[i] . mean_{cell}(f2),
//     dF1[c][i]  = s * sum_p F2cat[c][p] * dV[i][p]          F2cat = pooled f2 in the row layout (fsraft_corr_f2cat)
//     dF2cat[p][c] = s * sum_i dV[i][p] * f1[i][c]            then dF2 = sum_l 4^-l unpool_l (fsraft_corr_dfmap2)
// i.e. the un-pool runs on the 7 MB feature gradient instead of the 1 GB volume gradient.
#include "corr_layout.hpp"
#include "gemm_rec.hpp"

namespace {

struct Coords {
  const float* p;
  int64_t bs, cs, ps;
};
// The query position is either given (coords) or coords = pixel grid + flow: grid_w > 0 names the image width and the
// tensor holds the FLOW, so that the RAFT loop never materialises coords1 = coords0 + flow (raft.py:121-131).
__device__ __forceinline__ void query_xy(const Coords& c, int b, int pix, int grid_w, float& cx, float& cy) {
  cx = gload1(c.p + b * c.bs + pix * c.ps);
  cy = gload1(c.p + b * c.bs + c.cs + pix * c.ps);
  if (grid_w > 0) { cx += (float)(pix % grid_w); cy += (float)(pix / grid_w); }
}

template <int R>
struct TL {
  static constexpr int N1 = 2 * R + 1, WIN = 2 * R + 2, N2 = N1 * N1;
  static constexpr int RP = 20;             // region row pitch (floats): 16-byte aligned rows, <= 3-way bank conflicts in the blend
  static constexpr int REGION = 16 * RP;    // one level's 16x16-cell region
  static constexpr int ROUNDS = (N2 + 63) / 64;
};

struct LevelQ {      // one (query, level): integer window origin and bilinear weights
  int wx0, wy0;
  float fx, fy;
};

__device__ __forceinline__ LevelQ level_query(float cx, float cy, int l, int R) {
  const float s = 1.0f / (float)(1 << l);
  cx *= s; cy *= s;
  // anything this far out has an all-zero window; the clamp keeps floor->int defined (also for NaN)
  cx = (cx > -30000.f && cx < 30000.f) ? cx : -30000.f;
  cy = (cy > -30000.f && cy < 30000.f) ? cy : -30000.f;
  const float flx = floorf(cx), fly = floorf(cy);
  return LevelQ{(int)flx - R, (int)fly - R, cx - flx, cy - fly};
}

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving them across
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// A wave walks through QW consecutive queries.  The chain of one query is coords -> window origin -> region loads -> LDS ->
// blends -> stores, every link waiting for the one before, and 32 resident waves per CU do not hide it (measured: 2.5 TB/s
// algorithmic with the queries handled one after the other).  So the chain is software-pipelined over the queries of a wave:
// all QW coordinate pairs are fetched first (wave-uniform addresses), and the region loads of query k + 1 are issued before
// the blends of query k, i.e. two queries' regions (up to 8 KB per wave) are in flight while one is being consumed.
// The loads are buffer loads (base = the query's row, wave-uniform): a lane whose 16 bytes the window does not touch gets bit
// 31 in its offset, fails the range check and reads zeros -- no branch around the load, so the loop body is straight-line
// code and the compiler can count the loads in flight (`s_waitcnt vmcnt(N)`, not 0).
struct LevelGeo { int off, tw, h, w; };        // per-level constants of the layout, in SGPRs

// (only the loaded tile rows are carried from the issue to the blends; the window origin, the fractions and the pad mask of a level
//  are recomputed from the query position at the blend -- 36 registers per query in flight held the kernel at four waves per SIMD,
//  i.e. two rounds of waves for the 27 per CU a four-pair lookup needs)
template <int R>
struct LookupLoad {
  f32x4 v[4];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* row, unsigned bytes) {
  const uint64_t u = reinterpret_cast<uint64_t>(row);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uint64_t)hi << 32 | lo), 0, bytes, 0x00020000);
}

template <int R, int AUX = 0>      // AUX: cache policy of the window loads (0 plain, 2 nt, 16 sc1, 18 both: fsraft_set_lookup_policy)
__device__ __forceinline__ void lookup_issue(LookupLoad<R>& ld, const float* __restrict__ row, unsigned row_bytes, const LevelGeo (&g)[4],
                                             int nlev, float cx, float cy, int tsx, int tsy, int r) {
  using S = TL<R>;
  const __amdgpu_buffer_rsrc_t rs = row_rsrc(row, row_bytes);
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const LevelQ lq = level_query(cx, cy, l, R);
    const int tx = (lq.wx0 >> 2) + tsx, ty = (lq.wy0 >> 2) + tsy;     // >> on negatives = floor division
    const int y = 4 * ty + r, x = 4 * tx;
    const bool need = l < nlev && tx >= 0 && tx < g[l].tw && ty >= 0 && y < g[l].h && y >= lq.wy0 && y < lq.wy0 + S::WIN &&
                      x + 3 >= lq.wx0 && x < lq.wx0 + S::WIN;
    const unsigned voff = need ? (unsigned)(g[l].off + (ty * g[l].tw + tx) * 16 + r * 4) * 4u : 0x80000000u;
    ld.v[l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, AUX));
  }
}

// GATHER (measurement only, fsraft_set_lookup_policy(100)): the same window loads with the same masks and the same look-ahead, but
// nothing done with them -- no LDS staging, no blends, no output: what the memory system delivers for this access pattern.
template <int R, int QW, int AUX = 0, bool GATHER = false>
__global__ __launch_bounds__(256) void lookup_tiled_fwd_kernel(const float* __restrict__ vol, VolLayout L, Coords co,
                                                               float* __restrict__ out, int64_t nq, int HW, int grid_w,
                                                               unsigned* __restrict__ out_amax) {     // (nullable) word of `out`, raised
  using S = TL<R>;
  __shared__ __attribute__((aligned(16))) float region[4][4][S::REGION];
  unsigned amx = 0u;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nlev = L.nlev, CH = nlev * S::N2;
  LevelGeo g[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) g[l] = LevelGeo{L.off[l], L.tw[l], L.h[l], L.w[l]};
  const unsigned row_bytes = (unsigned)L.P * 4u;
  float* reg = &region[wave][0][0];
  const int tsx = (lane >> 2) & 3, tsy = lane >> 4, r = lane & 3;
  int choff[S::ROUNDS];            // channel (i, j) of a level handled by this lane in round k -> offset inside the region
#pragma unroll
  for (int k = 0; k < S::ROUNDS; ++k) {
    const int kk = lane + 64 * k;
    choff[k] = kk < S::N2 ? (kk % S::N1) * S::RP + kk / S::N1 : 0;      // i = kk / N1 (x offset, slow), j = kk % N1 (y offset)
  }
  const unsigned q0 = (blockIdx.x * 4u + (unsigned)wave) * QW;          // wave-uniform; nq < 2^31 (checked by the host)
  if (q0 >= (unsigned)nq) return;
  const int nqw = (int)((unsigned)nq - q0 < (unsigned)QW ? (unsigned)nq - q0 : (unsigned)QW);
  float cxs[QW], cys[QW];
#pragma unroll
  for (int qq = 0; qq < QW; ++qq) {
    const unsigned q = q0 + (qq < nqw ? qq : 0);
    query_xy(co, (int)(q / (unsigned)HW), (int)(q % (unsigned)HW), grid_w, cxs[qq], cys[qq]);
  }
  LookupLoad<R> cur, nxt;
  lookup_issue<R, AUX>(cur, vol + (int64_t)q0 * L.P, row_bytes, g, nlev, cxs[0], cys[0], tsx, tsy, r);
#pragma unroll
  for (int qq = 0; qq < QW; ++qq) {
    if (qq >= nqw) break;
    const unsigned q = q0 + qq;
    // the next query's regions are requested before this one's are consumed (the last valid query is simply requested again)
    const int qn = qq + 1 < nqw ? qq + 1 : qq;
    if (qq + 1 < QW) lookup_issue<R, AUX>(nxt, vol + (int64_t)(q0 + qn) * L.P, row_bytes, g, nlev, cxs[qn], cys[qn], tsx, tsy, r);
    if (GATHER) {
      float sink = 0.f;
#pragma unroll
      for (int l = 0; l < 4; ++l) sink += cur.v[l][0] + cur.v[l][1] + cur.v[l][2] + cur.v[l][3];
      if (sink == 1.2345e38f) gstore1(out + (int64_t)q * CH, sink);      // (never: keeps the loads and their waits alive)
      if (qq + 1 < QW) cur = nxt;
      continue;
    }
    LevelQ lqs[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) lqs[l] = level_query(cxs[qq], cys[qq], l, R);
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (l < nlev) {
        f32x4 v = cur.v[l];
        const int xr = g[l].w - 4 * ((lqs[l].wx0 >> 2) + tsx);      // true width minus the x of this lane's first cell: cells at or beyond it are pad
#pragma unroll
        for (int c = 1; c < 4; ++c) v[c] = c < xr ? v[c] : 0.f;
        *reinterpret_cast<f32x4*>(reg + l * S::REGION + (4 * tsy + r) * S::RP + 4 * tsx) = v;
      }
    wave_lds_sync();
    float* o = out + (int64_t)q * CH;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l >= nlev) continue;
      const float* base = reg + l * S::REGION + (lqs[l].wy0 & 3) * S::RP + (lqs[l].wx0 & 3);
      const float fx = lqs[l].fx, fy = lqs[l].fy;
#pragma unroll
      for (int k = 0; k < S::ROUNDS; ++k) {
        const float* p = base + choff[k];
        const float top = p[0] + fx * (p[1] - p[0]);
        const float bot = p[S::RP] + fx * (p[S::RP + 1] - p[S::RP]);
        const float res = top + fy * (bot - top);
        if (lane + 64 * k < S::N2) { gstore1(o + l * S::N2 + lane + 64 * k, res); amx = fs_umax(amx, fs_abs_bits(res)); }
      }
    }
    wave_lds_sync();
    if (qq + 1 < QW) cur = nxt;
    if (out_amax && qq == 0) fs_amax_early(out_amax, amx);        // (a sample from the first queries, long before the crowd)
  }
  if (out_amax) fs_amax_commit_wave(out_amax, amx);
}

// ---------------------------------------------------------------------------------------------------------------
// Gradient volume from the stashed lookups of a step.
constexpr int DV_MAXN = 16;        // lookups per launch (more: further launches with accumulate = 1)
constexpr int DV_SEG = 8192;       // floats of a row held in LDS at a time

struct DvolArgs {
  const float* dout[DV_MAXN];      // [B,H,W,CH] channels-last gradient of each lookup's output
  Coords co[DV_MAXN];
  int n;
  const unsigned* amax;            // records: the word the gradient rows are split with (a bound of |dV|: the lookups' windows
                                   // overlap, a cell collects at most one unit of bilinear weight per lookup); NULL: scale 1
};

// One workgroup = one (query, job): a job is a run of whole pyramid levels [l0, l1] whose cells fit the LDS segment
// (level 0 alone; levels 1..3 together), so two launches cover a row.  REC: the run leaves as records ([32 bf16 hi | 32
// bf16 lo] per 32 cells, gemm_rec.hpp) -- the operand format of the two volume-backward GEMMs, which then stage it by
// LDS-DMA without converting.  CLIP: the run is longer than the segment (very large images) and is processed in pieces.
// the finished gradient rows are written once and read (by the two volume-backward GEMMs) only after 1 GB more has gone by:
// policy 2 stores them `nt` so that they do not push the step's other working sets out of L2 (fsraft_set_dvol_policy)
__device__ __forceinline__ void dvstore4(void* p, f32x4 v, int policy) {
  if (policy == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else if (policy == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else gstore4(p, v);
}
template <int R, bool REC, bool CLIP>
__global__ __launch_bounds__(256) void corr_dvol_kernel(DvolArgs a, VolLayout L, float* __restrict__ dvol, int HW, int accumulate, int l0,
                                                        int l1, int grid_w, int64_t q0, int policy, const unsigned* __restrict__ qlist) {
  using S = TL<R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // qlist (nullable): [0] = count, [1 ..] = query numbers relative to q0 that corr_dvol_box_kernel left for this kernel
  // (their lookups spread further than its box); workgroup i then takes list slot i instead of query i
  // (one workgroup per list slot, surplus workgroups leave at once: a loop over the list inside the kernel tripled its
  //  register count -- 49 -> 158 VGPRs, 7 -> 3 waves per SIMD -- and cost 35 % on the plain route)
  if (qlist && blockIdx.x >= qlist[0]) return;
  const unsigned qrel = qlist ? qlist[1 + blockIdx.x] : blockIdx.x;
  const float rec_s = REC ? fs_scale_of_amax(fs_amax_load(a.amax)) : 1.0f, rec_inv = fs_inv_scale(rec_s);
  float* seg = smem;                                   // [min(run, DV_SEG)]
  const int nlev = L.nlev, CH = nlev * S::N2, n = a.n, nl = l1 - l0 + 1;
  const int rbeg = L.off[l0], rend = (l1 + 1 < nlev) ? L.off[l1 + 1] : L.P;       // the run, in floats of the row
  const int seglen = CLIP ? DV_SEG : rend - rbeg;
  const int GC = nl * S::N2;                           // channels of dOut this job needs: [l0 * N2, (l1 + 1) * N2)
  float* g = smem + seglen;                            // [n][GC]
  LevelQ* qi = reinterpret_cast<LevelQ*>(g + ((n * GC + 3) & ~3));   // [n][4]
  const int64_t q = q0 + qrel;                         // rows of dvol are numbered from q0 (a chunk of queries per call)
  const int b = (int)(q / HW), pix = (int)(q % HW);
  // dOut slices of the n lookups -> LDS; the lookup index is wave-uniform (pointer from the kernarg table by scalar loads)
  // and ALL loads are in flight before the first LDS store: one memory latency per workgroup, not one per lookup
  {
    float v[DV_MAXN][2];
#pragma unroll
    for (int t = 0; t < DV_MAXN; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (t < n && c < GC) v[t][k] = gload1(a.dout[t] + q * CH + l0 * S::N2 + c);
      }
#pragma unroll
    for (int t = 0; t < DV_MAXN; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (t < n && c < GC) g[t * GC + c] = v[t][k];
      }
  }
  if (threadIdx.x < n * 4) {
    const int t = threadIdx.x >> 2, l = threadIdx.x & 3;
    float cx, cy;
    query_xy(a.co[t], b, pix, grid_w, cx, cy);
    qi[threadIdx.x] = level_query(cx, cy, l, R);
  }
  float* row = dvol + (int64_t)qrel * L.P;
  for (int s0 = rbeg; s0 < rend; s0 += seglen) {
    const int len = min(seglen, rend - s0);
    __syncthreads();
    if (REC && accumulate) {
      for (int e = threadIdx.x * 8; e < len; e += 2048) {        // hi + lo back to fp32 (exact to ~2^-22 relative)
        const char* rp = reinterpret_cast<const char*>(row + s0) + (e >> 5) * 128 + (e & 31) * 2;
        const u32x4 h = __builtin_bit_cast(u32x4, gload4(rp)), l = __builtin_bit_cast(u32x4, gload4(rp + 64));
#pragma unroll
        for (int i = 0; i < 4; ++i) fs_unsplit2(h[i], l[i], rec_inv, seg[e + 2 * i], seg[e + 2 * i + 1]);
      }
    } else {
      for (int e = threadIdx.x * 4; e < len; e += 1024)
        *reinterpret_cast<f32x4*>(seg + e) = accumulate ? gload4(row + s0 + e) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    // The lookups are applied ONE AFTER THE OTHER, a barrier apart: inside one lookup the cells of its windows are
    // distinct, so the accumulation is a plain LDS read-add-write.  (LDS float atomics were tried first -- all lookups in
    // parallel, ds_add_f32 -- and ran at about one lane per clock: 540 of the kernel's 800 us.)  One thread per window
    // cell (level, wy, wx): 4 taps of the (2r+1)^2 gradient, bilinear weights of the lookup.
    // what does not depend on the lookup is decoded once per thread: its (at most two) window cells, their level's layout
    // fields, the four gradient taps and which of them exist
    constexpr int KC = (4 * S::WIN * S::WIN + 255) / 256;
    const int ncell = nl * S::WIN * S::WIN;
    int c_wy[KC], c_wx[KC], c_l[KC], c_base[KC], c_tw[KC], c_h[KC], c_w[KC], c_g[KC];
    float c_m[KC][4];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int c = threadIdx.x + 256 * k;
      const int lr = c / (S::WIN * S::WIN), cell = c % (S::WIN * S::WIN), wy = cell / S::WIN, wx = cell % S::WIN;
      const int l = l0 + lr;
      // (the level is per-lane: a dynamic index into the by-value layout would be a dependent vector load from the
      // kernarg segment per field -- select instead)
      c_base[k] = (l == 0 ? L.off[0] : l == 1 ? L.off[1] : l == 2 ? L.off[2] : L.off[3]) - s0;
      c_tw[k] = l == 0 ? L.tw[0] : l == 1 ? L.tw[1] : l == 2 ? L.tw[2] : L.tw[3];
      c_h[k] = c < ncell ? (l == 0 ? L.h[0] : l == 1 ? L.h[1] : l == 2 ? L.h[2] : L.h[3]) : 0;      // 0 rows: never in range
      c_w[k] = l == 0 ? L.w[0] : l == 1 ? L.w[1] : l == 2 ? L.w[2] : L.w[3];
      c_wy[k] = wy; c_wx[k] = wx; c_l[k] = l;
      // window cell (wy, wx) is tap (ay, ax) of output (j = wy - ay, i = wx - ax); taps outside the (2r+1)^2 gradient get weight 0
      const int i0 = wx >= 1 ? wx - 1 : 0, j0 = wy >= 1 ? wy - 1 : 0;
      c_g[k] = lr * S::N2 + i0 * S::N1 + j0;           // tap (i0, j0); tap i1 / j1 = + N1 / + 1 floats where both taps exist
      c_m[k][0] = wx < S::N1 ? 1.f : 0.f; c_m[k][1] = wx >= 1 ? 1.f : 0.f;
      c_m[k][2] = wy < S::N1 ? 1.f : 0.f; c_m[k][3] = wy >= 1 ? 1.f : 0.f;
    }
    // The lookups are applied ONE AFTER THE OTHER, a barrier apart: inside one lookup the cells of its windows are
    // distinct, so the accumulation is a plain LDS read-add-write.  (LDS float atomics -- all lookups in parallel,
    // ds_add_f32 -- were tried first and were slower.)
    for (int t = 0; t < n; ++t) {
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const LevelQ v = qi[t * 4 + c_l[k]];
        const int gy = v.wy0 + c_wy[k], gx = v.wx0 + c_wx[k];
        const int o = c_base[k] + ((gy >> 2) * c_tw[k] + (gx >> 2)) * 16 + (gy & 3) * 4 + (gx & 3);
        bool ok = (unsigned)gy < (unsigned)c_h[k] && (unsigned)gx < (unsigned)c_w[k];
        if (CLIP) ok = ok && o >= 0 && o < len;
        if (ok) {
          const float* gp = g + t * GC + c_g[k];
          // (a tap that does not exist has weight 0 and re-reads the other one)
          const int di = (c_m[k][0] != 0.f && c_m[k][1] != 0.f) ? S::N1 : 0, dj = (c_m[k][2] != 0.f && c_m[k][3] != 0.f) ? 1 : 0;
          const float wx1 = c_m[k][0] * (1.f - v.fx), wx0 = c_m[k][1] * v.fx;
          const float wy1 = c_m[k][2] * (1.f - v.fy), wy0 = c_m[k][3] * v.fy;
          seg[o] += wy1 * (wx1 * gp[di + dj] + wx0 * gp[dj]) + wy0 * (wx1 * gp[di] + wx0 * gp[0]);
        }
      }
      __syncthreads();
    }
    if (REC) {
      for (int e = threadIdx.x * 8; e < len; e += 2048) {        // (level sections are multiples of 16 cells; runs start on 32)
        uint2 h0, l0s, h1, l1s;
        rec_split4(seg + e, h0, l0s, rec_s);
        rec_split4(seg + e + 4, h1, l1s, rec_s);
        char* rp = reinterpret_cast<char*>(row + s0) + (e >> 5) * 128 + (e & 31) * 2;
        dvstore4(rp, __builtin_bit_cast(f32x4, u32x4{h0.x, h0.y, h1.x, h1.y}), policy);
        dvstore4(rp + 64, __builtin_bit_cast(f32x4, u32x4{l0s.x, l0s.y, l1s.x, l1s.y}), policy);
      }
    } else {
      for (int e = threadIdx.x * 4; e < len; e += 1024) dvstore4(row + s0 + e, *reinterpret_cast<const f32x4*>(seg + e), policy);
    }
  }
}

// ---- the same gradient rows without the whole row segment in LDS ---------------------------------------------------------
// corr_dvol_kernel parks a level-0 row segment (28 KB at 55x128) in LDS although the twelve lookups of a query touch a few
// hundred cells of it: five queries per CU are in flight, each a chain of [loads | zero | 12 x (read-add-write, barrier) |
// convert, store], and the kernel runs at 2-3 TB/s of its own traffic (544 us for 1.5 GB).  ONE WAVE per query that keeps, per
// level, only the bounding box of the lookups' windows does better: round 3's corr_dvol_box_kernel (lane = window cell, the
// gradient slices and a 24x24 box per level in LDS: 394-402 us in the step) and, replacing it, round 4's corr_dvol_sep_kernel
// below.  A query whose lookups spread further than the box at some level (a flow that jumped between iterations) is appended
// to a work list that corr_dvol_kernel then walks.
// ---- round 4: the window gradients built SEPARABLY in registers ----------------------------------------------------------
// corr_dvol_box_kernel (round 3) was instruction-bound (394 us for 285 MB of traffic; PMC, scripts/dvol_pmc.sh: 4800 vector and
// 1070 LDS instructions per query, 620 LDS bank-conflict cycles, three waves per SIMD): per query it runs 4 levels x 12 lookups
// x 2 rounds of "four LDS reads of gradient taps, one LDS read-add-write of the box, ~40 VALU instructions" for 19 K useful
// multiply-adds.  Here the 16 lanes of a DPP row own one LEVEL and a lane owns one window COLUMN: lane (l, i) holds the N1
// gradients G[i][0 .. N1-1] of lookup t (consecutive floats of dOut: channel = i * N1 + j) in registers,
//     T[j]   = (1 - fx) G_i[j] + fx G_{i-1}[j]          one row_shr:1 DPP move + two VALU per j   (x direction: the lane to the left)
//     W[wy] += (1 - fy) T[wy] + fy T[wy - 1]            two VALU per window row                  (y direction: registers)
// so a lookup costs ~60 VALU instructions for all four levels at once and no LDS traffic; the dOut slices never pass through
// LDS.  W is kept in registers while consecutive lookups have the same integer window origin at a level (after the first
// iterations the flow moves by fractions of a cell: a handful of distinct origins per step at level 0, one or two at the
// coarser levels) and is added into the level's box -- one LDS read-add-write per window row -- only when the origin changes
// and once at the end.  The boxes of the coarser levels are smaller (a flow that moves <= 14 cells at level 0 moves <= 8 / 4 / 2
// cells further up): 5 KB of LDS per wave, eight workgroups per CU.  The section writer decides per trip, on the scalar unit,
// whether the box can be touched at all; trips that cannot store zeros without looking at it.
// Box of level l: `side` cells of window origins + windows fit; stored tile-aligned (origin floored to a multiple of 4 cells) with
// pitch = rows = side + 3 rounded up to 4, so that every 4x4 tile the box overlaps is two aligned 16-byte LDS reads.
__host__ __device__ constexpr int dvs_side(int l) { return l == 0 ? 24 : l == 1 ? 18 : l == 2 ? 14 : 12; }
__host__ __device__ constexpr int dvs_pitch(int l) { return (dvs_side(l) + 3 + 3) / 4 * 4; }              // 28, 24, 20, 16
__host__ __device__ constexpr int dvs_base(int l) { return l == 0 ? 0 : dvs_base(l - 1) + dvs_pitch(l - 1) * dvs_pitch(l - 1); }
constexpr int DVS_TOT = dvs_base(3) + dvs_pitch(3) * dvs_pitch(3);      // 2016 floats per wave

template <int R, bool REC>
__global__ __launch_bounds__(256) void corr_dvol_sep_kernel(DvolArgs a, VolLayout L, float* __restrict__ dvol, int HW, int grid_w, int64_t q0,
                                                            unsigned nq, unsigned* __restrict__ qlist, int policy,
                                                            const unsigned* __restrict__ wmask, int wm_hw) {
  using S = TL<R>;
  constexpr int N1 = S::N1, WIN = S::WIN, N2 = S::N2;
  const float rec_s = REC ? fs_scale_of_amax(fs_amax_load(a.amax)) : 1.0f;
  __shared__ __attribute__((aligned(16))) float box_s[4][DVS_TOT];           // [wave][level boxes back to back]
  __shared__ __attribute__((aligned(16))) LevelQ lq_s[4][4][DV_MAXN];        // [wave][level][lookup]
  __shared__ float cxy[4][DV_MAXN][2];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __shared__ unsigned wl_s[6];                           // work-list hand-over: [wave] = 1 if its query does not fit, [4] = base slot
  const unsigned qraw = blockIdx.x * 4u + (unsigned)wave;
  const bool active = qraw < nq;                         // (wave-uniform; a surplus wave repeats the last query up to the hand-over)
  const unsigned qrel = active ? qraw : nq - 1u;
  const int64_t q = q0 + qrel;
  const int b = (int)(q / HW), pix = (int)(q % HW);
  const int n = a.n, nlev = L.nlev, CH = nlev * N2;
  const int lv = lane >> 4, i = lane & 15;               // this lane's level and window column (or lookup, in the set-up)
  const bool lvl_on = lv < nlev;
  // this lane's N1 gradients of lookup t: dOut[t][q][lv * N2 + i * N1 + j], 36 (28) consecutive bytes, 4-byte aligned.  Lanes right
  // of the gradient (i >= N1) and rows of levels that do not exist re-read a valid slice; the first get a zero x weight, the
  // rest are never handed in.
  const bool g_on = lvl_on && i < N1;
  const int64_t goff = q * CH + (lvl_on ? lv : 0) * N2 + (i < N1 ? i : N1 - 1) * N1;
  // PF lookups' slices in flight per wave (with one lookup of look-ahead the wave waited one memory latency per lookup: 335 us,
  // 92 % of its cycles in s_waitcnt); the first PF are requested before anything else -- their addresses depend on the query
  // number alone.  Every loaded element passes through an empty asm before its first use: left alone, the SLP vectoriser pairs
  // the nine floats for v_pk_* instructions and re-packs them into register pairs right behind the loads, which puts the wait
  // for a slice directly after its issue.
  constexpr int PF = 6;
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
  struct GS { f32x4 a, b; float c; };
  GS G[PF];
  auto loadg = [&](int t, GS& g) {
    // (unconditional: a load inside a wave-uniform branch makes the compiler's wait counting fall back to vmcnt(0) at the join;
    //  slots beyond the n-th lookup re-read lookup 0's slice, which the caches hold)
    const float* p = (a.dout[t] ? a.dout[t] : a.dout[0]) + goff;
    g.a = *(const FS_GLOBAL f32x4u*)p;
    if constexpr (N1 == 9) {
      g.b = *(const FS_GLOBAL f32x4u*)(p + 4);
      g.c = gload1(p + 8);
    } else {
      const f32x3u b3 = *(const FS_GLOBAL f32x3u*)(p + 4);
      g.b = f32x4{b3[0], b3[1], b3[2], 0.f};
      g.c = 0.f;
    }
  };
  auto gval = [&](const GS& g, int j) -> float { return j < 4 ? g.a[j] : (N1 == 9 ? (j < 8 ? g.b[j - 4] : g.c) : g.b[j - 4]); };
  // the row's record mask in registers (lane w holds word w; rows of up to 64 words = 65536 cells): one load per wave
  const unsigned* wm = wmask ? wmask + ((int64_t)(qrel / (unsigned)wm_hw) * ((wm_hw + 31) >> 5) + ((qrel % (unsigned)wm_hw) >> 5)) *
                                            (((L.P >> 5) + 31) >> 5) : nullptr;
  const int wm_words = ((L.P >> 5) + 31) >> 5;
  const bool wm_regs = wm && wm_words <= 64;
  const unsigned wm_mine = (wm_regs && lane < wm_words) ? wm[lane] : 0u;
  const int hl = lv == 0 ? L.h[0] : lv == 1 ? L.h[1] : lv == 2 ? L.h[2] : L.h[3];
  const int wl = lv == 0 ? L.w[0] : lv == 1 ? L.w[1] : lv == 2 ? L.w[2] : L.w[3];
  const int side = dvs_side(lv), pitch = dvs_pitch(lv);
  // the n query positions: every load issued before the first LDS store (one memory latency per wave, not one per lookup)
  float cxr[DV_MAXN], cyr[DV_MAXN];
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < DV_MAXN; ++t)
      if (t < n) query_xy(a.co[t], b, pix, grid_w, cxr[t], cyr[t]);
  }
#pragma unroll
  for (int k = 0; k < PF; ++k) loadg(k, G[k]);
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < DV_MAXN; ++t)
      if (t < n) { cxy[wave][t][0] = cxr[t]; cxy[wave][t][1] = cyr[t]; }
  }
  {   // zero the boxes (16-byte stores)
    f32x4* bz = reinterpret_cast<f32x4*>(&box_s[wave][0]);
#pragma unroll
    for (int k = 0; k < (DVS_TOT / 4 + 63) / 64; ++k)
      if (lane + 64 * k < DVS_TOT / 4) bz[lane + 64 * k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  wave_lds_sync();
  // every (level, lookup) pair's window origin and fractions in ONE pass: lane = (level, lookup); then the bounding box of a
  // level's origins by a reduction inside its 16-lane row.  A query either fits at every level or goes to the work list.
  int ox, oy, uw, uh;                                    // this lane's level: first window origin, extent of the windows' union
  {
    const int tt = i < n ? i : 0;
    const LevelQ v = level_query(cxy[wave][tt][0], cxy[wave][tt][1], lv, R);
    lq_s[wave][lv][i] = v;
    int mnx = v.wx0, mny = v.wy0, mxx = v.wx0, mxy = v.wy0;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
      mnx = min(mnx, __shfl_xor(mnx, d, 64)); mny = min(mny, __shfl_xor(mny, d, 64));
      mxx = max(mxx, __shfl_xor(mxx, d, 64)); mxy = max(mxy, __shfl_xor(mxy, d, 64));
    }
    ox = mnx; oy = mny; uw = mxx - mnx + WIN; uh = mxy - mny + WIN;
    const bool fits = !lvl_on || (uw <= side && uh <= side);
    const bool unfit = __builtin_amdgcn_ballot_w64(!fits) != 0;
    // one list append per WORKGROUP (a counter word takes ~90 atomics per microsecond: 28 K single appends are 0.3 ms)
    if (lane == 0) wl_s[wave] = (active && unfit) ? 1u : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned c = wl_s[0] + wl_s[1] + wl_s[2] + wl_s[3];
      wl_s[4] = c ? atomicAdd(qlist, c) : 0u;
    }
    __syncthreads();
    if (!active) return;
    if (unfit) {
      unsigned slot = wl_s[4];
      for (int w = 0; w < wave; ++w) slot += wl_s[w];
      if (lane == 0) qlist[1 + slot] = qrel;
      return;
    }
  }
  const int bax = ox & ~3, bay = oy & ~3;                 // the box's origin, tile-aligned
  float* box = &box_s[wave][lv == 0 ? dvs_base(0) : lv == 1 ? dvs_base(1) : lv == 2 ? dvs_base(2) : dvs_base(3)];
  float acc[WIN];
#pragma unroll
  for (int wy = 0; wy < WIN; ++wy) acc[wy] = 0.f;
  int cur_x0 = 0, cur_y0 = 0;
  bool have = false;                                      // acc holds lookups with window origin (cur_x0, cur_y0)
  auto flush = [&](bool rows) {                           // rows: this lane's level hands its window in
    float* bp = box + (cur_y0 - bay) * pitch + (cur_x0 + i - bax);
    const bool inside = cur_x0 >= 0 && cur_x0 + WIN <= wl && cur_y0 >= 0 && cur_y0 + WIN <= hl;
    if (!__builtin_amdgcn_ballot_w64(rows && !inside)) {  // no window of this hand-in is clipped by the image (the usual case)
      if (rows && i < WIN) {
#pragma unroll
        for (int wy = 0; wy < WIN; ++wy) bp[wy * pitch] += acc[wy];
      }
    } else {
      const bool xok = rows && i < WIN && (unsigned)(cur_x0 + i) < (unsigned)wl;
#pragma unroll
      for (int wy = 0; wy < WIN; ++wy)
        if (xok && (unsigned)(cur_y0 + wy) < (unsigned)hl) bp[wy * pitch] += acc[wy];
    }
#pragma unroll
    for (int wy = 0; wy < WIN; ++wy) acc[wy] = rows ? 0.f : acc[wy];
  };
#pragma unroll
  for (int t = 0; t < DV_MAXN; ++t) {
    if (t < n) {
      GS& g = G[t % PF];
      const LevelQ v = lq_s[wave][lv][t];
      const bool moved = lvl_on && have && (v.wx0 != cur_x0 || v.wy0 != cur_y0);
      if (__builtin_amdgcn_ballot_w64(moved)) {
        flush(moved);
        wave_lds_sync();
      }
      cur_x0 = v.wx0; cur_y0 = v.wy0; have = true;
      const float fx = v.fx, fy = v.fy, gx1 = g_on ? 1.f - fx : 0.f, gy1 = 1.f - fy;
      float T[N1];
#pragma unroll
      for (int j = 0; j < N1; ++j) {
        float gj = gval(g, j);
        asm volatile("" : "+v"(gj));
        const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gj), 0x111, 0xf, 0xf, true));
        T[j] = gx1 * gj + fx * left;
      }
#pragma unroll
      for (int wy = 0; wy < WIN; ++wy) {
        if (wy < N1) acc[wy] += gy1 * T[wy];
        if (wy >= 1) acc[wy] += fy * T[wy - 1];
      }
    }
    if (t + PF < DV_MAXN) loadg(t + PF, G[t % PF]);
  }
  flush(lvl_on);
  wave_lds_sync();

  char* rowb = reinterpret_cast<char*>(dvol + (int64_t)qrel * L.P);
  // ---- the row.  Pass 1: zeros, 1 KB (eight records, or 256 floats) per wave store, only where the list GEMMs read; the mask
  // bits of a trip are a scalar: trips without a flagged record are skipped on the scalar unit.
  {
    const int nbytes = L.P * 4;
    const int sub = lane >> 3;
    for (int o = 0; o < nbytes; o += 1024) {
      unsigned bits = 0xffu;
      if (REC && wm) {
        const int r0 = o >> 7;                             // first record of the trip (a multiple of 8: inside one mask word)
        const unsigned word = wm_regs ? (unsigned)__builtin_amdgcn_readlane((int)wm_mine, r0 >> 5) : wm[r0 >> 5];
        bits = (word >> (r0 & 31)) & 0xffu;
        if (!bits) continue;
      }
      if (((bits >> sub) & 1u) && o + lane * 16 < nbytes) dvstore4(rowb + o + lane * 16, f32x4{0.f, 0.f, 0.f, 0.f}, policy);
    }
  }
  // Pass 2: the 4x4 tiles that the union of a level's windows overlaps, out of the tile-aligned box: a lane takes half a tile (two
  // rows = two aligned 16-byte LDS reads -> 8 cells -> 16 bytes of hi and of lo, or 32 bytes of fp32).  Same-address stores of one
  // wave stay in program order, so these land on top of pass 1's zeros.
#pragma unroll 1
  for (int l = 0; l < nlev; ++l) {
    const int off = L.off[l], tw = L.tw[l], th = L.th[l];
    const int lox = __builtin_amdgcn_readlane(ox, 16 * l), loy = __builtin_amdgcn_readlane(oy, 16 * l);
    const int luw = __builtin_amdgcn_readlane(uw, 16 * l), luh = __builtin_amdgcn_readlane(uh, 16 * l);
    const int bpitch = dvs_pitch(l);
    const float* bl = &box_s[wave][l == 0 ? dvs_base(0) : l == 1 ? dvs_base(1) : l == 2 ? dvs_base(2) : dvs_base(3)];
    const int tx0 = max(lox >> 2, 0), tx1 = min((lox + luw - 1) >> 2, tw - 1);
    const int ty0 = max(loy >> 2, 0), ty1 = min((loy + luh - 1) >> 2, th - 1);
    const int ntx = tx1 - tx0 + 1, nty = ty1 - ty0 + 1;
    if (ntx <= 0 || nty <= 0) continue;                    // every window of this level lies outside the image
    const int ntask = ntx * nty * 2;
    const float rntx = 1.0f / (float)ntx;
    const int lbax = lox & ~3, lbay = loy & ~3;
    for (int k0 = 0; k0 < ntask; k0 += 64) {
      const int k = min(k0 + lane, ntask - 1);             // (surplus lanes repeat the last task up to the stores: the mask word below
      const int half = k & 1, tidx = k >> 1;               //  comes from another LANE, and ds_bpermute reads zero from a masked-off one)
      const int tyi = (int)(((float)tidx + 0.5f) * rntx), txi = tidx - tyi * ntx;      // (tidx < 64: exact)
      const int ty = ty0 + tyi, tx = tx0 + txi;
      const float* bp = bl + (4 * ty + 2 * half - lbay) * bpitch + (4 * tx - lbax);
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(bp), r1 = *reinterpret_cast<const f32x4*>(bp + bpitch);
      const int f = off + (ty * tw + tx) * 16 + half * 8;
      bool want = k0 + lane < ntask;
      if (REC && wm) {
        const int r = f >> 5;
        const unsigned word = wm_regs ? (unsigned)__shfl((int)wm_mine, r >> 5, 64) : wm[r >> 5];
        want = want && ((word >> (r & 31)) & 1u);
      }
      if (want) {
        if (REC) {
          const float v[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
          uint2 h0, l0s, h1, l1s;
          rec_split4(v, h0, l0s, rec_s);
          rec_split4(v + 4, h1, l1s, rec_s);
          char* rp = rowb + (f >> 5) * 128 + (f & 31) * 2;
          dvstore4(rp, __builtin_bit_cast(f32x4, u32x4{h0.x, h0.y, h1.x, h1.y}), policy);
          dvstore4(rp + 64, __builtin_bit_cast(f32x4, u32x4{l0s.x, l0s.y, l1s.x, l1s.y}), policy);
        } else {
          dvstore4(rowb + (int64_t)f * 4, r0, policy);
          dvstore4(rowb + (int64_t)f * 4 + 16, r1, policy);
        }
      }
    }
  }
}

__global__ void dvol_list_reset_kernel(unsigned* qlist) { qlist[0] = 0u; }

// ---- which k-tiles of the volume-backward GEMMs can hold anything but zeros ---------------------------------------------
// The gradient rows are zero outside the windows of the step's lookups (>= 80 % zero records at the bench shape), and both
// GEMMs contract over them: dF1 = s * F2cat . dV^T walks the records of 128 query rows at a time (the N tile of
// gemm_rec_nt_kernel), d2cat = s * dV^T . f1 walks, for 256 cells (its M tile), the queries in blocks of 32.  This pre-pass
// marks, from the lookups' coordinates alone, every (128-query tile, record) and every (256-cell tile, 32-query block) that a
// window can reach -- per query and level the bounding rectangle of its lookups' windows, a superset of what the gradient
// kernel writes -- and compacts the marks into ascending k-tile lists for fsraft_gemm_rec_nt_list / _tn_list.
int g_lookup_policy = -1;         // cache policy of the tiled lookup's window loads (fsraft_set_lookup_policy); -1: nt for volumes beyond the Infinity Cache
int g_ktile_exact = 0;            // levels whose windows are marked lookup by lookup instead of by bounding rectangle (fsraft_set_ktile_exact)
constexpr int KT_NQ = 128;        // queries per NT list
constexpr int KT_MC = 256;        // cells per TN list
constexpr int KT_MAXW = 256;      // words of a workgroup's record bitmap: P <= 256 * 32 * 32 floats
constexpr int KT_MAXM = 32;       // words of its cell-tile bitmap:      P <= 32 * 32 * 256 floats
struct KtArgs {
  Coords co[DV_MAXN];
  int n;
};

// bits [i0, i1] of a bitmap in LDS: one read (+ one atomic where something is missing) per word instead of per bit
__device__ __forceinline__ void kt_mark_range(unsigned* bits, int i0, int i1) {
  for (int w = i0 >> 5; w <= (i1 >> 5); ++w) {
    const int lo = w == (i0 >> 5) ? (i0 & 31) : 0, hi = w == (i1 >> 5) ? (i1 & 31) : 31;
    const unsigned m = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
    if ((bits[w] & m) != m) atomicOr(bits + w, m);
  }
}
// the cell rectangle [x0, x1] x [y0, y1] of level l: a row of 4x4-cell tiles is a run of consecutive cells, i.e. of consecutive
// records (32 cells) and cell tiles (KT_MC cells), so every tile row is one range in each bitmap
__device__ __forceinline__ void kt_mark_rect(unsigned* rb, unsigned* mbw, const VolLayout& L, int l, int x0, int x1, int y0, int y1, int mc) {
  for (int ty = y0 >> 2; ty <= (y1 >> 2); ++ty) {
    const int c0 = L.off[l] + (ty * L.tw[l] + (x0 >> 2)) * 16, c1 = L.off[l] + (ty * L.tw[l] + (x1 >> 2)) * 16 + 15;
    kt_mark_range(rb, c0 >> 5, c1 >> 5);
    kt_mark_range(mbw, c0 / mc, c1 / mc);
  }
}

template <int EXACT>      // levels below EXACT: every lookup's window marked on its own; from EXACT on: the bounding rectangle of all
__global__ __launch_bounds__(KT_NQ) void corr_ktiles_mark_kernel(KtArgs a, VolLayout L, int HW, int R, int grid_w, int* __restrict__ nt_list,
                                                                 int* __restrict__ nt_count, int nt_stride, unsigned* __restrict__ tn_bits,
                                                                 int mw, unsigned* __restrict__ wmask, int b0, int pix0) {
  // HW: queries per list set (a whole image, or the chunk [pix0, pix0 + HW) of image b0: AlternateCorrBlock's chunked backward)
  __shared__ unsigned rb[KT_MAXW];
  __shared__ unsigned mb[KT_NQ / 32][KT_MAXM];
  __shared__ int wpre[KT_MAXW + 1];
  const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int bq = b0 + b;                                  // the image the queries belong to
  const int nrec = L.P >> 5, nw = (nrec + 31) >> 5, nlev = L.nlev;
  for (int i = tid; i < nw; i += KT_NQ) rb[i] = 0u;
  for (int i = tid; i < (KT_NQ / 32) * KT_MAXM; i += KT_NQ) mb[i / KT_MAXM][i % KT_MAXM] = 0u;
  __syncthreads();
  const int pixr = tile * KT_NQ + tid, pix = pix0 + pixr;
  if (pixr < HW) {
    int mnx[4], mny[4], mxx[4], mxy[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) { mnx[l] = mny[l] = 0x7fffffff; mxx[l] = mxy[l] = -0x7fffffff; }
    // all lookups' coordinates are requested before the first one is used (fused with the marking, every lookup cost one
    // round trip: 24 us per launch for 12 lookups)
    float cxs[DV_MAXN], cys[DV_MAXN];
#pragma unroll
    for (int t = 0; t < DV_MAXN; ++t) {
      cxs[t] = cys[t] = 0.f;
      if (t < a.n) query_xy(a.co[t], bq, pix, grid_w, cxs[t], cys[t]);
    }
#pragma unroll
    for (int t = 0; t < DV_MAXN; ++t)
      if (t < a.n) {
        const float cx = cxs[t], cy = cys[t];
#pragma unroll
        for (int l = 0; l < 4; ++l)
          if (l < nlev) {
            const LevelQ q = level_query(cx, cy, l, R);
            if (l < EXACT) {
              const int x0 = max(q.wx0, 0), x1 = min(q.wx0 + 2 * R + 1, L.w[l] - 1);
              const int y0 = max(q.wy0, 0), y1 = min(q.wy0 + 2 * R + 1, L.h[l] - 1);
              if (x0 <= x1 && y0 <= y1) kt_mark_rect(rb, mb[tid >> 5], L, l, x0, x1, y0, y1, KT_MC);
            } else {
              mnx[l] = min(mnx[l], q.wx0); mxx[l] = max(mxx[l], q.wx0);
              mny[l] = min(mny[l], q.wy0); mxy[l] = max(mxy[l], q.wy0);
            }
          }
      }
#pragma unroll
    for (int l = EXACT; l < 4; ++l)
      if (l < nlev) {
        const int x0 = max(mnx[l], 0), x1 = min(mxx[l] + 2 * R + 1, L.w[l] - 1);
        const int y0 = max(mny[l], 0), y1 = min(mxy[l] + 2 * R + 1, L.h[l] - 1);
        if (x0 > x1 || y0 > y1) continue;
        kt_mark_rect(rb, mb[tid >> 5], L, l, x0, x1, y0, y1, KT_MC);
      }
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < nw; ++i) { wpre[i] = acc; acc += __popc(rb[i]); }
    wpre[nw] = acc;
    nt_count[b * gridDim.x + tile] = acc;
  }
  __syncthreads();
  int* list = nt_list + (int64_t)(b * gridDim.x + tile) * nt_stride;
  for (int k = tid; k < nrec; k += KT_NQ) {
    const unsigned w = rb[k >> 5];
    if (w >> (k & 31) & 1u) list[wpre[k >> 5] + __popc(w & ((1u << (k & 31)) - 1u))] = k;
  }
  // the cell-tile marks of this workgroup's four 32-query blocks: rows of the bit matrix corr_ktiles_tn_kernel reads
  const int ktq = (HW + 31) >> 5;
  for (int i = tid; i < (KT_NQ / 32) * mw; i += KT_NQ) {
    const int qb = tile * (KT_NQ / 32) + i / mw;
    if (qb < ktq) tn_bits[((int64_t)b * ktq + qb) * mw + i % mw] = mb[i / mw][i % mw];
  }
  // wmask [B][32-query block][nw]: the records of a query's row that either GEMM will read -- those of its 128-query tile's NT
  // list plus the eight records of every cell tile its 32-query block is listed for.  The gradient-volume kernel writes
  // these and nothing else (the rest of the row would be zero records nobody reads).
  if (wmask)
    for (int i = tid; i < (KT_NQ / 32) * nw; i += KT_NQ) {
      const int ql = i / nw, w = i % nw, qb = tile * (KT_NQ / 32) + ql;
      if (qb >= ktq) continue;
      const unsigned nib = (mb[ql][(4 * w) >> 5] >> ((4 * w) & 31)) & 0xFu;          // cell tiles 4 w .. 4 w + 3 = records 32 w .. 32 w + 31
      const unsigned ex = (nib & 1u ? 0xFFu : 0u) | (nib & 2u ? 0xFF00u : 0u) | (nib & 4u ? 0xFF0000u : 0u) | (nib & 8u ? 0xFF000000u : 0u);
      wmask[((int64_t)b * ktq + qb) * nw + w] = rb[w] | ex;
    }
}

// list of the 32-query blocks that reach cell tile blockIdx.x of batch entry blockIdx.y, ascending
__global__ __launch_bounds__(256) void corr_ktiles_tn_kernel(const unsigned* __restrict__ tn_bits, int mw, int ktq, int* __restrict__ tn_list,
                                                             int* __restrict__ tn_count, int tn_stride) {
  __shared__ int wsum[4];
  __shared__ int base_s;
  const int mt = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* list = tn_list + (int64_t)(b * gridDim.x + mt) * tn_stride;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int k0 = 0; k0 < ktq; k0 += 256) {
    const int kb = k0 + tid;
    const bool on = kb < ktq && (tn_bits[((int64_t)b * ktq + kb) * mw + (mt >> 5)] >> (mt & 31) & 1u);
    const unsigned long long bal = __ballot(on);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int pre = base_s;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    if (on) list[pre + __popcll(bal & ((1ull << lane) - 1ull))] = kb;
    __syncthreads();
    if (tid == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) tn_count[b * gridDim.x + mt] = base_s;
}

// F2cat[b][c][p]: the target-side operand of dF1 = s * F2cat . dV^T in the row layout -- level-l cell = mean of f2 over its
// 2^l x 2^l pixels where the cell exists in the floor pyramid, 0 in pad cells.
__global__ __launch_bounds__(256) void corr_f2cat_kernel(const float* __restrict__ f2, float* __restrict__ f2cat, VolLayout L, int C,
                                                         int64_t total) {
  const int H = L.H, W = L.W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int p = (int)(e % L.P);
    const int64_t bc = e / L.P;
    int l = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) l = (k < L.nlev && p >= L.off[k]) ? k : l;
    const int rel = p - L.off[l], t = rel >> 4, y = ((t / L.tw[l]) << 2) + ((rel >> 2) & 3), x = ((t % L.tw[l]) << 2) + (rel & 3);
    float v = 0.f;
    if (t < L.th[l] * L.tw[l] && y < L.h[l] && x < L.w[l]) {
      const float* s = f2 + bc * H * W;
      // the 2^l x 2^l block of the cell, all loads of a row group in flight (a run-time double loop issued them one by one:
      // 80 us per launch for 39 MB)
      float acc = 0.f;
      if (l == 0) {
        acc = gload1(s + y * W + x);
      } else if (l == 1) {
        const float* q = s + (y * 2) * W + x * 2;
        acc = (gload1(q) + gload1(q + 1)) + (gload1(q + W) + gload1(q + W + 1));
      } else if (l == 2) {
        const float* q = s + (y * 4) * W + x * 4;
        float r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = gload1(q + (i >> 2) * W + (i & 3));
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += r[i];
      } else {
        const float* q = s + (y * 8) * W + x * 8;
        for (int yy = 0; yy < 8; yy += 2) {        // two rows of eight per trip: 16 independent loads
          float r[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) r[i] = gload1(q + (yy + (i >> 3)) * W + (i & 7));
#pragma unroll
          for (int i = 0; i < 16; ++i) acc += r[i];
        }
      }
      v = acc * (1.0f / (float)(1 << (2 * l)));
    }
    f2cat[e] = v;
  }
}

// f2cat as RECORDS in one pass, one workgroup per (sample, channel) plane: the plane (H*W floats, <= F2C_MAX_PLANE) is read once
// with coalesced loads into LDS, pooled there level by level (the reference's own recursion, pytorch/core/corr.py:24-26: every
// level is the 2x2 mean of the one above it), and leaves as [32 hi | 32 lo] bf16 records -- what corr_f2cat_kernel (strided
// 4-byte reads, 46 us for 29 MB in) followed by to_records (39 MB in and out again) produced in two.
constexpr int F2C_MAX_PLANE = 12288;          // floats of one level-0 plane kept in LDS (48 KB; + 1/3 for the pooled levels)
__global__ __launch_bounds__(256) void corr_f2cat_rec_kernel(const float* __restrict__ f2, char* __restrict__ f2r, VolLayout L,
                                                             const unsigned* __restrict__ amax) {     // word of fmap2 (bounds its means too)
  extern __shared__ float pl[];               // level 0 | level 1 | level 2 | level 3, row-major, true sizes
  const float rec_s = fs_scale_of_amax(fs_amax_load(amax));
  const int H = L.H, W = L.W, HW = H * W;
  const float* src = f2 + (int64_t)blockIdx.x * HW;
  if ((HW & 3) == 0) {
    for (int e = threadIdx.x * 4; e < HW; e += 1024) *reinterpret_cast<f32x4*>(pl + e) = gload4(src + e);
  } else {
    for (int e = threadIdx.x; e < HW; e += 256) pl[e] = gload1(src + e);
  }
  int lo[4];
  lo[0] = 0;
#pragma unroll
  for (int l = 1; l < 4; ++l) lo[l] = lo[l - 1] + (l - 1 < L.nlev ? L.h[l - 1] * L.w[l - 1] : 0);
#pragma unroll
  for (int l = 1; l < 4; ++l) {
    __syncthreads();
    if (l < L.nlev) {
      const int h = L.h[l], w = L.w[l], wp = L.w[l - 1];
      const float* up = pl + lo[l - 1];
      float* dn = pl + lo[l];
      for (int e = threadIdx.x; e < h * w; e += 256) {
        const int y = e / w, x = e - y * w;
        const float* q = up + (2 * y) * wp + 2 * x;
        dn[e] = 0.25f * ((q[0] + q[1]) + (q[wp] + q[wp + 1]));
      }
    }
  }
  __syncthreads();
  char* dst = f2r + (int64_t)blockIdx.x * L.P * 4;
  for (int u = threadIdx.x; u < L.P / 8; u += 256) {     // 8 cells = two rows of one 4x4 tile
    const int p = u * 8;
    int l = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) l = (k < L.nlev && p >= L.off[k]) ? k : l;
    const int rel = p - L.off[l], t = rel >> 4, ty = t / L.tw[l], tx = t - ty * L.tw[l];
    const int y0 = (ty << 2) + ((rel >> 2) & 3), x0 = tx << 2;
    const float* lv = pl + lo[l];
    const int h = L.h[l], w = L.w[l];
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int y = y0 + (i >> 2), x = x0 + (i & 3);
      v[i] = (ty < L.th[l] && y < h && x < w) ? lv[y * w + x] : 0.f;
    }
    uint2 h0, l0, h1, l1;
    rec_split4(v, h0, l0, rec_s);
    rec_split4(v + 4, h1, l1, rec_s);
    char* d = dst + (u >> 2) * 128 + (u & 3) * 16;
    gstore4(d, __builtin_bit_cast(f32x4, u32x4{h0.x, h0.y, h1.x, h1.y}));
    gstore4(d + 64, __builtin_bit_cast(f32x4, u32x4{l0.x, l0.y, l1.x, l1.y}));
  }
}

// dF2 (channels-last [B][N][C]) = sum_l 4^-l * dF2cat[b][cell_l(y >> l, x >> l)][c] over the levels whose cell exists
__global__ __launch_bounds__(256) void corr_dfmap2_kernel(const float* __restrict__ d2cat, float* __restrict__ d2, VolLayout L, int C,
                                                          int64_t total) {
  const int N = L.H * L.W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t bp = e / C;
    const int pix = (int)(bp % N), b = (int)(bp / N);
    const int y = pix / L.W, x = pix % L.W;
    const float* src = d2cat + (int64_t)b * L.P * C + c;
    float acc = 0.f, wgt = 1.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l < L.nlev) {
        const int yl = y >> l, xl = x >> l;
        if (yl < L.h[l] && xl < L.w[l]) acc += wgt * gload1(src + (int64_t)vol_cell(L, l, yl, xl) * C);
        wgt *= 0.25f;
      }
    }
    d2[e] = acc;
  }
}

// the same with four channels per thread (C % 4 == 0): 16-byte loads and stores
__global__ __launch_bounds__(256) void corr_dfmap2_v4_kernel(const float* __restrict__ d2cat, float* __restrict__ d2, VolLayout L, int C4,
                                                             int64_t total) {
  const int N = L.H * L.W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C4);
    const int64_t bp = e / C4;
    const int pix = (int)(bp % N), b = (int)(bp / N);
    const int y = pix / L.W, x = pix % L.W;
    const float* src = d2cat + ((int64_t)b * L.P * C4 + c) * 4;
    f32x4 r[4];
    bool on[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int yl = y >> l, xl = x >> l;
      on[l] = l < L.nlev && yl < L.h[l] && xl < L.w[l];
      const int cell = on[l] ? vol_cell(L, l, yl, xl) : 0;
      r[l] = gload4(src + (int64_t)cell * C4 * 4);          // (unconditional: the four loads go out together)
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float wgt = 1.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (on[l]) acc += wgt * r[l];
      wgt *= 0.25f;
    }
    gstore4(d2 + e * 4, acc);
  }
}

template <int R>
int launch_lookup(const float* vol, const VolLayout& L, const Coords& co, float* out, int64_t nq, int HW, int grid_w, unsigned* out_amax,
                  hipStream_t s) {
  constexpr int QW = 4;
  const dim3 grid((unsigned)((nq + 4 * QW - 1) / (4 * QW)));
  // Non-temporal window loads once the volume is larger than the 256 MB Infinity Cache (every lookup then streams ~110 MB of a
  // 1 GB volume that will not be there next time anyway): 12 lookups 0.39 -> 0.367 ms in the step (same-box A/B,
  // scripts/lookup_policy_ab.sh; sc1 alone: no change).
  const int pol = g_lookup_policy >= 0 ? g_lookup_policy : ((int64_t)nq * L.P * 4 > ((int64_t)300 << 20) ? 2 : 0);
  if (g_lookup_policy == 100) { hipLaunchKernelGGL((lookup_tiled_fwd_kernel<R, QW, 2, true>), grid, dim3(256), 0, s, vol, L, co, out, nq, HW, grid_w, out_amax); return fs_launch_status(); }
  if (pol == 2) hipLaunchKernelGGL((lookup_tiled_fwd_kernel<R, QW, 2>), grid, dim3(256), 0, s, vol, L, co, out, nq, HW, grid_w, out_amax);
  else if (pol == 16) hipLaunchKernelGGL((lookup_tiled_fwd_kernel<R, QW, 16>), grid, dim3(256), 0, s, vol, L, co, out, nq, HW, grid_w, out_amax);
  else if (pol == 18) hipLaunchKernelGGL((lookup_tiled_fwd_kernel<R, QW, 18>), grid, dim3(256), 0, s, vol, L, co, out, nq, HW, grid_w, out_amax);
  else hipLaunchKernelGGL((lookup_tiled_fwd_kernel<R, QW, 0>), grid, dim3(256), 0, s, vol, L, co, out, nq, HW, grid_w, out_amax);
  return fs_launch_status();
}

}  // namespace

// out: [B, H, W, L*(2r+1)^2] channels-last.  coords element (b, c, pix) at coords[b*bs + c*cs + pix*ps].
extern "C" int fsraft_corr_lookup_tiled_fwd(const float* vol, int num_levels, const float* coords, int64_t coords_bs,
                                            int64_t coords_cs, int64_t coords_ps, float* out, int B, int H, int W, int radius,
                                            int add_grid, unsigned* out_amax, hipStream_t stream) {
  VolLayout L;
  if (!vol || !coords || !out || B < 1 || !vol_layout_make(H, W, num_levels, L) || ((uintptr_t)vol % 16) || ((uintptr_t)out_amax & 3)) return FS_ERR_ARG;
  Coords co{coords, coords_bs, coords_cs, coords_ps};
  const int64_t nq = (int64_t)B * H * W;
  if (nq >= (int64_t)1 << 31 || (int64_t)L.P * 4 >= (int64_t)1 << 31) return FS_ERR_ARG;
  if (radius == 4) return launch_lookup<4>(vol, L, co, out, nq, H * W, add_grid ? W : 0, out_amax, stream);
  if (radius == 3) return launch_lookup<3>(vol, L, co, out, nq, H * W, add_grid ? W : 0, out_amax, stream);
  return FS_ERR_ARG;
}

// dvol [B*H*W][P] (=, or += when accumulate) sum over the n lookups of (d out_t / d V)^T dout_t; dout[t]: [B,H,W,CH]
// channels-last; coords[t] with per-lookup strides coords_str[3*t + {0,1,2}] = (bs, cs, ps).  n <= 16 per call.
// records != 0: rows are written as [32 hi | 32 lo] fp16 records of dV * scale(dvol_amax) (operands of fsraft_gemm_rec_nt / _tn);
// dvol_amax: a word bounding |dV| -- (number of lookups of the step) x max |dout| does (fsraft_amax_scaled).
// Queries [q0, q0 + nq) only (nq == 0: all from q0), written to dvol rows 0 .. nq-1: the memory-efficient path builds the
// gradient volume a chunk of queries at a time.
int g_dvol_policy = 0;
extern "C" int fsraft_set_dvol_policy(int policy) {
  g_dvol_policy = policy;
  return FS_OK;
}

int g_dvol_box = 1;       // 1: corr_dvol_sep_kernel + work list where a scratch list is supplied, 0: corr_dvol_kernel for every query
extern "C" int fsraft_set_lookup_policy(int aux) {
  if (aux != -1 && aux != 0 && aux != 2 && aux != 16 && aux != 18 && aux != 100) return FS_ERR_ARG;
  g_lookup_policy = aux;
  return FS_OK;
}
extern "C" int fsraft_set_ktile_exact(int levels) {
  if (levels < 0 || levels > 2) return FS_ERR_ARG;
  g_ktile_exact = levels;
  return FS_OK;
}
extern "C" int fsraft_set_dvol_box(int on) {
  g_dvol_box = on;
  return FS_OK;
}

extern "C" int fsraft_corr_dvol_build(const float* const* dout, const float* const* coords, const int64_t* coords_str, int n,
                                      float* dvol, int num_levels, int B, int H, int W, int radius, int accumulate, int records,
                                      int add_grid, int64_t q0, int64_t nq, unsigned* qlist, const unsigned* wmask,
                                      const unsigned* dvol_amax, hipStream_t stream) {
  VolLayout L;
  if (!dout || !coords || !coords_str || !dvol || n < 1 || n > DV_MAXN || B < 1 || !vol_layout_make(H, W, num_levels, L) ||
      ((uintptr_t)dvol % 16))
    return FS_ERR_ARG;
  if (radius != 3 && radius != 4) return FS_ERR_ARG;
  DvolArgs a;
  a.n = n;
  a.amax = dvol_amax;
  for (int t = 0; t < n; ++t) {
    if (!dout[t] || !coords[t]) return FS_ERR_ARG;
    a.dout[t] = dout[t];
    a.co[t] = Coords{coords[t], coords_str[3 * t], coords_str[3 * t + 1], coords_str[3 * t + 2]};
  }
  for (int t = n; t < DV_MAXN; ++t) { a.dout[t] = nullptr; a.co[t] = Coords{nullptr, 0, 0, 0}; }
  const int N1 = 2 * radius + 1, N2 = N1 * N1;
  if (q0 < 0 || nq < 0 || q0 + nq > (int64_t)B * H * W) return FS_ERR_ARG;
  unsigned grid = (unsigned)(nq > 0 ? nq : (int64_t)B * H * W - q0);
  // fast route: one wave per query with only the lookups' bounding boxes in LDS; queries that do not fit go to `qlist`
  // (caller-owned scratch of 1 + rows unsigned), which the row-segment kernel below then walks
  const unsigned* list = nullptr;
  const bool whole = q0 == 0 && grid == (unsigned)((int64_t)B * H * W);
  const bool chunk = !whole && q0 / (H * W) == (q0 + grid - 1) / (H * W);                // a chunk inside one image
  if (wmask && !(g_dvol_box && qlist && !accumulate && records && (whole || chunk))) return FS_ERR_ARG;
  const int wm_hw = whole ? H * W : (int)grid;
  if (g_dvol_box && qlist && !accumulate && (L.P % 8) == 0) {
    hipLaunchKernelGGL(dvol_list_reset_kernel, dim3(1), dim3(1), 0, stream, qlist);
#define DVBOX(RR, REC) hipLaunchKernelGGL((corr_dvol_sep_kernel<RR, REC>), dim3((grid + 3) / 4), dim3(256), 0, stream, a, L, dvol, H * W, \
                                          add_grid ? W : 0, q0, grid, qlist, g_dvol_policy, wmask, wm_hw)
    if (radius == 4) { if (records) DVBOX(4, true); else DVBOX(4, false); }
    else { if (records) DVBOX(3, true); else DVBOX(3, false); }
#undef DVBOX
    list = qlist;
  }
  // jobs: runs of whole levels that fit the LDS segment (records need a run to start and end on a multiple of 32 floats)
  int l0 = 0;
  while (l0 < L.nlev) {
    int l1 = l0;
    auto run_end = [&](int l) { return l + 1 < L.nlev ? L.off[l + 1] : L.P; };
    while (l1 + 1 < L.nlev && run_end(l1 + 1) - L.off[l0] <= DV_SEG) ++l1;
    while (l1 + 1 < L.nlev && (run_end(l1) % 32) != 0) ++l1;                      // (only the last level may end off a record)
    const int run = run_end(l1) - L.off[l0];
    const bool clip = run > DV_SEG;
    const int GC = (l1 - l0 + 1) * N2;
    const size_t lds = (size_t)((clip ? DV_SEG : run) + ((n * GC + 3) & ~3)) * 4 + (size_t)n * 4 * sizeof(LevelQ);
#define DVOL_LAUNCH(RR, REC, CLIP) \
  hipLaunchKernelGGL((corr_dvol_kernel<RR, REC, CLIP>), dim3(grid), dim3(256), lds, stream, a, L, dvol, H * W, accumulate, l0, l1, add_grid ? W : 0, q0, g_dvol_policy, list)
    if (radius == 4) {
      if (records) { if (clip) DVOL_LAUNCH(4, true, true); else DVOL_LAUNCH(4, true, false); }
      else { if (clip) DVOL_LAUNCH(4, false, true); else DVOL_LAUNCH(4, false, false); }
    } else {
      if (records) { if (clip) DVOL_LAUNCH(3, true, true); else DVOL_LAUNCH(3, true, false); }
      else { if (clip) DVOL_LAUNCH(3, false, true); else DVOL_LAUNCH(3, false, false); }
    }
#undef DVOL_LAUNCH
    l0 = l1 + 1;
  }
  return fs_launch_status();
}

// k-tile lists of the two volume-backward GEMMs from the coordinates of the step's n <= 16 lookups (same arguments as
// fsraft_corr_dvol_build; nq > 0: ONE list set for the chunk of queries [q0, q0 + nq) of one image, B / H*W below then read
// 1 / nq).  nt_list [B][ceil(HW / 128)][nt_stride >= P / 32] + nt_count: records per 128-query tile, for
// fsraft_gemm_rec_nt_list(..., kl_by_n = 1) with dV as the B operand; tn_list [B][ceil(P / 256)][tn_stride >= ceil(HW / 32)] +
// tn_count: 32-query blocks per 256-cell tile, for fsraft_gemm_rec_tn_list(..., kl_by_n = 0) with dV as the A operand;
// tn_bits: scratch of B * ceil(HW / 32) * ceil(ceil(P / 256) / 32) unsigned; wmask (nullable): [B][ceil(HW / 32)][ceil(P / 1024)]
// unsigned, the records of each 32-query block's rows that the two list GEMMs read (for fsraft_corr_dvol_build).  Returns
// FS_ERR_ARG for shapes beyond the kernels' bitmaps (P > 262144 floats): the caller then runs the dense GEMMs.
extern "C" int fsraft_corr_bwd_ktiles(const float* const* coords, const int64_t* coords_str, int n, int num_levels, int B, int H, int W,
                                      int radius, int add_grid, int64_t q0, int64_t nq, int* nt_list, int* nt_count, int nt_stride,
                                      unsigned* tn_bits, int* tn_list, int* tn_count, int tn_stride, unsigned* wmask, hipStream_t stream) {
  VolLayout L;
  if (!coords || !coords_str || n < 1 || n > DV_MAXN || B < 1 || !vol_layout_make(H, W, num_levels, L) || !nt_list || !nt_count ||
      !tn_bits || !tn_list || !tn_count || (radius != 3 && radius != 4))
    return FS_ERR_ARG;
  // nq > 0: one list set for the queries [q0, q0 + nq) of one image (a chunk of the chunked backward), else one per image
  int HW = H * W, Bl = B, b0 = 0, pix0 = 0;
  if (nq > 0) {
    if (q0 < 0 || q0 + nq > (int64_t)B * HW || q0 / HW != (q0 + nq - 1) / HW) return FS_ERR_ARG;
    b0 = (int)(q0 / HW); pix0 = (int)(q0 % HW); HW = (int)nq; Bl = 1;
  }
  B = Bl;
  const int nrec = L.P / 32, mtiles = (L.P + KT_MC - 1) / KT_MC, ktq = (HW + 31) / 32, mw = (mtiles + 31) / 32;
  if (nrec > KT_MAXW * 32 || mw > KT_MAXM || nt_stride < nrec || tn_stride < ktq) return FS_ERR_ARG;
  KtArgs a;
  a.n = n;
  for (int t = 0; t < n; ++t) {
    if (!coords[t]) return FS_ERR_ARG;
    a.co[t] = Coords{coords[t], coords_str[3 * t], coords_str[3 * t + 1], coords_str[3 * t + 2]};
  }
  for (int t = n; t < DV_MAXN; ++t) a.co[t] = Coords{nullptr, 0, 0, 0};
#define KT_MARK(E) hipLaunchKernelGGL(corr_ktiles_mark_kernel<E>, dim3((HW + KT_NQ - 1) / KT_NQ, B), dim3(KT_NQ), 0, stream, a, L, HW, radius, \
                                      add_grid ? W : 0, nt_list, nt_count, nt_stride, tn_bits, mw, wmask, b0, pix0)
  if (g_ktile_exact >= 2) KT_MARK(2);
  else if (g_ktile_exact == 1) KT_MARK(1);
  else KT_MARK(0);
#undef KT_MARK
  hipLaunchKernelGGL(corr_ktiles_tn_kernel, dim3(mtiles, B), dim3(256), 0, stream, tn_bits, mw, ktq, tn_list, tn_count, tn_stride);
  return fs_launch_status();
}

extern "C" int fsraft_corr_f2cat(const float* fmap2, float* f2cat, int num_levels, int B, int C, int H, int W, hipStream_t stream) {
  VolLayout L;
  if (!fmap2 || !f2cat || B < 1 || C < 1 || !vol_layout_make(H, W, num_levels, L)) return FS_ERR_ARG;
  const int64_t total = (int64_t)B * C * L.P;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(corr_f2cat_kernel, dim3(blocks), dim3(256), 0, stream, fmap2, f2cat, L, C, total);
  return fs_launch_status();
}

// fmap2 [B][C][H][W] -> f2cat [B][C][P / 32] records (what fsraft_corr_f2cat + fsraft_to_records give), for planes of at most
// 12288 pixels (FS_ERR_ARG above that: the caller takes the two-kernel route)
extern "C" int fsraft_corr_f2cat_rec(const float* fmap2, void* f2r, int num_levels, int B, int C, int H, int W, const unsigned* amax2,
                                     hipStream_t stream) {
  VolLayout L;
  if (!fmap2 || !f2r || B < 1 || C < 1 || !vol_layout_make(H, W, num_levels, L) || (int64_t)H * W > F2C_MAX_PLANE ||
      ((uintptr_t)f2r % 16) || ((uintptr_t)fmap2 % 16))
    return FS_ERR_ARG;
  int fl = 0;
  for (int l = 0; l < L.nlev; ++l) fl += L.h[l] * L.w[l];
  hipLaunchKernelGGL(corr_f2cat_rec_kernel, dim3(B * C), dim3(256), (size_t)fl * 4, stream, fmap2, (char*)f2r, L, amax2);
  return fs_launch_status();
}

// d2cat [B][P][C] -> d2 [B][H*W][C] (channels-last feature gradient)
extern "C" int fsraft_corr_dfmap2(const float* d2cat, float* d2, int num_levels, int B, int C, int H, int W, hipStream_t stream) {
  VolLayout L;
  if (!d2cat || !d2 || B < 1 || C < 1 || !vol_layout_make(H, W, num_levels, L)) return FS_ERR_ARG;
  const int64_t total = (int64_t)B * H * W * C;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (C % 4 == 0 && ((uintptr_t)d2cat % 16) == 0 && ((uintptr_t)d2 % 16) == 0) {
    const int64_t t4 = total / 4;
    const int b4 = (int)((t4 + 255) / 256 < 16384 ? (t4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(corr_dfmap2_v4_kernel, dim3(b4), dim3(256), 0, stream, d2cat, d2, L, C / 4, t4);
    return fs_launch_status();
  }
  hipLaunchKernelGGL(corr_dfmap2_kernel, dim3(blocks), dim3(256), 0, stream, d2cat, d2, L, C, total);
  return fs_launch_status();
}
