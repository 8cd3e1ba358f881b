// Update-block convolutions as implicit GEMMs on the fp32 MFMA core (rows a6-a8 of
// SURVEY.md section 8; reference: pytorch/core/update.py:6-136, every nn.Conv2d there).
//
// Activations are channels-last: tensor element (pixel m, channel c) at p[m*ld + c], with
// ld % 4 == 0 and any padding channels holding zeros.  A convolution with "same" padding
//   Y[m, co] = act( bias[co] + sum_{src s} sum_{tap t} sum_{ci} X_s[m + off(t), ci] * W[co, coff_s + ci, t] )
// is the GEMM  [M = B*H*W pixels] x [N = Cout] x [K = sum_s taps * ceil32(C_s)]  whose A operand is
// gathered on the fly (zero outside the image) and whose B operand is a pre-packed weight
// matrix Wpk[n][k] (k ordered source-major, then tap, then channel, each channel run zero
// padded to a multiple of 32) produced by fsraft_pack_conv_weight.  Up to three sources
// replace torch.cat on the input side; up to three destination channel ranges replace the
// split on the output side (used by the data-gradient pass, which is this same kernel run
// on flipped/transposed packed weights).
//
// Fused epilogues: bias, ReLU, scale (mask head 0.25), and the two ConvGRU gate stages
//   ZR: z = sigmoid(.), r = sigmoid(.), stores z, r and r*h          (update.py:27-29 / 46-48)
//   Q : q = tanh(.), stores q and h' = (1-z)*h + z*q                 (update.py:29-31 / 48-50)
//
// The weight-gradient kernel is the transposed product  dWpk[co][k] += sum_m dY[m,co] * Xg[m,k]
// with the pixel dimension split across workgroups and fp32 atomics into the packed layout.
#include "gemm_core.hpp"
#include "gemm_core_split.hpp"
#include "gemm_rec.hpp"
#include <cstddef>
#include <type_traits>

namespace {

struct Src { const float* p; int C; int ld; };
struct Dst { float* p; int64_t bs, ps, cs; int n0; int accumulate; };   // channels [n0, next n0)

struct ConvArgs {
  Src src[3]; int nsrc;
  const float* wpk; int Ktot;
  const float* bias;
  int B, H, W, KH, KW, N;        // N = output channels of this GEMM
  int PH, PW;                    // tap t reads pixel (y + t / KW - PH, x + t % KW - PW); KH / 2, KW / 2 unless overridden
  Dst dst[3]; int ndst;
  int relu; float alpha;
  // GRU epilogues
  const float* h; int ldh;
  const float* z; int ldz;
  float* aux1; int ld1;          // ZR: r*h     Q: q
  float* aux2; int ld2;          // ZR: r
  int hid;
  const float* pre; int ldpre;   // GRU epilogues: per-pixel addend to the pre-activation, [M][ldpre] (NULL: none)
  // plain epilogue, per destination: ReLU-backward mask.  After scaling / accumulation, column j of the destination
  // range (j < maskc) is zeroed where rmask[m*ldmask + j] <= 0 -- the data gradient of a layer whose input came
  // out of a ReLU leaves the kernel already masked, instead of a separate pass over the tensor.
  const float* rmask[3]; int ldmask[3]; int maskc[3];
  // InstanceNorm statistics of the OUTPUT from the epilogue (kernels whose tiles lie inside one image: conv_patch.inc, the halo
  // kernel): st_sum / st_sq [B * st_slots][N] += column sums of the tile's results and of their squares (fsraft_conv_forward_stats)
  float* st_sum; float* st_sq; int st_slots;
  int swz;                       // 1: XCD-aware workgroup -> tile mapping (see tile_of_block)
  int ksplit;                    // > 1: blockIdx.z owns a slice of the k-tiles and parks its RAW partial tile in a workspace (dst[0],
                                 // rows z * M + m): the first pass of the split-K route for small M (conv_finish_kernel is the second)
  // split arithmetic (split_arith.hpp): amax words of the sources and of the packed weights (NULL: scale 1), and per
  // destination an optional word the epilogue raises to the largest magnitude it stored (GRU epilogues: damax[0] for the
  // new state h' (Q) / damax[1] for r*h (ZR); the gates themselves are bounded by 1)
  const unsigned* samax[3]; const unsigned* wamax; unsigned* damax[3];
};

// scale of the A operand (one for all sources: they share the accumulators) and the factor that takes the accumulators
// back: 1 / (s_a s_w), exact (powers of two)
__device__ __forceinline__ void conv_scales(const ConvArgs& a, float& sa, float& inv) {
  unsigned m = fs_amax_load(a.samax[0]);
  if (a.nsrc > 1) m = fs_umax(m, fs_amax_load(a.samax[1]));
  if (a.nsrc > 2) m = fs_umax(m, fs_amax_load(a.samax[2]));
  sa = fs_scale_of_amax(m);
  inv = fs_inv_scale(sa) * fs_inv_scale(fs_scale_of_amax(fs_amax_load(a.wamax)));
}

// Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  With the plain
// (x = N tile, y = M tile) mapping the N tiles of one M tile -- which read the same activation rows -- land on
// different XCDs.  This maps linear id i to logical tile t so that every XCD owns one contiguous run of tiles:
// XCD x holds ids {x, x+8, ...}; its run starts at x*(T/8) + min(x, T%8).
__device__ __forceinline__ void tile_of_block(int swz, int& bx, int& by) {
  if (!swz) return;
  const int nx = gridDim.x, T = nx * gridDim.y, i = by * nx + bx;
  const int x = i & 7, q = T >> 3, r = T & 7;
  const int t = x * q + (x < r ? x : r) + (i >> 3);
  bx = t % nx; by = t / nx;
}

// Arguments of the buffer-addressed split kernels: ConvArgs plus one dword per k-tile, built on the host, that
// says where the tile comes from -- bits 0..15: SGPR byte offset / 16 of (tap shift, channel chunk) inside the
// source, 16..19: tap, 20..21: source, 22..27: channels left in the source from this chunk (1..32).  The kernel
// reads it with one scalar load; without it the (source, tap, chunk) decode is two integer divisions per k-tile,
// which the compiler can only do on the vector ALU (~50 instructions) even though the values are wave-uniform.
// "Uniform" form (BUF = 2), used when all sources share one row pitch and lie within 2 GiB of each other: two dwords
// per k-tile -- the complete SGPR byte offset from ONE base pointer (source delta + tap shift + channel chunk), and
// tap | channels-left << 4.  One descriptor and one pitch for the whole k-loop: the per-tile scalar work shrinks from
// ~30 instructions (source selects, 64-bit base arithmetic) to a two-dword load and two bit-field extracts.
constexpr int KTAB_MAX = 512;
struct ConvArgsT {
  ConvArgs a;
  const float* ubase; int uld;   // uniform form: biased base pointer and the common pitch (floats)
  unsigned ktab[KTAB_MAX];
};

enum { EPI_PLAIN = 0, EPI_ZR = 2, EPI_Q = 3 };

template <class Cfg>
struct ConvALoader {
  static constexpr int BM = Cfg::BM, BK = Cfg::BK, LD = Cfg::LDA;
  static constexpr int F4 = BK / 4;
  static constexpr int NF4 = BM * F4 / 256;
  static constexpr int NREG = NF4 * 4;
  static constexpr int NCH = NF4;
  const float* p0; const float* p1; const float* p2;
  int C0, C1, C2, ld0, ld1, ld2;
  int cpt0, cpt1, cpt2;          // 32-channel chunks per tap for each source
  int taps, KW, PH, PW, H, W;
  int py[NF4], px[NF4];          // pixel coordinates of this thread's rows (py < 0: row outside M)
  int64_t pb[NF4];               // image base pixel index b*H*W
  __device__ __forceinline__ bool fetch_chunk(int kt, float (&r)[NREG], int j) const {
    // source select written as sums of two-way selects: a three-way select chain over kernel
    // arguments is turned by hipcc into a private (scratch) lookup table that is then re-read,
    // with s_waitcnt vmcnt(0), in front of every load of the k-loop
    const int n0 = taps * cpt0, n1 = taps * cpt1;
    const bool is1 = kt >= n0, is2 = kt >= n0 + n1;
    const int k = kt - (is1 ? n0 : 0) - (is2 ? n1 : 0);
    const int C = C0 + (is1 ? C1 - C0 : 0) + (is2 ? C2 - C1 : 0);
    const int ld = ld0 + (is1 ? ld1 - ld0 : 0) + (is2 ? ld2 - ld1 : 0);
    const int cpt = cpt0 + (is1 ? cpt1 - cpt0 : 0) + (is2 ? cpt2 - cpt1 : 0);
    const float* p = p0 + (is1 ? p1 - p0 : 0) + (is2 ? p2 - p1 : 0);
    const int tap = k / cpt, c0 = (k % cpt) * BK;
    const int dy = tap / KW - PH, dx = tap % KW - PW;
    const int kq = (threadIdx.x + 256 * j) % F4;
    const int yy = py[j] + dy, xx = px[j] + dx, c = c0 + kq * 4;
    // unconditional load from a clamped address; the zero-select happens in store_chunk so that
    // nothing consumes the load result here (a use would force s_waitcnt right behind the load)
    const bool ok = py[j] >= 0 && yy >= 0 && yy < H && xx >= 0 && xx < W && c < C;
    const f32x4 v = gload4(p + (ok ? (pb[j] + (int64_t)yy * W + xx) * ld + c : 0));
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
    return ok;
  }
  __device__ __forceinline__ void store_chunk(float* t, const float (&r)[NREG], int j, bool ok) const {
    const int e = threadIdx.x + 256 * j;
    const int row = e / F4, kq = e % F4;
#pragma unroll
    for (int c = 0; c < 4; ++c) t[(kq * 4 + c) * LD + row] = ok ? r[4 * j + c] : 0.f;
  }
};

// ---- loaders of the split-bf16 core ---------------------------------------------------------
template <class Cfg>
struct SplitConvALoader {                 // implicit-GEMM gather of fp32 activations, converted at staging time
  static constexpr int NCH = Cfg::NCH_A, NREG = NCH * 4;
  const float* p0; const float* p1; const float* p2;
  int C0, C1, C2, ld0, ld1, ld2, cpt0, cpt1, cpt2;
  int taps, KW, PH, PW, H, W;
  int py[NCH], px[NCH];                   // pixel coordinates of this thread's rows (py < 0: row outside M)
  int pofs[NCH];                          // pixel index b*H*W + y*W + x of the row
  float s;                                // scale of the activations (split_arith.hpp)
  __device__ __forceinline__ void fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int n0 = taps * cpt0, n1 = taps * cpt1;
    const bool is1 = kt >= n0, is2 = kt >= n0 + n1;
    const int k = kt - (is1 ? n0 : 0) - (is2 ? n1 : 0);
    const int C = C0 + (is1 ? C1 - C0 : 0) + (is2 ? C2 - C1 : 0);
    const int ld = ld0 + (is1 ? ld1 - ld0 : 0) + (is2 ? ld2 - ld1 : 0);
    const int cpt = cpt0 + (is1 ? cpt1 - cpt0 : 0) + (is2 ? cpt2 - cpt1 : 0);
    const float* p = p0 + (is1 ? p1 - p0 : 0) + (is2 ? p2 - p1 : 0);
    const int tap = k / cpt, c0 = (k % cpt) * 32;
    const int dy = tap / KW - PH, dx = tap % KW - PW;
    const int kq = (threadIdx.x + 256 * j) & 7;
    const int c = c0 + kq * 4;
    const bool ok = ((unsigned)(py[j] + dy) < (unsigned)H) && ((unsigned)(px[j] + dx) < (unsigned)W) && c < C;
    const float* src = ok ? p + (int64_t)(pofs[j] + dy * W + dx) * ld + c : g_fsraft_zero16;
    const f32x4 v = gload4(src);
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
  }
  __device__ __forceinline__ void stage_chunk(char* tile, const float* r4, int j) const {
    stage_convert<Cfg::PITCH>(tile, threadIdx.x + 256 * j, r4, s);
  }
};

// ---- buffer-addressed variants -----------------------------------------------------------------
// Same tiles, but fetched with buffer_load_dwordx4: the per-lane part of the address is a 32-bit byte offset that
// is constant over the whole k-loop (this thread's pixel row and 16-byte column), the per-k-tile part (tap shift,
// channel chunk, source) is wave-uniform and travels in the SGPR offset, and "outside the image / outside the
// matrix" is expressed by pointing the lane offset past num_records, which makes the hardware return zeros.
// That removes the 64-bit per-lane address arithmetic and the bounds compares (about 15 VALU per 16-byte chunk)
// from the k-loop; what is left per chunk is one mask test and one select.
#define FS_RSRC_FLAGS 0x00020000            // raw buffer, 32-bit data format (gfx9 family word 3)
#define FS_OOB 0x80000000u                  // lane offset that always fails the num_records check

__device__ __forceinline__ unsigned uni(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ const T* uni_ptr(const T* p) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  return reinterpret_cast<const T*>((uint64_t)uni((unsigned)(u >> 32)) << 32 | uni((unsigned)u));
}
// p and bytes must be wave-uniform; the readfirstlanes state that (a descriptor the compiler believes to be
// divergent is wrapped in a waterfall loop around every load).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  void* q = reinterpret_cast<void*>((uint64_t)hi << 32 | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), FS_RSRC_FLAGS);
}

template <class Cfg>
struct BufConvALoader {
  static constexpr int NCH = Cfg::NCH_A, NREG = NCH * 4;
  // descriptor bases, biased down by (PH*W + PW) pixels so that tap shifts stay non-negative, and sizes in bytes.
  // (Descriptors are rebuilt from these scalars at each use: a struct holding __amdgpu_buffer_rsrc_t members is not
  // scalarised by the compiler and ends up in scratch, which turns every uniform value read back from it divergent.)
  const float* b0; const float* b1; const float* b2; unsigned nb0, nb1, nb2;
  unsigned ld0x4, ld1x4, ld2x4;           // row pitches in bytes
  const unsigned __attribute__((address_space(4)))* ktab;   // table in the kernarg segment (constant address space -> s_load), see ConvArgsT
  unsigned tapmask[NCH];                  // bit t: tap t of this thread's pixel row lies inside the image (0: row outside M)
  unsigned pofs[NCH];                     // pixel index of the row
  unsigned kq16;                          // byte offset of this thread's 16-byte column inside a 128-byte chunk row
  float s;                                // scale of the activations
  __device__ __forceinline__ void fetch_tile(int kt, float (&r)[NREG]) const {
    // kt is wave-uniform; saying so explicitly lets the table entry come in with s_load_dword and keeps everything
    // derived from it (descriptor, SGPR offset) in scalar registers -- once per tile, not once per chunk
    const unsigned e = ktab[__builtin_amdgcn_readfirstlane(kt)];
    const unsigned soff = (e & 0xffffu) << 4, tap = (e >> 16) & 15u, src = (e >> 20) & 3u, crem = (e >> 22) & 63u;
    const bool s1 = src >= 1, s2 = src >= 2;
    // (sums of two-way selects: a three-way select chain over kernel arguments becomes a scratch lookup table)
    const unsigned ldb = ld0x4 + (s1 ? ld1x4 - ld0x4 : 0u) + (s2 ? ld2x4 - ld1x4 : 0u);
    const float* bp = b0 + (s1 ? b1 - b0 : 0) + (s2 ? b2 - b1 : 0);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(bp, 0x7fffffffu);
    const unsigned cok = kq16 < crem * 4u ? 1u : 0u;              // this lane's 16-byte column holds channels of the source
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      // branch-free: lanes whose tap falls outside the image (or whose column is padding) get bit 31 set in their
      // offset, which fails the descriptor's range check and reads as zeros
      const unsigned bad = (__builtin_amdgcn_ubfe(tapmask[j], tap, 1u) & cok) ^ 1u;
      const unsigned voff = (bad << 31) | (__umul24(pofs[j], ldb) + kq16);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
      const f32x4 f = __builtin_bit_cast(f32x4, v);
      r[4 * j + 0] = f[0]; r[4 * j + 1] = f[1]; r[4 * j + 2] = f[2]; r[4 * j + 3] = f[3];
    }
  }
  __device__ __forceinline__ void stage_chunk(char* tile, const float* r4, int j) const {
    stage_convert<Cfg::PITCH>(tile, threadIdx.x + Cfg::NT * j, r4, s);
  }
};

template <class Cfg>
struct BufConvALoaderU {                  // uniform form: one base, one pitch, two table dwords per k-tile
  static constexpr int NCH = Cfg::NCH_A, NREG = NCH * 4;
  const float* base;
  const unsigned __attribute__((address_space(4)))* ktab;
  unsigned tapmask[NCH];                  // bit t: tap t of this thread's pixel row lies inside the image (0: row outside M)
  unsigned voff0[NCH];                    // pixel * pitch + 16-byte column, in bytes (constant over the k-loop)
  unsigned kq16;
  float s;                                // scale of the activations
  __device__ __forceinline__ void fetch_tile(int kt, float (&r)[NREG]) const {
    const int ku = __builtin_amdgcn_readfirstlane(kt);
    const unsigned soff = ktab[2 * ku], e = ktab[2 * ku + 1];
    const unsigned tap = e & 15u, crem4 = (e >> 4) * 4u;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, 0x7fffffffu);
    const unsigned cok = kq16 < crem4 ? 1u : 0u;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const unsigned bad = (__builtin_amdgcn_ubfe(tapmask[j], tap, 1u) & cok) ^ 1u;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (bad << 31) | voff0[j], soff, 0);
      const f32x4 f = __builtin_bit_cast(f32x4, v);
      r[4 * j + 0] = f[0]; r[4 * j + 1] = f[1]; r[4 * j + 2] = f[2]; r[4 * j + 3] = f[3];
    }
  }
  __device__ __forceinline__ void stage_chunk(char* tile, const float* r4, int j) const {
    stage_convert<Cfg::PITCH>(tile, threadIdx.x + Cfg::NT * j, r4, s);
  }
};

template <class Cfg>
struct BufWeightLoader {                  // pre-split packed weights through one buffer descriptor
  static constexpr int NCH = Cfg::NCH_B, NREG = NCH * 4;
  const char* base; unsigned nbytes;
  unsigned voff[NCH];                     // row * row_bytes + 16 * part, or FS_OOB for rows outside the matrix
  __device__ __forceinline__ void fetch_tile(int kt, float (&r)[NREG]) const {
    const int ku = __builtin_amdgcn_readfirstlane(kt);            // kt < 0: zero tile = a descriptor with no records
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, ku < 0 ? 0u : nbytes);
    const unsigned soff = ku < 0 ? 0u : (unsigned)ku * 128u;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j], soff, 0);
      const f32x4 f = __builtin_bit_cast(f32x4, v);
      r[4 * j + 0] = f[0]; r[4 * j + 1] = f[1]; r[4 * j + 2] = f[2]; r[4 * j + 3] = f[3];
    }
  }
  __device__ __forceinline__ void stage_chunk(char* tile, const float* r4, int j) const {
    stage_copy<Cfg::PITCH>(tile, threadIdx.x + Cfg::NT * j, r4);
  }
};

template <class Cfg>
struct SplitWeightLoader {                // pre-split packed weights: row n = Ktot/32 records of [32 hi | 32 lo] bf16
  static constexpr int NCH = Cfg::NCH_B, NREG = NCH * 4;
  const char* base;                       // row n0 of the packed matrix
  int64_t row_bytes;                      // Ktot * 4
  int rows_valid;
  __device__ __forceinline__ void fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int row = e >> 3, part = e & 7;
    const char* src = (row < rows_valid && kt >= 0) ? base + row * row_bytes + (int64_t)kt * 128 + part * 16
                                                    : reinterpret_cast<const char*>(g_fsraft_zero16);   // kt < 0: zero tile
    const f32x4 v = gload4(src);
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
  }
  __device__ __forceinline__ void stage_chunk(char* tile, const float* r4, int j) const {
    stage_copy<Cfg::PITCH>(tile, threadIdx.x + 256 * j, r4);
  }
};

// inv: factor that un-scales the accumulators of the split kernels (conv_scales); 1 for the exact-fp32 kernels
template <class Cfg, int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0, int n0, float inv = 1.0f) {
  const int HW = a.H * a.W;
  const int M = a.B * HW;
  unsigned mx[3] = {0u, 0u, 0u};              // largest stored magnitude per destination (bit patterns), for a.damax
  // Epilogue.  Everything is batched per 32x32 MFMA tile: 16 addresses, then (optionally) 16
  // loads in flight, then 16 stores, with no wait between consecutive stores.  Rows of one
  // accumulator tile are m = mbase + (r&3) + 8*(r>>2); a 128-row tile crosses at most one
  // image boundary, so (batch, pixel) is derived from one division per tile.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int nt = 0; nt < Cfg::TN; ++nt) {
    const int n = n0 + acc_col<Cfg>(nt);
    const bool nok = n < a.N;
    const float bias = (nok && a.bias) ? a.bias[n] : 0.f;
    int di = 0;
    if (a.ndst > 1 && n >= a.dst[1].n0) di = 1;
    if (a.ndst > 2 && n >= a.dst[2].n0) di = 2;
    float* dp = di == 0 ? a.dst[0].p : di == 1 ? a.dst[1].p : a.dst[2].p;
    const int64_t dbs = di == 0 ? a.dst[0].bs : di == 1 ? a.dst[1].bs : a.dst[2].bs;
    const int64_t dps = di == 0 ? a.dst[0].ps : di == 1 ? a.dst[1].ps : a.dst[2].ps;
    const int64_t dcs = di == 0 ? a.dst[0].cs : di == 1 ? a.dst[1].cs : a.dst[2].cs;
    const int dn0 = di == 0 ? a.dst[0].n0 : di == 1 ? a.dst[1].n0 : a.dst[2].n0;
    const bool dacc = (di == 0 ? a.dst[0].accumulate : di == 1 ? a.dst[1].accumulate : a.dst[2].accumulate) != 0;
    const float* mk = di == 0 ? a.rmask[0] : di == 1 ? a.rmask[1] : a.rmask[2];
    const int ldm = di == 0 ? a.ldmask[0] : di == 1 ? a.ldmask[1] : a.ldmask[2];
    const int mkc = di == 0 ? a.maskc[0] : di == 1 ? a.maskc[1] : a.maskc[2];
#pragma unroll
    for (int mt = 0; mt < Cfg::TM; ++mt) {
      const int mbase = m0 + (wave / Cfg::WN) * (Cfg::TM * 32) + mt * 32 + 4 * (lane >> 5);
      const int b0 = mbase / HW, pix0 = mbase - b0 * HW;
      if (EPI == EPI_PLAIN) {
        // two batches of 8 rows: 8 element offsets (32-bit), 8 optional loads in flight, 8 stores
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          int off[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = hb * 8 + q;
            const int d = (r & 3) + 8 * (r >> 2);
            int pix = pix0 + d, b = b0;
            if (pix >= HW) { pix -= HW; ++b; }
            const bool ok = nok && (mbase + d < M);
            off[q] = ok ? (int)(b * dbs + pix * dps + (n - dn0) * dcs) : -1;
          }
          float old[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) old[q] = (dacc && off[q] >= 0) ? dp[off[q]] : 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float v = __builtin_fmaf(acc[mt][nt][hb * 8 + q], inv, bias) * a.alpha;
            if (a.relu) v = fmaxf(v, 0.f);
            v += old[q];
            if (off[q] >= 0 && mk && n - dn0 < mkc) {
              const int r = hb * 8 + q;
              const int64_t m = mbase + (r & 3) + 8 * (r >> 2);
              if (mk[m * ldm + (n - dn0)] <= 0.f) v = 0.f;
            }
            if (off[q] >= 0) { dp[off[q]] = v; mx[di] = fs_umax(mx[di], fs_abs_bits(v)); }
          }
        }
      } else if (EPI == EPI_ZR) {
        const bool isz = n < a.hid;
        const int c = isz ? n : n - a.hid;
        float hh[16];
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          ok[r] = nok && m < M;
          hh[r] = (ok[r] && !isz) ? a.h[(int64_t)m * a.ldh + c] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = mbase + (r & 3) + 8 * (r >> 2);
          const float pre = (a.pre && ok[r]) ? a.pre[m * a.ldpre + n] : 0.f;
          const float sg = 1.0f / (1.0f + expf(-(__builtin_fmaf(acc[mt][nt][r], inv, bias) + pre)));
          if (ok[r]) {
            if (isz) {
              a.dst[0].p[m * a.dst[0].ps + c] = sg;                 // z
            } else {
              a.aux2[m * a.ld2 + c] = sg;                           // r
              a.aux1[m * a.ld1 + c] = sg * hh[r];                   // r*h
              mx[1] = fs_umax(mx[1], fs_abs_bits(sg * hh[r]));
            }
          }
        }
      } else {   // EPI_Q
        float hh[16], zz[16];
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = mbase + (r & 3) + 8 * (r >> 2);
          ok[r] = nok && m < M;
          hh[r] = ok[r] ? a.h[m * a.ldh + n] : 0.f;
          zz[r] = ok[r] ? a.z[m * a.ldz + n] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = mbase + (r & 3) + 8 * (r >> 2);
          const float pre = (a.pre && ok[r]) ? a.pre[m * a.ldpre + n] : 0.f;
          const float q = tanhf(__builtin_fmaf(acc[mt][nt][r], inv, bias) + pre);
          if (ok[r]) {
            const float hn = (1.f - zz[r]) * hh[r] + zz[r] * q;
            a.aux1[m * a.ld1 + n] = q;
            a.dst[0].p[m * a.dst[0].ps + n] = hn;
            mx[0] = fs_umax(mx[0], fs_abs_bits(hn));
          }
        }
      }
    }
  }
  if (a.damax[0] || a.damax[1] || a.damax[2]) {          // (workgroup-uniform)
    __shared__ unsigned red[16];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (a.damax[i]) fs_amax_commit(a.damax[i], mx[i], red);
  }
}

// Epilogue through LDS for channels-last destinations: the accumulator tile is parked in LDS (row
// pitch BN+4 floats) and leaves as whole rows, 16 bytes per lane and 512 contiguous bytes per row of a
// 128-wide tile, instead of 64 four-byte stores per lane.  GRU gate math runs on the float4s.
// PATCH: the tile's 256 rows are an 8 x 32 patch of pixels (conv_patch.inc): row r is pixel (py0 + r / 32, px0 + r % 32) of
// image pb, rows outside the image are skipped.  NTW: threads taking part (the workgroup's).
template <class Cfg, int EPI, int NTW = Cfg::NT, bool PATCH = false>
__device__ __forceinline__ void conv_epilogue_lds(const ConvArgs& a, f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0, int n0,
                                                  float* __restrict__ tile, bool owner = true, int pb = 0, int py0 = 0, int px0 = 0,
                                                  int64_t mofs = 0, float inv = 1.0f) {
  constexpr int LD = Cfg::BN + 4;
  const int HW = a.H * a.W;
  const int M = a.B * HW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // row of the tile -> row of the [M][*] tensors (-1: nothing there)
  auto row_m = [&](int row) -> int64_t {
    if constexpr (PATCH) {
      const int y = py0 + (row >> 5), x = px0 + (row & 31);
      return (y < a.H && x < a.W) ? (int64_t)(pb * a.H + y) * a.W + x : -1;
    } else {
      return m0 + row < M ? (int64_t)m0 + row : -1;
    }
  };
  // (the look at the destination words is taken here, long before they are needed: split_arith.hpp fs_amax_peek)
  const unsigned peek0 = fs_amax_peek(a.damax[0]), peek1 = fs_amax_peek(a.damax[1]), peek2 = fs_amax_peek(a.damax[2]);
  __syncthreads();                                   // the k-loop's last LDS reads are done
#pragma unroll
  for (int nt = 0; nt < Cfg::TN; ++nt) {
    const int nl = acc_col<Cfg>(nt);
    const float bias = (n0 + nl < a.N && a.bias) ? a.bias[n0 + nl] : 0.f;
#pragma unroll
    for (int mt = 0; mt < Cfg::TM; ++mt) {
      const int rbase = (wave / Cfg::WN) * (Cfg::TM * 32) + mt * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[(rbase + (r & 3) + 8 * (r >> 2)) * LD + nl] = __builtin_fmaf(acc[mt][nt][r], inv, bias);
    }
  }
  __syncthreads();
  constexpr int C4 = Cfg::BN / 4;                    // float4 columns per row
  constexpr int RPP = NTW / C4;                      // rows covered per pass
  (void)owner;
  const int c4 = threadIdx.x % C4, rsub = threadIdx.x / C4;
  const int n = n0 + c4 * 4;
  const bool stats = PATCH && EPI == EPI_PLAIN && a.st_sum != nullptr;      // (workgroup-uniform: the barriers below are safe)
  const bool track = a.damax[0] != nullptr || a.damax[1] != nullptr || a.damax[2] != nullptr;   // (workgroup-uniform too)
  if (n >= a.N && !stats && !track) return;
  unsigned mx = 0u;                                  // largest magnitude this thread stored (bit pattern), for a.damax
  int mydi = 0;
  if (EPI == EPI_PLAIN) {
    f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
    if (n < a.N) {
    int di = 0;
    if (a.ndst > 1 && n >= a.dst[1].n0) di = 1;
    if (a.ndst > 2 && n >= a.dst[2].n0) di = 2;
    mydi = di;
    float* dp = di == 0 ? a.dst[0].p : di == 1 ? a.dst[1].p : a.dst[2].p;
    const int64_t dps = di == 0 ? a.dst[0].ps : di == 1 ? a.dst[1].ps : a.dst[2].ps;
    const int dn0 = di == 0 ? a.dst[0].n0 : di == 1 ? a.dst[1].n0 : a.dst[2].n0;
    const bool dacc = (di == 0 ? a.dst[0].accumulate : di == 1 ? a.dst[1].accumulate : a.dst[2].accumulate) != 0;
    const float* mk = di == 0 ? a.rmask[0] : di == 1 ? a.rmask[1] : a.rmask[2];
    const int ldm = di == 0 ? a.ldmask[0] : di == 1 ? a.ldmask[1] : a.ldmask[2];
    const int mkc = di == 0 ? a.maskc[0] : di == 1 ? a.maskc[1] : a.maskc[2];
    const int nv = a.N - n < 4 ? a.N - n : 4;        // valid columns of this float4 (N need not be a multiple of 4)
#pragma unroll 4
    for (int row = rsub; row < Cfg::BM; row += RPP) {
      const int64_t m = row_m(row);
      if (m < 0) { if (PATCH) continue; else break; }
      f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * LD + c4 * 4);
      float* o = dp + (int64_t)(m + mofs) * dps + (n - dn0);      // (mofs: the k-slice's slab of the split-K workspace)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] *= a.alpha;
        if (a.relu) v[i] = fmaxf(v[i], 0.f);
      }
      if (nv == 4) {
        if (stats) { ssum += v; ssq += v * v; }
        if (dacc) {
          const f32x4 old = gload4(o);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += old[i];
        }
        if (mk) {
          const f32x4 y = gload4(mk + (int64_t)m * ldm + (n - dn0));
#pragma unroll
          for (int i = 0; i < 4; ++i) if (n - dn0 + i < mkc && y[i] <= 0.f) v[i] = 0.f;
        }
        gstore4(o, v);
        if (track) mx = fs_umax(fs_umax(mx, fs_umax(fs_abs_bits(v[0]), fs_abs_bits(v[1]))), fs_umax(fs_abs_bits(v[2]), fs_abs_bits(v[3])));
      } else {
        for (int i = 0; i < nv; ++i) {
          float r = dacc ? gload1(o + i) + v[i] : v[i];
          if (mk && n - dn0 + i < mkc && gload1(mk + (int64_t)m * ldm + (n - dn0) + i) <= 0.f) r = 0.f;
          gstore1(o + i, r);
          mx = fs_umax(mx, fs_abs_bits(r));
        }
      }
    }
    }
    if constexpr (PATCH) {
      if (stats) {
        // column sums of the tile: every thread summed its rows of its four columns above; the RPP row groups meet in LDS
        // (the parked tile is no longer needed) and one thread per column adds the workgroup's share to the slot row of its image
        __syncthreads();
        float* red = tile;
        *reinterpret_cast<f32x4*>(red + (rsub * Cfg::BN) + c4 * 4) = ssum;
        *reinterpret_cast<f32x4*>(red + ((RPP + rsub) * Cfg::BN) + c4 * 4) = ssq;
        __syncthreads();
        if (threadIdx.x < Cfg::BN && n0 + (int)threadIdx.x < a.N) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll 4
          for (int g = 0; g < RPP; ++g) { t1 += red[g * Cfg::BN + threadIdx.x]; t2 += red[(RPP + g) * Cfg::BN + threadIdx.x]; }
          const int64_t o = ((int64_t)pb * a.st_slots + (int)(blockIdx.y % (unsigned)a.st_slots)) * a.N + n0 + threadIdx.x;
          atomicAdd(a.st_sum + o, t1);
          atomicAdd(a.st_sq + o, t2);
        }
      }
    }
  } else if (EPI == EPI_ZR) {
    const bool isz = n < a.hid;
    const int c = isz ? n : n - a.hid;
    mydi = 1;
    if (n < a.N) {
#pragma unroll 4
    for (int row = rsub; row < Cfg::BM; row += RPP) {
      const int64_t mr = row_m(row);
      if (mr < 0 && !PATCH) break;
      // (patch tiles: rows outside the image read row 0 and store nothing -- no branch around the loads, so the unrolled
      //  iterations' loads are issued together; with half as many threads as the 16-wave kernels the loop is latency-bound)
      const int64_t m = mr < 0 ? 0 : mr;
      f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * LD + c4 * 4);
      if (a.pre) v += gload4(a.pre + m * a.ldpre + n);
      f32x4 hh = {0.f, 0.f, 0.f, 0.f};
      if (!isz) hh = gload4(a.h + m * a.ldh + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = 1.0f / (1.0f + expf(-v[i]));
      if (mr < 0) continue;
      if (isz) {
        *reinterpret_cast<f32x4*>(a.dst[0].p + m * a.dst[0].ps + c) = v;                 // z
      } else {
        *reinterpret_cast<f32x4*>(a.aux2 + m * a.ld2 + c) = v;                            // r
        f32x4 rh;
#pragma unroll
        for (int i = 0; i < 4; ++i) rh[i] = v[i] * hh[i];
        *reinterpret_cast<f32x4*>(a.aux1 + m * a.ld1 + c) = rh;                           // r*h
        if (track) mx = fs_umax(fs_umax(mx, fs_umax(fs_abs_bits(rh[0]), fs_abs_bits(rh[1]))), fs_umax(fs_abs_bits(rh[2]), fs_abs_bits(rh[3])));
      }
    }
    }
  } else {   // EPI_Q
    if (n < a.N) {
#pragma unroll 4
    for (int row = rsub; row < Cfg::BM; row += RPP) {
      const int64_t m = row_m(row);
      if (m < 0) { if (PATCH) continue; else break; }
      f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * LD + c4 * 4);
      if (a.pre) v += gload4(a.pre + m * a.ldpre + n);
      const f32x4 hh = gload4(a.h + m * a.ldh + n);
      const f32x4 zz = gload4(a.z + m * a.ldz + n);
      f32x4 hn;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = tanhf(v[i]);
        hn[i] = (1.f - zz[i]) * hh[i] + zz[i] * v[i];
      }
      *reinterpret_cast<f32x4*>(a.aux1 + m * a.ld1 + n) = v;                              // q
      *reinterpret_cast<f32x4*>(a.dst[0].p + m * a.dst[0].ps + n) = hn;                   // h'
      if (track) mx = fs_umax(fs_umax(mx, fs_umax(fs_abs_bits(hn[0]), fs_abs_bits(hn[1]))), fs_umax(fs_abs_bits(hn[2]), fs_abs_bits(hn[3])));
    }
    }
  }
  if (track) {                                       // the parked tile is no longer needed: its memory takes the reduction
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (a.damax[i]) fs_amax_commit_peeked(a.damax[i], mydi == i ? mx : 0u, reinterpret_cast<unsigned*>(tile), i == 0 ? peek0 : i == 1 ? peek1 : peek2);
  }
}

// rows of every destination are 16-byte aligned and channels-last (column stride 1)?
__device__ __forceinline__ bool epilogue_rows_ok(const ConvArgs& a) {
  bool ok = true;
  for (int i = 0; i < a.ndst; ++i)
    ok = ok && a.dst[i].cs == 1 && (a.dst[i].ps & 3) == 0 && (a.dst[i].n0 & 3) == 0 && ((uintptr_t)a.dst[i].p & 15) == 0 &&
         a.dst[i].bs == a.dst[i].ps * (int64_t)(a.H * a.W);
  return ok;
}

bool epilogue_rows_ok_host(const ConvArgs& a) {
  bool ok = true;
  for (int i = 0; i < a.ndst; ++i)
    ok = ok && a.dst[i].cs == 1 && (a.dst[i].ps & 3) == 0 && (a.dst[i].n0 & 3) == 0 && ((uintptr_t)a.dst[i].p & 15) == 0 &&
         a.dst[i].bs == a.dst[i].ps * (int64_t)(a.H * a.W);
  return ok;
}

// exact-fp32 variant: v_mfma_f32_32x32x2_f32
template <class Cfg, int EPI>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  const int HW = a.H * a.W;
  const int M = a.B * HW;
  const int n0 = blockIdx.x * Cfg::BN, m0 = blockIdx.y * Cfg::BM;
  ConvALoader<Cfg> la;
  // k-tiles per tap: channel runs are packed in multiples of 32 whatever BK is
  la.p0 = a.src[0].p; la.C0 = a.src[0].C; la.ld0 = a.src[0].ld; la.cpt0 = (a.src[0].C + 31) / 32 * (32 / Cfg::BK);
  la.p1 = a.src[1].p; la.C1 = a.src[1].C; la.ld1 = a.src[1].ld; la.cpt1 = a.nsrc > 1 ? (a.src[1].C + 31) / 32 * (32 / Cfg::BK) : 0;
  la.p2 = a.src[2].p; la.C2 = a.src[2].C; la.ld2 = a.src[2].ld; la.cpt2 = a.nsrc > 2 ? (a.src[2].C + 31) / 32 * (32 / Cfg::BK) : 0;
  if (a.nsrc < 2) { la.p1 = a.src[0].p; la.C1 = 0; la.ld1 = 4; la.cpt1 = 1; }
  if (a.nsrc < 3) { la.p2 = a.src[0].p; la.C2 = 0; la.ld2 = 4; la.cpt2 = 1; }
  la.taps = a.KH * a.KW; la.KW = a.KW; la.PH = a.PH; la.PW = a.PW; la.H = a.H; la.W = a.W;
#pragma unroll
  for (int j = 0; j < ConvALoader<Cfg>::NF4; ++j) {
    const int row = (threadIdx.x + 256 * j) / ConvALoader<Cfg>::F4;
    const int m = m0 + row;
    if (m < M) {
      const int b = m / HW, pix = m % HW;
      la.py[j] = pix / a.W; la.px[j] = pix % a.W; la.pb[j] = (int64_t)b * HW;
    } else {
      la.py[j] = -1; la.px[j] = 0; la.pb[j] = 0;
    }
  }
  RowMajorTileLoader<Cfg::BN, Cfg::BK, Cfg::LDB> lb{a.wpk + (int64_t)n0 * a.Ktot, a.Ktot, a.N - n0, a.Ktot};
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  gemm_mainloop<Cfg>(lds, a.Ktot / Cfg::BK, la, lb, acc);
  conv_epilogue<Cfg, EPI>(a, acc, m0, n0);
}

// split-bf16 variant: 3 x v_mfma_f32_32x32x16_bf16 per product block, weights pre-split at pack time
template <class Cfg, int EPI, int BUF = 0>       // BUF 0: flat 64-bit addressing, 1: buffer loads + k-tile table, 2: uniform-pitch table
__global__ __launch_bounds__(Cfg::NT) void conv_igemm_split_kernel(const std::conditional_t<BUF != 0, ConvArgsT, ConvArgs> args) {
  static_assert(Cfg::NT == 256 || BUF != 0, "wider workgroups use the buffer-addressed loaders");
  __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS_ALLOC];
  const ConvArgs& a = [&]() -> const ConvArgs& { if constexpr (BUF != 0) return args.a; else return args; }();
  const int HW = a.H * a.W;
  const int M = a.B * HW;
  int bx = blockIdx.x, by = blockIdx.y;
  tile_of_block(a.swz, bx, by);
  const int n0 = bx * Cfg::BN, m0 = by * Cfg::BM;
  float sa, inv;
  conv_scales(a, sa, inv);
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if constexpr (BUF != 0) {
    const int PH = a.PH, PW = a.PW, taps = a.KH * a.KW;
    // Index the table where it lives, in the kernarg segment: going through the by-value struct would make the
    // compiler copy it to scratch (dynamic index), and a scratch load is per-lane, i.e. no longer provably uniform.
    const auto* ktab = (const unsigned __attribute__((address_space(4)))*)(
        (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ConvArgsT, ktab));
    // split-K (small M): this workgroup's slice of the k-tiles; the table and the weight rows are entered kt0 tiles further on
    int kt0 = 0, KTs = a.Ktot / 32;
    if (a.ksplit > 1) {
      const int per = (KTs + a.ksplit - 1) / a.ksplit;
      kt0 = (int)blockIdx.z * per;
      KTs = min(per, KTs - kt0);
      ktab += kt0 * (BUF >= 2 ? 2 : 1);
    }
    unsigned tapmask[Cfg::NCH_A], pofs[Cfg::NCH_A];
#pragma unroll
    for (int j = 0; j < Cfg::NCH_A; ++j) {
      const int m = m0 + ((threadIdx.x + Cfg::NT * j) >> 3);
      unsigned mask = 0;
      if (m < M) {
        const int pix = m % HW, y = pix / a.W, x = pix % a.W;
        for (int t = 0; t < taps; ++t) {
          const int yy = y + t / a.KW - PH, xx = x + t % a.KW - PW;
          if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) mask |= 1u << t;
        }
      }
      tapmask[j] = mask; pofs[j] = m < M ? m : 0;
    }
    BufWeightLoader<Cfg> lb;
    lb.base = uni_ptr(reinterpret_cast<const char*>(a.wpk) + (int64_t)n0 * a.Ktot * 4 + (int64_t)kt0 * 128);
    lb.nbytes = 0x7fffffffu;   // validity is carried by the lane offsets alone (FS_OOB), whatever the range check adds to them
#pragma unroll
    for (int j = 0; j < BufWeightLoader<Cfg>::NCH; ++j) {
      const int e = threadIdx.x + Cfg::NT * j;
      lb.voff[j] = (e >> 3) < a.N - n0 ? (unsigned)((e >> 3) * a.Ktot * 4 + (e & 7) * 16) : FS_OOB;
    }
    if constexpr (BUF >= 2) {
      BufConvALoaderU<Cfg> la;
      la.base = uni_ptr(args.ubase); la.ktab = ktab; la.kq16 = (threadIdx.x & 7) * 16; la.s = sa;
      const unsigned ldb = uni((unsigned)args.uld * 4u);
#pragma unroll
      for (int j = 0; j < Cfg::NCH_A; ++j) { la.tapmask[j] = tapmask[j]; la.voff0[j] = pofs[j] * ldb + la.kq16; }
      split_mainloop<Cfg, BufConvALoaderU<Cfg>, BufWeightLoader<Cfg>, true>(lds, KTs, la, lb, acc);
    } else {
      BufConvALoader<Cfg> la;
      la.ld0x4 = uni(a.src[0].ld * 4); la.ld1x4 = uni(a.src[1].ld * 4); la.ld2x4 = uni(a.src[2].ld * 4);
      // descriptor base = tensor base - (PH*W + PW) pixels, so the per-tap SGPR offset (dy*W + dx)*ld is >= 0
      la.b0 = uni_ptr(a.src[0].p - (int64_t)(PH * a.W + PW) * a.src[0].ld); la.nb0 = 0x7fffffffu;
      la.b1 = uni_ptr(a.src[1].p - (int64_t)(PH * a.W + PW) * a.src[1].ld); la.nb1 = 0x7fffffffu;
      la.b2 = uni_ptr(a.src[2].p - (int64_t)(PH * a.W + PW) * a.src[2].ld); la.nb2 = 0x7fffffffu;
      la.ktab = ktab;
      la.kq16 = (threadIdx.x & 7) * 16; la.s = sa;
#pragma unroll
      for (int j = 0; j < Cfg::NCH_A; ++j) { la.tapmask[j] = tapmask[j]; la.pofs[j] = pofs[j]; }
      split_mainloop<Cfg, BufConvALoader<Cfg>, BufWeightLoader<Cfg>, true>(lds, KTs, la, lb, acc);
    }
    if constexpr (EPI == EPI_PLAIN && Cfg::LDS_ALLOC >= Cfg::BM * (Cfg::BN + 4) * 4) {
      if (a.ksplit > 1) {               // first pass of the split-K route: the raw partial tile -> slab blockIdx.z of the workspace
        conv_epilogue_lds<Cfg, EPI_PLAIN>(a, acc, m0, n0, reinterpret_cast<float*>(lds), true, 0, 0, 0, (int64_t)blockIdx.z * M, inv);
        return;
      }
    }
  } else {
  SplitConvALoader<Cfg> la;
  la.p0 = a.src[0].p; la.C0 = a.src[0].C; la.ld0 = a.src[0].ld; la.cpt0 = (a.src[0].C + 31) / 32;
  la.p1 = a.src[1].p; la.C1 = a.src[1].C; la.ld1 = a.src[1].ld; la.cpt1 = a.nsrc > 1 ? (a.src[1].C + 31) / 32 : 0;
  la.p2 = a.src[2].p; la.C2 = a.src[2].C; la.ld2 = a.src[2].ld; la.cpt2 = a.nsrc > 2 ? (a.src[2].C + 31) / 32 : 0;
  if (a.nsrc < 2) { la.p1 = a.src[0].p; la.C1 = 0; la.ld1 = 4; la.cpt1 = 1; }
  if (a.nsrc < 3) { la.p2 = a.src[0].p; la.C2 = 0; la.ld2 = 4; la.cpt2 = 1; }
  la.taps = a.KH * a.KW; la.KW = a.KW; la.PH = a.PH; la.PW = a.PW; la.H = a.H; la.W = a.W; la.s = sa;
#pragma unroll
  for (int j = 0; j < SplitConvALoader<Cfg>::NCH; ++j) {
    const int m = m0 + ((threadIdx.x + 256 * j) >> 3);
    if (m < M) {
      const int pix = m % HW;
      la.py[j] = pix / a.W; la.px[j] = pix % a.W; la.pofs[j] = m;
    } else {
      la.py[j] = -(1 << 20); la.px[j] = 0; la.pofs[j] = 0;      // fails every (unsigned)(py+dy) < H test
    }
  }
  SplitWeightLoader<Cfg> lb{reinterpret_cast<const char*>(a.wpk) + (int64_t)n0 * a.Ktot * 4, (int64_t)a.Ktot * 4, a.N - n0};
  split_mainloop<Cfg>(lds, a.Ktot / 32, la, lb, acc);
  }
  if constexpr (Cfg::LDS_ALLOC >= Cfg::BM * (Cfg::BN + 4) * 4) {     // the parked tile must fit the LDS allocation
    if (EPI != EPI_PLAIN || epilogue_rows_ok(a)) {
      conv_epilogue_lds<Cfg, EPI>(a, acc, m0, n0, reinterpret_cast<float*>(lds), true, 0, 0, 0, 0, inv);
      return;
    }
  }
  conv_epilogue<Cfg, EPI>(a, acc, m0, n0, inv);
}

// ---------------------------------------------------------------- 3x3 convolution over a resident input patch
// The implicit GEMM above re-reads every input pixel once per tap: a few-channel 3x3 layer at encoder resolution
// (64 -> 64 at 8 x 220 x 512) moves 9 x 231 MB through the load path for 2 x 0.2 GB of unique data and is bound by
// bytes in flight, not by the matrix pipe.  Here a workgroup owns a 4 x 32 patch of output pixels: it loads the
// 6 x 34 input patch (halo included) ONCE, splits it to [hi | lo] bf16 and keeps it in LDS (one 128-byte record per
// pixel and 32-channel group, same swizzle as the A image of split_mainloop); the k-loop then walks (tap, channel
// group), reading the A fragments straight out of the patch at the tap's row shift, and streams only the weight
// tiles.  Global traffic per output pixel drops from 9 to 1.6 input pixels and the per-k-tile staging to the B tile.
constexpr int HALO_TH = 4, HALO_TW = 32, HALO_PW = HALO_TW + 2, HALO_ROWS = (HALO_TH + 2) * HALO_PW;
struct HaloArgs {
  const float* x; int ldx, C;
  const char* wpk;                    // fragment-order split pack (fsraft_conv_desc.wpk_frag)
  const float* bias;
  float* out; int64_t obs, ops;       // batch / pixel strides of the destination (floats), channels contiguous
  int N, B, H, W, relu;
  int acc;                            // out += result (the data gradient of a residual unit's first convolution adds to the shortcut gradient)
  float* st_sum; float* st_sq; int st_slots;   // as in ConvArgs (NULL: none)
  const unsigned* xamax; const unsigned* wamax; unsigned* damax;   // as ConvArgs::samax / wamax / damax
};

template <int CG, int TN>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const HaloArgs a) {
  constexpr int PLANE = HALO_ROWS * 128;
  constexpr int NCHH = (HALO_ROWS * CG * 8 + 255) / 256, KT = 9 * CG;
  __shared__ __attribute__((aligned(16))) char lds[CG * PLANE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
  const int x0 = blockIdx.x * HALO_TW, y0 = blockIdx.y * HALO_TH, b = blockIdx.z;
  const float sa = fs_scale_of_amax(fs_amax_load(a.xamax));
  const float inv = fs_inv_scale(sa) * fs_inv_scale(fs_scale_of_amax(fs_amax_load(a.wamax)));

  // Weight fragments come straight from the packed matrix (L1/L2-resident: 4 * N * Ktot bytes for the whole grid): lane
  // (n, k half) reads its 16 bytes of hi and of lo for both k-steps of a k-tile.  No B image in LDS, hence no barrier in
  // the k-loop: once the patch is staged the waves run independently.
  // (a.wpk is the FRAGMENT-ORDER pack: [k-tile][32-column block][hi s0, hi s1, lo s0, lo s1][lane][16 B], so that one
  // wave-load is 1 KB of consecutive bytes -- read out of the row-major pack the same fragment touches 32 cache lines
  // and the kernel becomes L1-bound: 352 us instead of 270 us with an LDS-staged B on the 64 -> 64 encoder layer.)
  const int NB = (a.N + 31) >> 5;
  const char* wbase = uni_ptr(a.wpk);
  struct BFrag { u32x4 v[TN][4]; };             // [nt][hi s0, hi s1, lo s0, lo s1]
  auto fetch_b = [&](int kt, BFrag& f) {
    const int ku = __builtin_amdgcn_readfirstlane(kt < KT ? kt : KT - 1);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(wbase, 0x7fffffffu);
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int nb = wn * TN + nt;
      const unsigned soff = (unsigned)((ku * NB + (nb < NB ? nb : 0)) * 4096);
      const unsigned voff = nb < NB ? (unsigned)lane * 16u : FS_OOB;
#pragma unroll
      for (int q = 0; q < 4; ++q) f.v[nt][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + 1024u * q, 0);
    }
  };
  // Four fragment sets, three k-tiles of look-ahead (an L2 hit is several k-tiles of MFMA time away): 238 -> 231 us.
  // PMC on this kernel: 6.9 VALU instructions per MFMA, 40 % of them in the patch conversion prologue and the 4-byte
  // store epilogue, waves waiting 29 % of their lifetime.
  BFrag f[4];
  fetch_b(0, f[0]);
  fetch_b(1, f[1]);
  fetch_b(2, f[2]);

  // the input patch: chunk e = (patch row, channel group, 16-byte part); consecutive lanes read consecutive bytes
  {
    const float* img = uni_ptr(a.x + (int64_t)b * a.H * a.W * a.ldx);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(img, 0x7fffffffu);
    float rh[NCHH * 4];
#pragma unroll
    for (int j = 0; j < NCHH; ++j) {
      const int e = threadIdx.x + 256 * j;
      const int row = e / (CG * 8), rem = e % (CG * 8), ch = (rem >> 3) * 32 + (rem & 7) * 4;
      const int y = y0 - 1 + row / HALO_PW, x = x0 - 1 + row % HALO_PW;
      const bool ok = row < HALO_ROWS && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W && ch < a.C;
      const unsigned voff = ok ? (unsigned)((y * a.W + x) * a.ldx + ch) * 4u : FS_OOB;
      const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
      rh[4 * j + 0] = f[0]; rh[4 * j + 1] = f[1]; rh[4 * j + 2] = f[2]; rh[4 * j + 3] = f[3];
    }
#pragma unroll
    for (int j = 0; j < NCHH; ++j) {
      const int e = threadIdx.x + 256 * j;
      const int row = e / (CG * 8), rem = e % (CG * 8);
      if (row < HALO_ROWS) stage_convert<128>(lds + (rem >> 3) * PLANE, row * 8 + (rem & 7), rh + 4 * j, sa);
    }
  }
  __syncthreads();

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int kt, const BFrag& f) {
    const int ku = __builtin_amdgcn_readfirstlane(kt);
    const int tap = ku / CG, g = ku - tap * CG, dy = tap / 3, dx = tap - dy * 3;
    const char* Ap = lds + g * PLANE;
    const int rowa = (2 * wm + dy) * HALO_PW + dx + l31;          // patch row of this lane for mt = 0; mt = 1 is one patch line down
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      p16x8 ah[2], al[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = rowa + mt * HALO_PW;
        ah[mt] = *reinterpret_cast<const p16x8*>(Ap + slot_offset<128>(row, 2 * s + lh));
        al[mt] = *reinterpret_cast<const p16x8*>(Ap + slot_offset<128>(row, 4 + 2 * s + lh));
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          const p16x8 bh = __builtin_bit_cast(p16x8, f.v[nt][s]), bl = __builtin_bit_cast(p16x8, f.v[nt][2 + s]);
          acc[mt][nt] = fs_mfma_32x32x16(al[mt], bh, acc[mt][nt]);
          acc[mt][nt] = fs_mfma_32x32x16(ah[mt], bl, acc[mt][nt]);
          acc[mt][nt] = fs_mfma_32x32x16(ah[mt], bh, acc[mt][nt]);
        }
    }
  };
  for (int kt = 0; kt < KT; kt += 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (kt + i < KT) {
        fetch_b(kt + i + 3, f[(i + 3) & 3]);      // (past the end: re-reads the last tile; unused)
        compute(kt + i, f[i]);
      }
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] *= inv;
  if (a.st_sum) {
    // InstanceNorm statistics of the result: a lane owns column n of 2 x 16 pixels per nt; the two lane halves and the two
    // pixel-row waves (wm) meet through shuffles / LDS (the patch is no longer needed), one atomic pair per column and workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);          // [wm][2][64 * TN]
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int py = y0 + 2 * wm + mt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float v = (py < a.H && px < a.W) ? acc[mt][nt][r] : 0.f;
          s1 += v; s2 += v * v;
        }
      }
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      const int col = wn * (32 * TN) + nt * 32 + l31;
      if (lh == 0) { red[(wm * 2 + 0) * (64 * TN) + col] = s1; red[(wm * 2 + 1) * (64 * TN) + col] = s2; }
    }
    __syncthreads();
    if ((int)threadIdx.x < 64 * TN && (int)threadIdx.x < a.N) {
      const int col = threadIdx.x;
      const int slot = (int)((blockIdx.y * gridDim.x + blockIdx.x) % (unsigned)a.st_slots);
      const int64_t o = ((int64_t)b * a.st_slots + slot) * a.N + col;
      atomicAdd(a.st_sum + o, red[col] + red[2 * (64 * TN) + col]);
      atomicAdd(a.st_sq + o, red[(64 * TN) + col] + red[3 * (64 * TN) + col]);
    }
  }
  float* outb = a.out + (int64_t)b * a.obs;
  unsigned mx = 0u;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int py = y0 + 2 * wm + mt;
    if (py >= a.H) continue;
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int n = wn * (32 * TN) + nt * 32 + l31;
      if (n >= a.N) continue;
      const float bv = a.bias ? a.bias[n] : 0.f;
      float old[16];                               // accumulating epilogue: the 16 old values are requested together
#pragma unroll                                     // (`*o = acc ? *o + v : v` is a branch and a round trip per element)
      for (int r = 0; r < 16; ++r) {
        const int px = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh, pc = px < a.W ? px : a.W - 1;
        old[r] = a.acc ? outb[(int64_t)(py * a.W + pc) * a.ops + n] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (px >= a.W) continue;
        float v = acc[mt][nt][r] + bv;
        if (a.relu) v = fmaxf(v, 0.f);
        v += old[r];
        outb[(int64_t)(py * a.W + px) * a.ops + n] = v;
        mx = fs_umax(mx, fs_abs_bits(v));
      }
    }
  }
  if (a.damax) fs_amax_commit(a.damax, mx, reinterpret_cast<unsigned*>(lds));
}

template <int CG, int TN>
int launch_halo(const HaloArgs& h, hipStream_t s) {
  dim3 grid(ceil_div(h.W, HALO_TW), ceil_div(h.H, HALO_TH), h.B);
  hipLaunchKernelGGL((conv3x3_halo_kernel<CG, TN>), grid, dim3(256), 0, s, h);
  return fs_launch_status();
}


// ---------------------------------------------------------------- weight gradient
struct WgradArgs {
  const float* dy; int ldy; int Cout;     // dY (already multiplied by act'), [M][ldy]
  Src src[3]; int nsrc;
  float* dwpk; int Ktot;
  int B, H, W, KH, KW;
  int kchunk;                              // pixels per split (multiple of 32)
  float* dbias;                            // optional: dbias[co] += sum_pixels dY[pixel][co] (fused in the split kernel)
  // XCD-aware launch (xcd_xt > 0): 1-D grid; the xcd_xt packed-K tiles that read the SAME dY tile -- one (Cout tile, pixel
  // split) group -- get linear ids 8 apart, i.e. the same XCD and its L2, instead of being dealt round-robin over all 8.
  int xcd_xt, xcd_yt, xcd_groups;
  const unsigned* dyamax; const unsigned* samax[3];     // amax words of dY and of the sources (NULL: scale 1), single-segment launches
};

// Several (dY, X) pairs of identical shape in one launch -- the 12 iterations of a step: dW = sum_t dY_t^T X_t is one
// reduction over 12 x M pixels, so the per-launch prologue / atomic epilogue is paid once per step instead of once
// per iteration.  blockIdx.z = segment * zs + pixel split.  The pointer tables are read from the kernarg segment.
constexpr int WGRAD_MAX_SEG = 16;
struct WgradArgsM {
  WgradArgs a;
  int nseg, zs;
  const float* dys[WGRAD_MAX_SEG];
  const float* srcs[3][WGRAD_MAX_SEG];
  const unsigned* dyam[WGRAD_MAX_SEG];        // amax words per segment (NULL: scale 1); a launch uses ONE scale per operand,
  const unsigned* sam[3][WGRAD_MAX_SEG];      // from the largest word over its segments (their products share accumulators)
};

// scales of the weight gradient's two operands: dY (all segments) and source s (all segments)
template <bool MULTI, class Args>
__device__ __forceinline__ void wgrad_scales(const Args& args, const WgradArgs& a, int s, size_t table_base, float& sdy, float& sx) {
  unsigned mdy, mx;
  if constexpr (MULTI) {
    typedef const unsigned* uptr;
    const auto* karg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + table_base;
    const auto* tdy = (const uptr __attribute__((address_space(4)))*)(karg + offsetof(WgradArgsM, dyam));
    const auto* tsx = (const uptr __attribute__((address_space(4)))*)(karg + offsetof(WgradArgsM, sam)) + s * WGRAD_MAX_SEG;
    mdy = 0u; mx = 0u;
    for (int i = 0; i < args.nseg; ++i) {
      mdy = fs_umax(mdy, fs_amax_load(tdy[i]));
      mx = fs_umax(mx, fs_amax_load(tsx[i]));
    }
  } else {
    mdy = fs_amax_load(a.dyamax);
    mx = fs_amax_load(s == 0 ? a.samax[0] : s == 1 ? a.samax[1] : a.samax[2]);
  }
  sdy = fs_scale_of_amax(mdy);
  sx = fs_scale_of_amax(mx);
}

template <class Cfg>
struct ShiftedXLoader {                    // Bs[k = pixel][n = ci] <- X[pixel + off][ci0 + n]
  static constexpr int BN = Cfg::BN, BK = Cfg::BK, LD = Cfg::LDB;
  static constexpr int F4 = BN / 4;
  static constexpr int NF4 = BK * F4 / 256;
  static constexpr int NREG = NF4 * 4;
  static constexpr int NCH = NF4;
  const float* p; int ld, cvalid;          // p already offset by ci0; cvalid = channels left from ci0
  int dy, dx, H, W, HW; int64_t M; int64_t m_begin, m_end;
  __device__ __forceinline__ bool fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int k = e / F4, c4 = e % F4;
    const int64_t m = m_begin + (int64_t)kt * BK + k;
    const int64_t mm = m < m_end ? m : m_begin;
    const int64_t b = mm / HW; const int pix = (int)(mm % HW);
    const int yy = pix / W + dy, xx = pix % W + dx;
    const bool ok = m < m_end && c4 * 4 < cvalid && yy >= 0 && yy < H && xx >= 0 && xx < W;
    const f32x4 v = gload4(p + (ok ? (b * HW + (int64_t)yy * W + xx) * ld + c4 * 4 : 0));
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
    return ok;
  }
  __device__ __forceinline__ void store_chunk(float* t, const float (&r)[NREG], int j, bool ok) const {
    const int e = threadIdx.x + 256 * j;
    const int k = e / F4, c4 = e % F4;
    f32x4 v = {r[4 * j + 0], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]};
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(t + k * LD + c4 * 4) = ok ? v : z;
  }
};

template <class Cfg>
struct DyLoader {                          // As[k = pixel][m = co] <- dY[pixel][co0 + m]
  static constexpr int BM = Cfg::BM, BK = Cfg::BK, LD = Cfg::LDA;
  static constexpr int F4 = BM / 4;
  static constexpr int NF4 = BK * F4 / 256;
  static constexpr int NREG = NF4 * 4;
  static_assert((BK * F4) % 256 == 0, "dy tile must divide over 256 threads");
  static constexpr int NCH = NF4;
  const float* p; int ld, cvalid; int64_t m_begin, m_end;
  __device__ __forceinline__ bool fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int k = e / F4, c4 = e % F4;
    const int64_t m = m_begin + (int64_t)kt * BK + k;
    const bool ok = m < m_end && c4 * 4 < cvalid;
    const f32x4 v = gload4(p + (ok ? m * ld + c4 * 4 : 0));
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
    return ok;
  }
  __device__ __forceinline__ void store_chunk(float* t, const float (&r)[NREG], int j, bool ok) const {
    const int e = threadIdx.x + 256 * j;
    const int k = e / F4, c4 = e % F4;
    f32x4 v = {r[4 * j + 0], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]};
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(t + k * LD + c4 * 4) = ok ? v : z;
  }
};

// ---- split-bf16 weight gradient (k-major operands, transposed LDS reads) ---------------------
template <class Cfg>
struct SplitDyLoader {                    // chunk e: pixel k = e / 32, channels 4*(e % 32) .. +3 of the 128-wide co tile
  static constexpr int NCH = Cfg::NCH_A, NREG = NCH * 4;
  const float* p; int ld, cvalid; int64_t m_begin, m_end;
  __device__ __forceinline__ void fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int k = e / (Cfg::BM / 4), c4 = e % (Cfg::BM / 4);
    const int64_t m = m_begin + (int64_t)kt * 32 + k;
    const bool ok = m < m_end && c4 * 4 < cvalid;
    const f32x4 v = gload4(ok ? p + m * ld + c4 * 4 : g_fsraft_zero16);
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
  }
};
template <class Cfg>
struct SplitShiftedXLoader {
  static constexpr int NCH = Cfg::NCH_B, NREG = NCH * 4;
  const float* p; int ld, cvalid;
  int dy, dx, H, W, HW; int64_t m_begin, m_end;
  __device__ __forceinline__ void fetch_chunk(int kt, float (&r)[NREG], int j) const {
    const int e = threadIdx.x + 256 * j;
    const int k = e / (Cfg::BN / 4), c4 = e % (Cfg::BN / 4);
    const int64_t m = m_begin + (int64_t)kt * 32 + k;
    const int64_t mm = m < m_end ? m : m_begin;
    const int pix = (int)(mm % HW);
    const int yy = pix / W + dy, xx = pix % W + dx;
    const bool ok = m < m_end && c4 * 4 < cvalid && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    const f32x4 v = gload4(ok ? p + (mm + dy * W + dx) * ld + c4 * 4 : g_fsraft_zero16);
    r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
  }
};

// buffer-addressed variants of the two loaders above (see BufConvALoader): per-lane offsets are fixed for the whole
// k-loop, the k-tile advance is one SGPR offset, and the "shifted pixel inside the image" test -- three integer
// divisions per 16-byte chunk in the loaders above, ~300 VALU instructions per k-tile -- is a bit test against a
// per-workgroup pixel mask that is computed once (one word per k-tile, in LDS).
constexpr int WGRAD_MASK_WORDS = 2048;     // pixels per workgroup / 32 (host caps the pixel split at 65536)
template <class Cfg>
struct BufDyLoader {
  static constexpr int NCH = Cfg::NCH_A, NREG = NCH * 4;
  const float* base; unsigned ld4; int npix;
  unsigned voff[NCH]; int krow[NCH];
  __device__ __forceinline__ void fetch_tile(int kt, float (&r)[NREG]) const {
    const int ku = __builtin_amdgcn_readfirstlane(kt);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, 0x7fffffffu);
    const unsigned soff = (unsigned)ku * 32u * ld4;
    const int left = npix - ku * 32;                              // rows of this tile that exist
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const unsigned voffj = voff[j] | (krow[j] < left ? 0u : FS_OOB);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voffj, soff, 0);
      const f32x4 f = __builtin_bit_cast(f32x4, v);
      r[4 * j + 0] = f[0]; r[4 * j + 1] = f[1]; r[4 * j + 2] = f[2]; r[4 * j + 3] = f[3];
    }
  }
};
template <class Cfg>
struct BufShiftedXLoader {
  static constexpr int NCH = Cfg::NCH_B, NREG = NCH * 4;
  const float* base; unsigned ld4; const unsigned* mask;       // mask: LDS, bit k of word kt = pixel mb + 32 kt + k usable
  unsigned voff[NCH]; int krow[NCH];
  __device__ __forceinline__ void fetch_tile(int kt, float (&r)[NREG]) const {
    const int ku = __builtin_amdgcn_readfirstlane(kt);
    const unsigned w = __builtin_amdgcn_readfirstlane(mask[ku]);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, 0x7fffffffu);
    const unsigned soff = (unsigned)ku * 32u * ld4;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const unsigned voffj = voff[j] | ((__builtin_amdgcn_ubfe(w, (unsigned)krow[j], 1u) ^ 1u) << 31);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voffj, soff, 0);
      const f32x4 f = __builtin_bit_cast(f32x4, v);
      r[4 * j + 0] = f[0]; r[4 * j + 1] = f[1]; r[4 * j + 2] = f[2]; r[4 * j + 3] = f[3];
    }
  }
};

using SWCfg128 = SplitTnCfg<128, 128, 2, 2, 2>;
using SWCfg128S = SplitTnCfg<128, 128, 2, 2, 1>;
using SWCfg128W8 = SplitTnCfg<128, 128, 2, 4, 1, 512>;   // eight waves per workgroup (multi-segment launches)

template <class Cfg, bool BUF = false, bool MULTI = false>
__global__ __launch_bounds__(Cfg::NT) void conv_wgrad_split_kernel(const std::conditional_t<MULTI, WgradArgsM, WgradArgs> args) {
  static_assert(Cfg::NT == 256 || BUF, "512-thread workgroups use the buffer-addressed loaders");
  __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS_BYTES];
  __shared__ unsigned pixmask[BUF ? WGRAD_MASK_WORDS : 1];
  const WgradArgs& a = [&]() -> const WgradArgs& { if constexpr (MULTI) return args.a; else return args; }();
  int bx = blockIdx.x, by = blockIdx.y, zblock = blockIdx.z, seg = 0;
  if (a.xcd_xt > 0) {
    const int span = 8 * a.xcd_xt, r = blockIdx.x / span, rem = blockIdx.x - r * span;
    const int g = r * 8 + (rem & 7);
    if (g >= a.xcd_groups) return;
    if (a.xcd_yt > 0) { bx = rem >> 3; by = g % a.xcd_yt; zblock = g / a.xcd_yt; }        // group = (Cout tile, pixel split)
    else { const int tile = rem >> 3, x0 = -a.xcd_yt; bx = tile % x0; by = tile / x0; zblock = g; }   // group = pixel split
  }
  if constexpr (MULTI) { seg = zblock / args.zs; zblock -= seg * args.zs; seg = __builtin_amdgcn_readfirstlane(seg); }
  const int HW = a.H * a.W;
  const int64_t M = (int64_t)a.B * HW;
  const int taps = a.KH * a.KW;
  int t = bx, s = 0, kofs = 0;
  for (;; ++s) {
    const int ct = (a.src[s].C + Cfg::BN - 1) / Cfg::BN;
    if (t < taps * ct) break;
    t -= taps * ct;
    kofs += taps * ((a.src[s].C + 31) / 32) * 32;
  }
  Src sc = s == 0 ? a.src[0] : s == 1 ? a.src[1] : a.src[2];
  const float* dyp = a.dy;
  if constexpr (MULTI) {       // this segment's tensors, read from the kernarg tables with a uniform index
    typedef const float* fptr;
    const auto* karg = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    dyp = ((const fptr __attribute__((address_space(4)))*)(karg + offsetof(WgradArgsM, dys)))[seg];
    sc.p = ((const fptr __attribute__((address_space(4)))*)(karg + offsetof(WgradArgsM, srcs)))[s * WGRAD_MAX_SEG + seg];
  }
  const int ct = (sc.C + Cfg::BN - 1) / Cfg::BN;
  const int tap = t / ct, ci0 = (t % ct) * Cfg::BN;
  const int cpad = ((sc.C + 31) / 32) * 32;
  kofs += tap * cpad + ci0;
  const int co0 = by * Cfg::BM;
  const int64_t mb = (int64_t)zblock * a.kchunk;
  const int64_t me = mb + a.kchunk < M ? mb + a.kchunk : M;
  if (mb >= M) return;
  const int coleft = ((a.Cout + 3) / 4) * 4 - co0;     // dy may be a channel slice of a wider buffer: never read past it
  const int cleft = ((sc.C + 3) / 4) * 4 - ci0;
  const int dyy = tap / a.KW - a.KH / 2, dxx = tap % a.KW - a.KW / 2;
  const int KT = (int)((me - mb + 31) / 32);
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_bias = a.dbias != nullptr && bx == 0;        // one x-tile per (co tile, pixel split) owns the bias
  float sdy, sx;
  wgrad_scales<MULTI>(args, a, s, 0, sdy, sx);
  const float inv = fs_inv_scale(sdy) * fs_inv_scale(sx);
  if constexpr (BUF) {
    // pixel mask: bit k of word w <=> pixel mb + 32 w + k exists and its (dy, dx)-shifted neighbour is inside the image
    for (int i = threadIdx.x; i < KT * 32; i += Cfg::NT) {
      const int64_t m = mb + i;
      bool ok = m < me;
      if (ok) {
        const int pix = (int)(m % HW), yy = pix / a.W + dyy, xx = pix % a.W + dxx;
        ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
      }
      const unsigned long long bal = __ballot(ok);
      if ((threadIdx.x & 63) == 0) { pixmask[i >> 5] = (unsigned)bal; pixmask[(i >> 5) + 1] = (unsigned)(bal >> 32); }
    }
    __syncthreads();
    BufDyLoader<Cfg> la;
    la.base = uni_ptr(dyp + co0 + mb * a.ldy); la.ld4 = uni((unsigned)a.ldy * 4u); la.npix = (int)(me - mb);
    const int cva = coleft < Cfg::BM ? coleft : Cfg::BM;
#pragma unroll
    for (int j = 0; j < BufDyLoader<Cfg>::NCH; ++j) {
      const int e = threadIdx.x + Cfg::NT * j, k = e / (Cfg::BM / 4), c4 = e % (Cfg::BM / 4);
      la.krow[j] = k; la.voff[j] = c4 * 4 < cva ? (unsigned)(k * a.ldy + c4 * 4) * 4u : FS_OOB;
    }
    BufShiftedXLoader<Cfg> lb;
    lb.base = uni_ptr(sc.p + ci0 + (mb + dyy * a.W + dxx) * sc.ld); lb.ld4 = uni((unsigned)sc.ld * 4u); lb.mask = pixmask;
    const int cvb = cleft < Cfg::BN ? cleft : Cfg::BN;
#pragma unroll
    for (int j = 0; j < BufShiftedXLoader<Cfg>::NCH; ++j) {
      const int e = threadIdx.x + Cfg::NT * j, k = e / (Cfg::BN / 4), c4 = e % (Cfg::BN / 4);
      lb.krow[j] = k; lb.voff[j] = c4 * 4 < cvb ? (unsigned)(k * sc.ld + c4 * 4) * 4u : FS_OOB;
    }
    if (want_bias) split_mainloop_tn<Cfg, BufDyLoader<Cfg>, BufShiftedXLoader<Cfg>, true>(lds, KT, la, lb, acc, colsum, sdy, sx);
    else split_mainloop_tn<Cfg>(lds, KT, la, lb, acc, nullptr, sdy, sx);
  } else {
  SplitDyLoader<Cfg> la{dyp + co0, a.ldy, coleft < Cfg::BM ? coleft : Cfg::BM, mb, me};
  SplitShiftedXLoader<Cfg> lb{sc.p + ci0, sc.ld, cleft < Cfg::BN ? cleft : Cfg::BN, dyy, dxx, a.H, a.W, HW, mb, me};
  if (want_bias) split_mainloop_tn<Cfg, SplitDyLoader<Cfg>, SplitShiftedXLoader<Cfg>, true>(lds, KT, la, lb, acc, colsum, sdy, sx);
  else split_mainloop_tn<Cfg>(lds, KT, la, lb, acc, nullptr, sdy, sx);
  }
  if (want_bias) {
    // this thread's columns are co0 + 4*(tid % 32) .. +3; NT/32 threads (tid / 32) share them
    float* part = reinterpret_cast<float*>(lds);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) part[(threadIdx.x >> 5) * Cfg::BM + 4 * (threadIdx.x & 31) + q] = colsum[q];
    __syncthreads();
    if (threadIdx.x < Cfg::BM) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < Cfg::NT / 32; ++g) s += part[g * Cfg::BM + threadIdx.x];
      if (co0 + threadIdx.x < a.Cout) atomicAdd(a.dbias + co0 + threadIdx.x, s);
    }
  }
#pragma unroll
  for (int nt = 0; nt < Cfg::TN; ++nt) {
    const int n = acc_col<Cfg>(nt);
    if (ci0 + n >= cpad) continue;
#pragma unroll
    for (int mt = 0; mt < Cfg::TM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + acc_row<Cfg>(mt, r);
        if (co < a.Cout) atomicAdd(a.dwpk + (int64_t)co * a.Ktot + kofs + n, acc[mt][nt][r] * inv);
      }
  }
}

// ---- few-channel layers (encoder residual stages, C = 32 / 64 / 96) -----------------------------------------------
// A 128-column x tile holding one tap of a 64-channel source is half empty, and so is a 128-row dY tile of a 64-channel
// output: the kernel above then spends 4x the useful MFMA work.  Here the x tile packs TP = BN / cpad CONSECUTIVE taps
// side by side (the packed-K layout is tap-major, so the tile's columns are one contiguous run of dW columns) and the
// dY tile is 64 wide.  A lane's chunk belongs to one tap for the whole k-loop: the tap's pixel shift is folded into its
// fixed buffer offset and its "neighbour inside the image" bit comes from that tap's own pixel mask.
constexpr int WGRAD_PACK_WORDS = 256;      // pixel-mask words per tap slot (pixel split <= 32 * 254)
constexpr int WGRAD_PACK_SLOTS = 8;
using SWCfgPack = SplitTnCfg<64, 192, 2, 2, 1>;

template <class Cfg>
struct BufPackedXLoader {
  static constexpr int NCH = Cfg::NCH_B, NREG = NCH * 4;
  const float* base; unsigned ld4; const unsigned* mask;       // mask[word * SLOTS + slot]
  unsigned voff[NCH]; int krow[NCH]; int slot[NCH];
  __device__ __forceinline__ void fetch_tile(int kt, float (&r)[NREG]) const {
    const int ku = __builtin_amdgcn_readfirstlane(kt);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, 0x7fffffffu);
    const unsigned soff = (unsigned)ku * 32u * ld4;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const unsigned w = mask[ku * WGRAD_PACK_SLOTS + slot[j]];
      const unsigned voffj = voff[j] | ((__builtin_amdgcn_ubfe(w, (unsigned)krow[j], 1u) ^ 1u) << 31);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voffj, soff, 0);
      const f32x4 f = __builtin_bit_cast(f32x4, v);
      r[4 * j + 0] = f[0]; r[4 * j + 1] = f[1]; r[4 * j + 2] = f[2]; r[4 * j + 3] = f[3];
    }
  }
};

// grid: x = tap group, y = 64-row Cout tile, z = pixel split.  Single source, split-bf16 arithmetic.
template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void conv_wgrad_pack_kernel(const WgradArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS_BYTES];
  __shared__ unsigned pixmask[WGRAD_PACK_WORDS * WGRAD_PACK_SLOTS];
  const int HW = a.H * a.W;
  const int64_t M = (int64_t)a.B * HW;
  const int taps = a.KH * a.KW;
  const Src sc = a.src[0];
  const int cpad = ((sc.C + 31) / 32) * 32;
  int TP = Cfg::BN / cpad;
  if (TP > WGRAD_PACK_SLOTS) TP = WGRAD_PACK_SLOTS;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd_xt > 0) {          // tap groups of one (Cout tile, pixel split) on one XCD: they read the same dY and x rows
    const int span = 8 * a.xcd_xt, r = blockIdx.x / span, rem = blockIdx.x - r * span;
    const int g = r * 8 + (rem & 7);
    if (g >= a.xcd_groups) return;
    bx = rem >> 3; by = g % a.xcd_yt; bz = g / a.xcd_yt;
  }
  const int tap0 = bx * TP;
  const int ntap = taps - tap0 < TP ? taps - tap0 : TP;
  const int kofs = tap0 * cpad;
  const int co0 = by * Cfg::BM;
  const int64_t mb = (int64_t)bz * a.kchunk;
  const int64_t me = mb + a.kchunk < M ? mb + a.kchunk : M;
  if (mb >= M) return;
  const int coleft = ((a.Cout + 3) / 4) * 4 - co0;
  const int cva = coleft < Cfg::BM ? coleft : Cfg::BM;
  const int cvb = ((sc.C + 3) / 4) * 4;
  const int KT = (int)((me - mb + 31) / 32);
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_bias = a.dbias != nullptr && bx == 0;
  float sdy, sx;
  wgrad_scales<false>(a, a, 0, 0, sdy, sx);
  const float inv = fs_inv_scale(sdy) * fs_inv_scale(sx);
  // pixel masks, one per tap slot: bit k of word w <=> pixel mb + 32 w + k exists and its shifted neighbour is inside the image
  for (int i = threadIdx.x; i < KT * 32; i += Cfg::NT) {
    const int64_t m = mb + i;
    const int pix = (int)(m % HW), py = pix / a.W, px = pix % a.W;
    for (int tp = 0; tp < ntap; ++tp) {
      const int tap = tap0 + tp;
      const int yy = py + tap / a.KW - a.KH / 2, xx = px + tap % a.KW - a.KW / 2;
      const bool ok = m < me && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
      const unsigned long long bal = __ballot(ok);
      if ((threadIdx.x & 63) == 0) {
        pixmask[(i >> 5) * WGRAD_PACK_SLOTS + tp] = (unsigned)bal;
        pixmask[((i >> 5) + 1) * WGRAD_PACK_SLOTS + tp] = (unsigned)(bal >> 32);
      }
    }
  }
  __syncthreads();
  BufDyLoader<Cfg> la;
  la.base = uni_ptr(a.dy + co0 + mb * a.ldy); la.ld4 = uni((unsigned)a.ldy * 4u); la.npix = (int)(me - mb);
#pragma unroll
  for (int j = 0; j < BufDyLoader<Cfg>::NCH; ++j) {
    const int e = threadIdx.x + Cfg::NT * j, k = e / (Cfg::BM / 4), c4 = e % (Cfg::BM / 4);
    la.krow[j] = k; la.voff[j] = c4 * 4 < cva ? (unsigned)(k * a.ldy + c4 * 4) * 4u : FS_OOB;
  }
  // taps ascend in (dy, dx), so the group's first tap has the smallest pixel shift: every lane offset is >= 0
  const int shift0 = (tap0 / a.KW - a.KH / 2) * a.W + (tap0 % a.KW - a.KW / 2);
  BufPackedXLoader<Cfg> lb;
  lb.base = uni_ptr(sc.p + (mb + shift0) * sc.ld); lb.ld4 = uni((unsigned)sc.ld * 4u); lb.mask = pixmask;
#pragma unroll
  for (int j = 0; j < BufPackedXLoader<Cfg>::NCH; ++j) {
    const int e = threadIdx.x + Cfg::NT * j, k = e / (Cfg::BN / 4), col = (e % (Cfg::BN / 4)) * 4;
    const int tp = col / cpad, cc = col - tp * cpad, tap = tap0 + tp;
    const int shift = (tap / a.KW - a.KH / 2) * a.W + (tap % a.KW - a.KW / 2) - shift0;
    const bool valid = tp < ntap && cc < cvb;
    lb.krow[j] = k; lb.slot[j] = valid ? tp : 0;
    lb.voff[j] = valid ? (unsigned)((k + shift) * sc.ld + cc) * 4u : FS_OOB;
  }
  if (want_bias) split_mainloop_tn<Cfg, BufDyLoader<Cfg>, BufPackedXLoader<Cfg>, true>(lds, KT, la, lb, acc, colsum, sdy, sx);
  else split_mainloop_tn<Cfg>(lds, KT, la, lb, acc, nullptr, sdy, sx);
  if (want_bias) {
    // this thread's dY columns are co0 + 4*(tid % (BM/4)) .. +3; NT / (BM/4) threads share them
    constexpr int Q = Cfg::BM / 4, G = Cfg::NT / Q;
    float* part = reinterpret_cast<float*>(lds);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) part[(threadIdx.x / Q) * Cfg::BM + 4 * (threadIdx.x % Q) + q] = colsum[q];
    __syncthreads();
    if (threadIdx.x < Cfg::BM) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) s += part[g * Cfg::BM + threadIdx.x];
      if (co0 + threadIdx.x < a.Cout) atomicAdd(a.dbias + co0 + threadIdx.x, s);
    }
  }
#pragma unroll
  for (int nt = 0; nt < Cfg::TN; ++nt) {
    const int n = acc_col<Cfg>(nt);
    if (n >= ntap * cpad) continue;
#pragma unroll
    for (int mt = 0; mt < Cfg::TM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + acc_row<Cfg>(mt, r);
        if (co < a.Cout) atomicAdd(a.dwpk + (int64_t)co * a.Ktot + kofs + n, acc[mt][nt][r] * inv);
      }
  }
}

// grid: x = packed-K tile (source, tap, 128-channel tile), y = Cout tile, z = pixel split
template <class Cfg>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  const int HW = a.H * a.W;
  const int64_t M = (int64_t)a.B * HW;
  const int taps = a.KH * a.KW;
  // decode blockIdx.x -> (source, tap, channel tile)
  int t = blockIdx.x, s = 0, kofs = 0;
  for (;; ++s) {
    const int ct = (a.src[s].C + Cfg::BN - 1) / Cfg::BN;
    if (t < taps * ct) break;
    t -= taps * ct;
    kofs += taps * ((a.src[s].C + 31) / 32) * 32;
  }
  const Src sc = s == 0 ? a.src[0] : s == 1 ? a.src[1] : a.src[2];
  const int ct = (sc.C + Cfg::BN - 1) / Cfg::BN;
  const int tap = t / ct, ci0 = (t % ct) * Cfg::BN;
  const int cpad = ((sc.C + 31) / 32) * 32;
  kofs += tap * cpad + ci0;
  const int co0 = blockIdx.y * Cfg::BM;
  const int64_t mb = (int64_t)blockIdx.z * a.kchunk;
  const int64_t me = mb + a.kchunk < M ? mb + a.kchunk : M;
  if (mb >= M) return;

  const int coleft = ((a.Cout + 3) / 4) * 4 - co0;     // dy may be a channel slice of a wider buffer: never read past it
  DyLoader<Cfg> la{a.dy + co0, a.ldy, coleft < Cfg::BM ? coleft : Cfg::BM, mb, me};
  const int cleft = ((sc.C + 3) / 4) * 4 - ci0;
  ShiftedXLoader<Cfg> lb{sc.p + ci0, sc.ld, cleft < Cfg::BN ? cleft : Cfg::BN,
                         tap / a.KW - a.KH / 2, tap % a.KW - a.KW / 2, a.H, a.W, HW, M, mb, me};

  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gemm_mainloop<Cfg>(lds, (int)((me - mb + Cfg::BK - 1) / Cfg::BK), la, lb, acc);

#pragma unroll
  for (int nt = 0; nt < Cfg::TN; ++nt) {
    const int n = acc_col<Cfg>(nt);
    if (ci0 + n >= cpad) continue;
#pragma unroll
    for (int mt = 0; mt < Cfg::TM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + acc_row<Cfg>(mt, r);
        if (co < a.Cout) atomicAdd(a.dwpk + (int64_t)co * a.Ktot + kofs + n, acc[mt][nt][r]);
      }
  }
}

// ---------------------------------------------------------------- weight (un)packing
// mode 0 (forward):  wpk[n][k(s,t,c)] = W[n][coff_s + c][t]                       n < Cout
// mode 1 (dgrad):    wpk[n][k(t',c)]  = W[c][n][taps-1-t']                        n < Cin_total, c < Cout
// mode 2 (unpack dW): W[n][coff_s + c][t] (+)= wpk[n][k(s,t,c)]   (inverse of mode 0)
struct PackArgs {
  float* w;            // OIHW [Cout][Cin][KH*KW]
  float* wpk;
  int Cout, Cin, taps;
  int C[3]; int nsrc;  // forward source split of Cin (mode 0/2); ignored for mode 1
  int Ktot, rows;
  int mode, accumulate;
  int split;           // modes 0/1: write [32 hi | 32 lo] fp16 records of w * scale instead of fp32 (same byte size)
  const unsigned* amax; // split: amax word of the weights (NULL: scale 1)
};

__device__ __forceinline__ void store_packed(const PackArgs& a, int64_t e, float v) {
  if (!a.split) { a.wpk[e] = v; return; }
  // element e = n*Ktot + k  ->  record (e / 32) of 64 shorts: hi at [k % 32], lo at [32 + k % 32]
  _Float16 h, l;
  fs_split1(v, fs_scale_of_amax(fs_amax_load(a.amax)), h, l);
  _Float16* rec = reinterpret_cast<_Float16*>(a.wpk) + (e >> 5) * 64;
  rec[e & 31] = h;
  rec[32 + (e & 31)] = l;
}

__global__ __launch_bounds__(256) void pack_weight_kernel(PackArgs a) {
  const int64_t total = (int64_t)a.rows * a.Ktot;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int n = (int)(e / a.Ktot);
    int k = (int)(e % a.Ktot);
    if (a.mode == 1) {
      const int cpad = ((a.Cout + 31) / 32) * 32;
      const int tp = k / cpad, c = k % cpad;
      float v = 0.f;
      if (c < a.Cout && n < a.Cin) v = a.w[((int64_t)c * a.Cin + n) * a.taps + (a.taps - 1 - tp)];
      store_packed(a, e, v);
    } else {
      int s = 0, coff = 0;
      for (; s < a.nsrc; ++s) {
        const int span = a.taps * (((a.C[s] + 31) / 32) * 32);
        if (k < span) break;
        k -= span; coff += a.C[s];
      }
      const int cpad = ((a.C[s] + 31) / 32) * 32;
      const int t = k / cpad, c = k % cpad;
      const bool ok = c < a.C[s] && n < a.Cout;
      if (a.mode == 0) {
        store_packed(a, e, ok ? a.w[((int64_t)n * a.Cin + coff + c) * a.taps + t] : 0.f);
      } else if (ok) {
        float* d = a.w + ((int64_t)n * a.Cin + coff + c) * a.taps + t;
        *d = a.accumulate ? *d + a.wpk[e] : a.wpk[e];
      }
    }
  }
}

// ---- batched packing: every GEMM-ready weight image of a module (and, in reverse, every weight gradient) in one launch ----
// One job = one packed matrix.  Its logical OIHW weight is read in place from the parameter tensors: up to three tensors
// stacked along the output channels (fused layers: z|r gates, flow-head|mask-head), GEMM sources = channel ranges of the
// parameters' input channels (a GRU convolution split into its (h, motion) part and its context part), optionally seen
// through the space-to-depth rewrite of a stride-2 3x3 weight.  (Mirror of fsraft_pack_job in include/fsraft.h.)
struct PackJob {
  float* w[3]; int rows[3]; int npiece;
  float* wpk;
  int cin_full, kh, kw;
  int srcC[3], srcOff[3], nsrc;
  int mode, flags;
  float scale; int accumulate;
  const unsigned* amax;          // modes 10 / 11: amax word of the job's weights (fsraft_amax_jobs over its pieces); NULL: scale 1
};
constexpr int PACK_JOBS = 16;
struct PackJobs { PackJob j[PACK_JOBS]; };

// logical element (output channel n, concatenated-source channel given as (source s, channel c), tap t of kh x kw) -> address
// in the parameter tensors, or nullptr where the logical weight is structurally zero (space-to-depth slots)
__device__ __forceinline__ float* pack_elem(const PackJob& j, int n, int s, int c, int t) {
  int p = 0;
  while (p + 1 < j.npiece && n >= j.rows[p]) { n -= j.rows[p]; ++p; }
  float* w = j.w[p];
  if (j.flags & 2) {
    // parameter [N][C][3][3] (stride 2, pad 1); logical [N][4C][2][2]: channel (sy*2+sx)*C + c0, tap (ty, tx);
    // input row 2y - 1 + ky = 2(y + ty - 1) + sy  ->  ky = 2*ty + sy - 1
    const int C = j.cin_full;
    const int sp = c / C, c0 = c % C;
    const int ky = 2 * (t >> 1) + (sp >> 1) - 1, kx = 2 * (t & 1) + (sp & 1) - 1;
    if (ky < 0 || kx < 0) return nullptr;
    return w + ((int64_t)n * C + c0) * 9 + ky * 3 + kx;
  }
  return w + ((int64_t)n * j.cin_full + j.srcOff[s] + c) * (j.kh * j.kw) + t;
}

__device__ __forceinline__ void pack_store(const PackJob& j, int Ktot, int rows, int n, int k, float v) {
  const int64_t e = (int64_t)n * Ktot + k;
  if (j.mode < 10) { j.wpk[e] = v; return; }
  _Float16 h, l;
  fs_split1(v, fs_scale_of_amax(fs_amax_load(j.amax)), h, l);
  _Float16* out = reinterpret_cast<_Float16*>(j.wpk);
  if (!(j.flags & 1)) {
    _Float16* rec = out + (e >> 5) * 64;
    rec[e & 31] = h;
    rec[32 + (e & 31)] = l;
    return;
  }
  // fragment order (resident-patch kernel): [k-tile][32-row block][hi/lo][k quarter pair s][k half][row][4 dwords];
  // bf16 q of a record's 32-k run sits in dword q / 2: s = q / 16, k half = (q / 8) % 2, dword = (q / 2) % 4
  const int nb = (rows + 31) / 32, kt = k >> 5, q = k & 31;
  const int64_t base = (((int64_t)kt * nb + (n >> 5)) * 2) ;
  const int sidx = (q >> 4) & 1, kh2 = (q >> 3) & 1, dw = (q >> 1) & 3, half = q & 1;
  const int64_t dh = ((((base + 0) * 2 + sidx) * 2 + kh2) * 32 + (n & 31)) * 4 + dw;
  const int64_t dl = ((((base + 1) * 2 + sidx) * 2 + kh2) * 32 + (n & 31)) * 4 + dw;
  out[dh * 2 + half] = h;
  out[dl * 2 + half] = l;
}

__global__ __launch_bounds__(256) void pack_jobs_kernel(PackJobs tab) {
  const PackJob& j = tab.j[blockIdx.y];
  const int taps = j.kh * j.kw;
  int cout = 0;
  for (int p = 0; p < j.npiece; ++p) cout += j.rows[p];
  if (j.mode == 3) {                         // concatenated bias vectors
    for (int e = blockIdx.x * 256 + threadIdx.x; e < cout; e += gridDim.x * 256) {
      int n = e, p = 0;
      while (p + 1 < j.npiece && n >= j.rows[p]) { n -= j.rows[p]; ++p; }
      j.wpk[e] = j.w[p][n];
    }
    return;
  }
  int cin = 0;
  for (int s = 0; s < j.nsrc; ++s) cin += j.srcC[s];
  const int m = j.mode % 10;
  int Ktot, rows;
  if (m == 1) { Ktot = taps * ((cout + 31) / 32 * 32); rows = cin; }
  else { Ktot = 0; for (int s = 0; s < j.nsrc; ++s) Ktot += taps * ((j.srcC[s] + 31) / 32 * 32); rows = cout; }
  const int rows_out = (j.flags & 1) ? (rows + 31) / 32 * 32 : rows;     // fragment order pads the rows with zeros
  const int64_t total = (int64_t)rows_out * Ktot;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int n = (int)(e / Ktot);
    int k = (int)(e % Ktot);
    if (m == 1) {
      const int cpad = (cout + 31) / 32 * 32;
      const int tp = k / cpad, c = k % cpad;
      float v = 0.f;
      if (c < cout && n < rows) {
        int s = 0, cs = n;
        while (s + 1 < j.nsrc && cs >= j.srcC[s]) { cs -= j.srcC[s]; ++s; }
        const float* q = pack_elem(j, c, s, cs, taps - 1 - tp);
        if (q) v = *q;
      }
      pack_store(j, Ktot, rows, n, k, v);
    } else {
      const int k0 = k;
      int s = 0;
      for (; s < j.nsrc; ++s) {
        const int span = taps * ((j.srcC[s] + 31) / 32 * 32);
        if (k < span) break;
        k -= span;
      }
      const int cpad = (j.srcC[s] + 31) / 32 * 32;
      const int t = k / cpad, c = k % cpad;
      const bool ok = c < j.srcC[s] && n < rows;
      float* q = ok ? pack_elem(j, n, s, c, t) : nullptr;
      if (m == 0) pack_store(j, Ktot, rows, n, k0, q ? *q : 0.f);
      else if (q) *q = j.accumulate ? *q + j.scale * j.wpk[e] : j.scale * j.wpk[e];
    }
  }
}

using Cfg128 = GemmCfg<128, 128, 32, 2, 2, 2, 2>;
using Cfg64 = GemmCfg<128, 64, 32, 4, 1, 2, 2>;
using CfgM64 = GemmCfg<64, 128, 32, 1, 4, 2, 2>;     // half-height tile: doubles the workgroup count for narrow N
using Cfg6464 = GemmCfg<64, 64, 32, 2, 2, 2, 2>;     // small tile: 4 workgroups/CU, fine-grained balance over 256 CUs
using WCfg6464 = GemmCfg<64, 64, 32, 2, 2, 0, 0>;
using Cfg6464K16 = GemmCfg<64, 64, 16, 2, 2, 2, 2>;  // 17 KB of LDS: 8 workgroups/CU
using CfgM64K16 = GemmCfg<64, 128, 16, 1, 4, 2, 2>;  // 25 KB: 6 workgroups/CU

using SCfg128 = SplitCfg<128, 128, 2, 2>;
using SCfgN256 = SplitCfg<64, 256, 1, 4, 2, true>;   // 80 KB of LDS: two workgroups per CU; each wave owns 64x64, A rows are read once for N = 256
using SCfg256W16 = SplitCfg<256, 128, 4, 4, 2, true, 1024>;  // sixteen waves, one workgroup per CU
using SCfg128W8 = SplitCfg<128, 128, 2, 4, 2, true, 512>;   // eight waves per workgroup, 66 KB of LDS: two workgroups per CU
using SCfg256N64 = SplitCfg<256, 64, 4, 2, 2, true, 512>;   // N <= 64 layers at large M (encoder layer1, f2): a 128-wide tile would be half empty
using SCfgM64 = SplitCfg<64, 128, 1, 4, 2, true>;    // swizzled 128-byte rows: 48 KB of LDS -> three workgroups per CU
int g_wgrad_split = 2;  // 0: exact fp32; 1/2: split-bf16 weight gradient (double / single LDS image)   (key 4)
int g_wgrad_w8 = 0;       // 512-thread workgroups in the multi-segment weight gradient (key 15); measured slower (7.96 vs 7.32 ms/step): its grid is large already
int g_conv_w8 = 1;        // 512-thread 128x128 tiles for wide layers (key 13); g_conv_w8_min: minimum workgroup count (key 14)
int g_conv_w8_min = 64;      // (measured faster than 64x128 four-wave tiles on every update-block shape, N = 64 .. 576)
int g_conv_uniform = 1;   // uniform-pitch k-tile table when the sources allow it (key 12)
int g_wgrad_blocks_multi = 2048;   // workgroup target of the multi-segment launch (key 11); measured 512: 9.0, 1024: 8.5, 2048: 8.35 ms/step
int g_wgrad_multi = 1;  // one weight-gradient launch per layer per step over all stashed iterations (key 10)
int g_conv_n256 = 0;    // 64x256 tiles for layers whose N fills them (key 9); measured slower than 64x128 (zr 139 vs 119 us, hd 182 vs 125 us)
int g_wgrad_buf = 1;    // buffer-addressed loaders + pixel mask in the split weight-gradient kernel (key 8)
int g_xcd_swizzle = 0;  // experiment switch (key 7)
int g_conv_buf = 1;     // buffer-addressed loaders in the split conv kernels (key 5): 0 never, 1 on 64-row tiles, 2 always
int g_conv_split = 1;   // 0: exact fp32 MFMA; 1: split-bf16 (3-MFMA) core for forward / data-gradient convolutions (key 3)
int g_conv_tile = 0;    // 0 auto, 1 force 128x128, 2 force 64x128, 3 force 64x64   (fsraft_set_tuning key 0)
int g_wgrad_tile = 0;   // 0 auto (128x128), 3 force 64x64                           (key 1)
int g_conv_n64 = 1;            // 256x64 tiles for N <= 64 (key 18) once M reaches g_conv_n64_min_m (key 19)
int g_conv_n64_min_m = 65536;
int g_conv_halo = 1;           // resident-patch 3x3 kernel for few-channel layers at large M (key 20; threshold key 21)
int g_conv_halo_min_m = 65536;
int g_wgrad_xcd = 1;           // XCD-aware workgroup order in the multi-segment weight gradient (key 22; 2: the few-channel kernel too).
                               // Measured: 10.73 -> 9.72 ms/step of weight-gradient time (15 K-tiles re-read each dY tile)
int g_wgrad_patch = 1;         // resident-pixel-block weight gradient for the 3x3 / 1x5 / 5x1 layers (wgrad_patch.inc, key 27)
int g_wgrad_pack = 1;          // few-channel single-source layers on conv_wgrad_pack_kernel (key 16)
int g_wgrad_patch1 = 8192;     // single-segment 3x3 layers with at least this many pixels on the resident-block kernel (key 29; 0: never)
int g_wgrad_blocks_pack = 1024;   // its workgroup target (key 17)
int g_wgrad_blocks = 512;   // target workgroup count of the pixel split (key 2); measured 256: 12.9, 512: 11.5, 1024: 12.9, 2048: 14.1 ms/step
using Cfg32 = GemmCfg<128, 32, 32, 4, 1, 2, 2>;
// weight-gradient tiles: LDS images are filled with float4 rows, so pitches stay multiples of 4
using WCfg128 = GemmCfg<128, 128, 32, 2, 2, 0, 0>;
using WCfg32 = GemmCfg<32, 128, 32, 1, 4, 0, 0>;

// Fills the per-k-tile table of the buffer-addressed kernels; false when the shape does not fit its fields
// (more than KTAB_MAX k-tiles, more than 15 taps, tap/channel offsets beyond 1 MiB, tensors of 2 GiB or more).
bool build_ktab(const ConvArgs& a, ConvArgsT& t) {
  const int taps = a.KH * a.KW, KT = a.Ktot / 32;
  const int64_t M = (int64_t)a.B * a.H * a.W;
  if (KT > KTAB_MAX || taps > 15) return false;
  int kt = 0;
  for (int s = 0; s < a.nsrc; ++s) {
    const int C = a.src[s].C, ld = a.src[s].ld, cpt = (C + 31) / 32;
    if (M * ld * 4 >= (int64_t)FS_OOB) return false;
    for (int i = 0; i < taps * cpt; ++i, ++kt) {
        const int tap = i / cpt, c = i % cpt;
        const int64_t soff = ((int64_t)((tap / a.KW) * a.W + tap % a.KW) * ld + c * 32) * 4;
        if (soff % 16 != 0 || soff / 16 > 0xffff) return false;
        const int crem = C - c * 32 < 32 ? C - c * 32 : 32;
        t.ktab[kt] = (unsigned)(soff / 16) | (unsigned)tap << 16 | (unsigned)s << 20 | (unsigned)crem << 22;
      }
  }
  if (kt != KT) return false;
  t.a = a;
  return true;
}

// Uniform form of the table (see ConvArgsT): possible when every source has the same pitch and all of them, shifted
// taps included, lie inside one 2 GiB window.
bool build_ktab_uniform(const ConvArgs& a, ConvArgsT& t) {
  const int taps = a.KH * a.KW, KT = a.Ktot / 32, PH = a.PH, PW = a.PW;
  const int64_t M = (int64_t)a.B * a.H * a.W;
  if (!g_conv_uniform || 2 * KT > KTAB_MAX || taps > 15) return false;
  const int ld = a.src[0].ld;
  const float* lo = a.src[0].p;
  for (int s = 0; s < a.nsrc; ++s) {
    if (a.src[s].ld != ld) return false;
    if (a.src[s].p < lo) lo = a.src[s].p;
  }
  const float* base = lo - (int64_t)(PH * a.W + PW) * ld;      // tap shifts become non-negative offsets
  int kt = 0;
  for (int s = 0; s < a.nsrc; ++s) {
    const int C = a.src[s].C, cpt = (C + 31) / 32;
    const int64_t delta = (a.src[s].p - lo) * 4;                  // bytes
    for (int tap = 0; tap < taps; ++tap)
      for (int c = 0; c < cpt; ++c, ++kt) {
        const int64_t soff = delta + ((int64_t)((tap / a.KW) * a.W + tap % a.KW) * ld + c * 32) * 4;
        if (soff + M * ld * 4 + 256 >= (int64_t)FS_OOB) return false;
        const int crem = C - c * 32 < 32 ? C - c * 32 : 32;
        t.ktab[2 * kt] = (unsigned)soff;
        t.ktab[2 * kt + 1] = (unsigned)tap | (unsigned)crem << 4;
      }
  }
  if (kt != KT) return false;
  t.a = a; t.ubase = base; t.uld = ld;
  return true;
}


// ---- split-K for small M ----------------------------------------------------------------------------------------------
// At one or two pairs per GPU (the flow-supervisor recipe, config 4) a layer has 35-140 tiles of 64 pixels for 256 CUs and
// its time is the LENGTH of one workgroup's k-loop: 3x3 256 -> 192 takes 55 us at 4416 pixels, 57 us at 7332, 59 us at 8832
// (scripts/conv_micro.py) -- 72 k-tiles one after the other at ~0.75 us.  Here the k-tiles are dealt to `ksplit` workgroups per
// tile (grid.z), which park raw partial tiles in a workspace (conv_igemm_split_kernel, ksplit > 1), and this kernel adds the
// slabs and applies the layer's real epilogue (the row code of conv_epilogue_lds): bias / scale / ReLU / mask / accumulate
// into up to three destinations, or the GRU gate maths.
template <int EPI>
__global__ __launch_bounds__(256) void conv_finish_kernel(const ConvArgs a, const float* __restrict__ ws, int S, int ldw) {
  const int HW = a.H * a.W;
  const int64_t M = (int64_t)a.B * HW;
  const int C4 = (a.N + 3) / 4;
  unsigned mx[3] = {0u, 0u, 0u};
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < M * C4; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / C4;
    const int n = (int)(e % C4) * 4;
    const int nv = a.N - n < 4 ? a.N - n : 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = i < nv ? a.bias[n + i] : 0.f;
    }
    for (int s = 0; s < S; ++s) {
      const f32x4 p = gload4(ws + ((int64_t)s * M + m) * ldw + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += p[i];
    }
    if (EPI == EPI_PLAIN) {
      int di = 0;
      if (a.ndst > 1 && n >= a.dst[1].n0) di = 1;
      if (a.ndst > 2 && n >= a.dst[2].n0) di = 2;
      float* dp = di == 0 ? a.dst[0].p : di == 1 ? a.dst[1].p : a.dst[2].p;
      const int64_t dbs = di == 0 ? a.dst[0].bs : di == 1 ? a.dst[1].bs : a.dst[2].bs;
      const int64_t dps = di == 0 ? a.dst[0].ps : di == 1 ? a.dst[1].ps : a.dst[2].ps;
      const int64_t dcs = di == 0 ? a.dst[0].cs : di == 1 ? a.dst[1].cs : a.dst[2].cs;
      const int dn0 = di == 0 ? a.dst[0].n0 : di == 1 ? a.dst[1].n0 : a.dst[2].n0;
      const bool dacc = (di == 0 ? a.dst[0].accumulate : di == 1 ? a.dst[1].accumulate : a.dst[2].accumulate) != 0;
      const float* mk = di == 0 ? a.rmask[0] : di == 1 ? a.rmask[1] : a.rmask[2];
      const int ldm = di == 0 ? a.ldmask[0] : di == 1 ? a.ldmask[1] : a.ldmask[2];
      const int mkc = di == 0 ? a.maskc[0] : di == 1 ? a.maskc[1] : a.maskc[2];
      const int64_t b = m / HW, pix = m - b * HW;
      float* o = dp + b * dbs + pix * dps + (int64_t)(n - dn0) * dcs;
      for (int i = 0; i < nv; ++i) {
        float r = v[i] * a.alpha;
        if (a.relu) r = fmaxf(r, 0.f);
        if (dacc) r += o[i * dcs];
        if (mk && n - dn0 + i < mkc && mk[m * ldm + (n - dn0) + i] <= 0.f) r = 0.f;
        o[i * dcs] = r;
        mx[di] = fs_umax(mx[di], fs_abs_bits(r));
      }
    } else if (EPI == EPI_ZR) {
      const bool isz = n < a.hid;
      const int c = isz ? n : n - a.hid;
      if (a.pre) v += gload4(a.pre + m * a.ldpre + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = 1.0f / (1.0f + expf(-v[i]));
      if (isz) {
        *reinterpret_cast<f32x4*>(a.dst[0].p + m * a.dst[0].ps + c) = v;                 // z
      } else {
        const f32x4 hh = gload4(a.h + m * a.ldh + c);
        *reinterpret_cast<f32x4*>(a.aux2 + m * a.ld2 + c) = v;                            // r
        f32x4 rh;
#pragma unroll
        for (int i = 0; i < 4; ++i) { rh[i] = v[i] * hh[i]; mx[1] = fs_umax(mx[1], fs_abs_bits(rh[i])); }
        *reinterpret_cast<f32x4*>(a.aux1 + m * a.ld1 + c) = rh;                           // r*h
      }
    } else {   // EPI_Q
      if (a.pre) v += gload4(a.pre + m * a.ldpre + n);
      const f32x4 hh = gload4(a.h + m * a.ldh + n);
      const f32x4 zz = gload4(a.z + m * a.ldz + n);
      f32x4 hn;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = tanhf(v[i]);
        hn[i] = (1.f - zz[i]) * hh[i] + zz[i] * v[i];
        mx[0] = fs_umax(mx[0], fs_abs_bits(hn[i]));
      }
      *reinterpret_cast<f32x4*>(a.aux1 + m * a.ld1 + n) = v;                              // q
      *reinterpret_cast<f32x4*>(a.dst[0].p + m * a.dst[0].ps + n) = hn;                   // h'
    }
  }
  if (a.damax[0] || a.damax[1] || a.damax[2]) {
    __shared__ unsigned red[4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (a.damax[i]) fs_amax_commit(a.damax[i], mx[i], red);
  }
}

int g_conv_ksplit = -1;          // -1 auto (small grids only), 0 / 1 off, >= 2 forced slice count (fsraft_set_tuning key 32)
// Split-K scratch: the ABI allocates nothing, the caller lends it -- per CALL (fsraft_conv_desc.ws, what the Python mirror does) or,
// for bindings written against the round-3 header, per calling THREAD (fsraft_conv_workspace).  Both live in thread_local storage,
// so two host threads enqueueing convolutions for two devices / streams (the reference's nn.DataParallel caller,
// pytorch/train.py:192) never see each other's pointer.
thread_local float* t_reg_ws = nullptr;        // fsraft_conv_workspace of this thread
thread_local int64_t t_reg_ws_floats = 0;
thread_local float* g_conv_ws = nullptr;       // the scratch of the call this thread is inside (set by fsraft_conv_forward)
thread_local int64_t g_conv_ws_floats = 0;

// Slices for a tile grid of `tiles` workgroups, KT k-tiles and N outputs.  Measured (scripts/conv_micro.py, 1 x 47x156, 1 x 46x96,
// 2 x 46x96, 2 x 54x128; round 3, docs/history): the route pays where fewer than ~half of the CUs have a workgroup AND
// the second pass is small -- layers with <= 256 outputs (3x3 256 -> 126: 52 -> 40 us, 1x5 384 -> 128: 42 -> 34, the 3x3
// 512 -> 128 data gradient: 87 -> 52 us; at 4416 pixels 3x3 256 -> 192: 55 -> 44, 1x5 384 -> 256: 46 -> 39); with 512 outputs
// the slabs cost more than the shorter k-loops save (3x3 128 -> 512: 37 -> 55 us), and from ~230 tiles on the CUs are busy
// anyway.  128-row tiles (eight waves): the one-column layers of a two-pair batch leave 69-108 workgroups for 256 CUs and
// gain 15-45 % from two or three slices when the k-loop is long enough to share (3x3 256 -> 126 at 2 x 46x96: 59 -> 38 us,
// 3x3 512 -> 128 data gradient 100 -> 56, 1x5 384 -> 128: 45 -> 32); two-column layers only with >= 48 k-tiles (1x5 384 -> 256
// at 1 x 47x156: 47 -> 41 us, but the 20-tile data gradient 1x5 128 -> 256: 23 -> 29); above ~120 tiles nothing gains.
int pick_ksplit(int64_t tiles, int KT, int N, int64_t M, int ldw, int bm) {
  int S = 1;
  if (g_conv_ksplit >= 2) S = g_conv_ksplit;
  else if (g_conv_ksplit == -1 && bm == 64 && tiles <= 150 && N <= 256 && KT >= 16) S = (int)((400 + tiles - 1) / tiles);
  else if (g_conv_ksplit == -1 && bm == 128 && tiles <= 120 && N <= 256 && KT >= (N <= 128 ? 36 : 48)) S = tiles <= 115 ? (int)(230 / tiles) : 2;
  if (S > KT / 6) S = KT / 6;
  if (S > 6) S = 6;
  while (S > 1 && (int64_t)S * M * ldw > g_conv_ws_floats) --S;
  if (S > 1) {                                   // no empty slice
    const int per = (KT + S - 1) / S;
    S = (KT + per - 1) / per;
  }
  return S < 2 ? 1 : S;
}

template <class Cfg>
int launch_conv_split(const ConvArgs& a, int epi, hipStream_t s) {
  const int M = a.B * a.H * a.W;
  dim3 grid(ceil_div(a.N, Cfg::BN), ceil_div(M, Cfg::BM));
  ConvArgsT t;
  // XCD-aware tile order (tile_of_block) once several N tiles re-read each A tile and the grid is many rounds deep:
  // 192 -> 256 at M = 225 K (encoder-sized data gradients): 776 -> 714 us; nothing at the update block's 220..880 tiles.
  const int swz = g_xcd_swizzle || (grid.x >= 2 && (int64_t)grid.x * grid.y >= 2048);
  // buffer-addressed loaders + branch-free k-loop: measured faster on the 64-row tiles, slower on 128x128
  const bool buf = g_conv_buf == 2 || (g_conv_buf == 1 && Cfg::BM == 64) || Cfg::BN == 256 || Cfg::NT != 256;
  if constexpr (Cfg::LDS_ALLOC >= Cfg::BM * (Cfg::BN + 4) * 4) {
    // split-K route (small M): partial tiles into the workspace, then conv_finish_kernel with the layer's own epilogue
    const int ldw = (a.N + 3) / 4 * 4;
    const bool gru_ok = epi == EPI_PLAIN || (a.N % 4 == 0 && (!a.pre || a.ldpre % 4 == 0));
    const int S = (buf && g_conv_ws && gru_ok && (Cfg::BM == 64 || (Cfg::BM == 128 && Cfg::NT == 512) || g_conv_ksplit >= 2)) ? pick_ksplit((int64_t)grid.x * grid.y, a.Ktot / 32, a.N, M, ldw, Cfg::BM) : 1;
    if (S > 1) {
      const bool uni_tab = build_ktab_uniform(a, t);
      if (uni_tab || build_ktab(a, t)) {
        ConvArgs& p = t.a;
        p.swz = 0;
        p.ksplit = S;
        p.dst[0] = Dst{g_conv_ws, (int64_t)a.H * a.W * ldw, ldw, 1, 0, 0};
        p.dst[1] = p.dst[2] = p.dst[0];
        p.ndst = 1; p.bias = nullptr; p.relu = 0; p.alpha = 1.0f;
        for (int i = 0; i < 3; ++i) { p.rmask[i] = nullptr; p.damax[i] = nullptr; }      // (the finish pass raises the amax words)
        dim3 g3(grid.x, grid.y, S);
        if (uni_tab) hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_PLAIN, 2>), g3, dim3(Cfg::NT), 0, s, t);
        else hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_PLAIN, 1>), g3, dim3(Cfg::NT), 0, s, t);
        int rc = fs_launch_status();
        if (rc) return rc;
        const int64_t work = (int64_t)M * (ldw / 4);
        const int fb = (int)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096);
        if (epi == EPI_PLAIN) hipLaunchKernelGGL(conv_finish_kernel<EPI_PLAIN>, dim3(fb), dim3(256), 0, s, a, g_conv_ws, S, ldw);
        else if (epi == EPI_ZR) hipLaunchKernelGGL(conv_finish_kernel<EPI_ZR>, dim3(fb), dim3(256), 0, s, a, g_conv_ws, S, ldw);
        else hipLaunchKernelGGL(conv_finish_kernel<EPI_Q>, dim3(fb), dim3(256), 0, s, a, g_conv_ws, S, ldw);
        return fs_launch_status();
      }
    }
  }
  if (buf && build_ktab_uniform(a, t)) {
    t.a.swz = swz;
    if (epi == EPI_PLAIN) hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_PLAIN, 2>), grid, dim3(Cfg::NT), 0, s, t);
    else if (epi == EPI_ZR) hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_ZR, 2>), grid, dim3(Cfg::NT), 0, s, t);
    else hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_Q, 2>), grid, dim3(Cfg::NT), 0, s, t);
    return fs_launch_status();
  }
  if (buf && build_ktab(a, t)) {
    t.a.swz = swz;
    if (epi == EPI_PLAIN) hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_PLAIN, 1>), grid, dim3(Cfg::NT), 0, s, t);
    else if (epi == EPI_ZR) hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_ZR, 1>), grid, dim3(Cfg::NT), 0, s, t);
    else hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_Q, 1>), grid, dim3(Cfg::NT), 0, s, t);
    return fs_launch_status();
  }
  if constexpr (Cfg::NT != 256) {
    return -1;          // the wide configurations exist for the buffer-addressed paths only: caller falls back
  } else {
  if (epi == EPI_PLAIN) hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_PLAIN>), grid, dim3(256), 0, s, a);
  else if (epi == EPI_ZR) hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_ZR>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_Q>), grid, dim3(256), 0, s, a);
  return fs_launch_status();
  }
}

// plain-epilogue-only launcher of a wide configuration (the narrow-N tiles never carry a GRU epilogue)
template <class Cfg>
int launch_conv_split_plain(const ConvArgs& a, hipStream_t s) {
  const int M = a.B * a.H * a.W;
  dim3 grid(ceil_div(a.N, Cfg::BN), ceil_div(M, Cfg::BM));
  ConvArgsT t;
  if (build_ktab_uniform(a, t)) {
    hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_PLAIN, 2>), grid, dim3(Cfg::NT), 0, s, t);
    return fs_launch_status();
  }
  if (build_ktab(a, t)) {
    hipLaunchKernelGGL((conv_igemm_split_kernel<Cfg, EPI_PLAIN, 1>), grid, dim3(Cfg::NT), 0, s, t);
    return fs_launch_status();
  }
  return -1;
}

template <class Cfg>
int launch_conv(const ConvArgs& a, int epi, hipStream_t s) {
  const int M = a.B * a.H * a.W;
  dim3 grid(ceil_div(a.N, Cfg::BN), ceil_div(M, Cfg::BM));
  if (epi == EPI_PLAIN) hipLaunchKernelGGL((conv_igemm_kernel<Cfg, EPI_PLAIN>), grid, dim3(256), 0, s, a);
  else if (epi == EPI_ZR) hipLaunchKernelGGL((conv_igemm_kernel<Cfg, EPI_ZR>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_igemm_kernel<Cfg, EPI_Q>), grid, dim3(256), 0, s, a);
  return fs_launch_status();
}

int g_conv_patch_min_m = 8192;   // ... from this many pixels on (key 31)
int g_conv_patch = 1;      // resident-patch, channel-streaming kernel for the 3x3 / 1x5 / 5x1 layers (conv_patch.inc, key 26; 2: 128-pixel tiles too)
int g_conv_patch64 = 1;    // ... also for the 3x3 layers with 33..64 outputs (64-column tiles; key 28; 2: 128-pixel tiles)
#include "conv_patch.inc"

int conv_ktot(const int* C, int nsrc, int taps) {
  int k = 0;
  for (int s = 0; s < nsrc; ++s) k += taps * (((C[s] + 31) / 32) * 32);
  return k;
}

}  // namespace

// Flat C descriptor so the ABI stays plain-old-data (mirrored by ctypes in _lib.py).
struct fsraft_conv_desc {
  const float* src[3]; int srcC[3]; int srcld[3]; int nsrc;
  const float* wpk; const float* bias;
  const float* wpk_split;        // same matrix packed with mode 10/11 (or NULL): enables the split-bf16 core
  int B, H, W, KH, KW, N;
  float* dst[3]; int64_t dst_bs[3]; int64_t dst_ps[3]; int64_t dst_cs[3]; int dst_n0[3]; int dst_acc[3]; int ndst;
  int relu; float alpha;
  int epi;                       // 0 plain, 2 GRU z/r, 3 GRU q
  const float* h; int ldh;
  const float* z; int ldz;
  float* aux1; int ld1;
  float* aux2; int ld2;
  int hid;
  const float* pre; int ldpre;   // GRU epilogues: addend to the pre-activation (e.g. the context part of the conv), or NULL
  const float* rmask[3]; int ldmask[3]; int maskc[3];   // epi 0, per destination: zero column j < maskc where rmask[m*ldmask+j] <= 0
  const float* wpk_frag;         // wpk_split in fragment order (or NULL): enables the resident-patch 3x3 kernel
  int pad_h1, pad_w1;            // 0: taps centred (KH / 2, KW / 2); else 1 + the top / left padding (even kernel sizes)
  float* ws; int64_t ws_floats;  // split-K scratch of THIS call (NULL: the calling thread's fsraft_conv_workspace registration)
  // split arithmetic (fsraft.h "amax words"): per source the word of its tensor, the word wpk_split / wpk_frag were packed
  // with, and per destination an optional word the epilogue raises (GRU epilogues: [0] new state, [1] r*h)
  const unsigned* src_amax[3]; const unsigned* w_amax; unsigned* dst_amax[3];
};

extern "C" int fsraft_conv_ktot(const int* srcC, int nsrc, int KH, int KW) {
  if (!srcC || nsrc < 1 || nsrc > 3) return -1;
  return conv_ktot(srcC, nsrc, KH * KW);
}

namespace {
// fsraft_conv_forward_stats parks its request here for the call it wraps (same thread); the launch sites whose kernels can
// carry the statistics pick it up and say so
struct StatReq { float* sum; float* sq; int slots; bool done; };
thread_local StatReq t_stat{nullptr, nullptr, 0, false};
}  // namespace

extern "C" int fsraft_conv_forward(const fsraft_conv_desc* d, hipStream_t stream) {
  if (!d || d->nsrc < 1 || d->nsrc > 3 || d->ndst < 1 || d->ndst > 3 || !d->wpk || d->N < 1) return FS_ERR_ARG;
  // (statistics of the raw result only: one destination, no bias / ReLU / scale / mask / accumulation in the epilogue)
  const bool want_stats = t_stat.sum != nullptr && d->epi == EPI_PLAIN && d->ndst == 1 && !d->relu && d->alpha == 1.0f && !d->dst_acc[0] &&
                          !d->rmask[0] && !d->bias && d->dst_n0[0] == 0 && d->N % 4 == 0;
  if (d->ws && (((uintptr_t)d->ws & 15) || d->ws_floats < 0)) return FS_ERR_ARG;
  g_conv_ws = d->ws ? d->ws : t_reg_ws;                     // scratch of THIS call (thread_local: see the declaration)
  g_conv_ws_floats = d->ws ? d->ws_floats : t_reg_ws_floats;
  ConvArgs a{};
  for (int s = 0; s < 3; ++s) {
    a.src[s] = Src{s < d->nsrc ? d->src[s] : d->src[0], s < d->nsrc ? d->srcC[s] : 0, s < d->nsrc ? d->srcld[s] : 4};
    if (s < d->nsrc && (!d->src[s] || d->srcld[s] % 4 != 0 || d->srcC[s] < 1)) return FS_ERR_ARG;
  }
  a.nsrc = d->nsrc;
  a.wpk = d->wpk; a.Ktot = conv_ktot(d->srcC, d->nsrc, d->KH * d->KW); a.bias = d->bias;
  a.B = d->B; a.H = d->H; a.W = d->W; a.KH = d->KH; a.KW = d->KW; a.N = d->N;
  a.PH = d->pad_h1 ? d->pad_h1 - 1 : d->KH / 2; a.PW = d->pad_w1 ? d->pad_w1 - 1 : d->KW / 2;
  if (a.PH < 0 || a.PH >= d->KH || a.PW < 0 || a.PW >= d->KW) return FS_ERR_ARG;
  for (int i = 0; i < 3; ++i) {
    const int j = i < d->ndst ? i : 0;
    a.dst[i] = Dst{d->dst[j], d->dst_bs[j], d->dst_ps[j], d->dst_cs[j], d->dst_n0[j], d->dst_acc[j]};
    if (i < d->ndst && !d->dst[i]) return FS_ERR_ARG;
  }
  a.ndst = d->ndst; a.relu = d->relu; a.alpha = d->alpha;
  a.h = d->h; a.ldh = d->ldh; a.z = d->z; a.ldz = d->ldz; a.aux1 = d->aux1; a.ld1 = d->ld1; a.aux2 = d->aux2; a.ld2 = d->ld2;
  a.hid = d->hid;
  a.pre = d->epi == EPI_PLAIN ? nullptr : d->pre; a.ldpre = d->ldpre;
  if (a.pre && (d->ldpre % 4 != 0 || ((uintptr_t)d->pre & 15))) return FS_ERR_ARG;
  for (int i = 0; i < 3; ++i) {
    const bool on = d->epi == EPI_PLAIN && i < d->ndst && d->rmask[i] != nullptr;
    a.rmask[i] = on ? d->rmask[i] : nullptr; a.ldmask[i] = d->ldmask[i]; a.maskc[i] = d->maskc[i];
    if (on && (d->ldmask[i] % 4 != 0 || ((uintptr_t)d->rmask[i] & 15))) return FS_ERR_ARG;
  }
  a.swz = g_xcd_swizzle;
  for (int i = 0; i < 3; ++i) {
    a.samax[i] = i < d->nsrc ? d->src_amax[i] : nullptr;
    a.damax[i] = (i < d->ndst || d->epi != EPI_PLAIN) ? d->dst_amax[i] : nullptr;
    if ((uintptr_t)a.samax[i] & 3 || (uintptr_t)a.damax[i] & 3) return FS_ERR_ARG;
  }
  a.wamax = d->w_amax;
  if (d->epi == EPI_ZR && (!d->h || !d->aux1 || !d->aux2 || d->hid * 2 != d->N)) return FS_ERR_ARG;
  if (d->epi == EPI_Q && (!d->h || !d->z || !d->aux1)) return FS_ERR_ARG;
  if (d->epi != EPI_PLAIN && d->epi != EPI_ZR && d->epi != EPI_Q) return FS_ERR_ARG;
  if (g_conv_patch && g_conv_patch64 && g_conv_split == 1 && d->wpk_split && d->epi == EPI_PLAIN && d->KH == 3 && d->KW == 3 &&
      d->N > 32 && d->N <= 64 && (int64_t)d->B * d->H * d->W >= g_conv_patch_min_m) {
    ConvArgs p = a;
    p.wpk = d->wpk_split;
    if (want_stats) { p.st_sum = t_stat.sum; p.st_sq = t_stat.sq; p.st_slots = t_stat.slots; }
    const int rc = launch_conv_patch(p, d->epi, stream);
    if (rc >= 0) { t_stat.done = want_stats; return rc; }
  }
  if (g_conv_halo && g_conv_split == 1 && d->wpk_frag && d->epi == EPI_PLAIN && d->nsrc == 1 && d->KH == 3 && d->KW == 3 &&
      a.PH == 1 && a.PW == 1 &&
      d->srcC[0] % 4 == 0 && d->srcC[0] > 32 && d->srcC[0] <= 64 && d->N <= 128 && d->N > 32 && d->ndst == 1 &&
      d->dst_cs[0] == 1 && d->dst_n0[0] == 0 && !(d->dst_acc[0] && d->relu) && !a.rmask[0] && d->alpha == 1.0f &&
      (int64_t)d->B * d->H * d->W >= g_conv_halo_min_m && (int64_t)d->H * d->W * d->srcld[0] * 4 < 0x7fffffff) {
    HaloArgs h{d->src[0], d->srcld[0], d->srcC[0], reinterpret_cast<const char*>(d->wpk_frag), d->bias,
               d->dst[0], d->dst_bs[0], d->dst_ps[0], d->N, d->B, d->H, d->W, d->relu, d->dst_acc[0], nullptr, nullptr, 0,
               d->src_amax[0], d->w_amax, d->dst_amax[0]};
    bool halo_stats = false;
    // Measured (scripts/conv_micro.py, halo on / off): 64 -> 64 at 8x220x512 238 vs 442 us.  With three or four channel
    // groups the patch takes 78 / 104 KB of LDS, one or two 4-wave workgroups per CU, and the kernel loses to the implicit
    // GEMM (96 -> 96 at 8x110x256: 249 vs 165 us; 128 -> 128 at 8x55x128: 93 vs 65 us), so only two-group layers come here.
    if (want_stats) { h.st_sum = t_stat.sum; h.st_sq = t_stat.sq; h.st_slots = t_stat.slots; halo_stats = true; }
    t_stat.done = halo_stats;
    return d->N > 64 ? launch_halo<2, 2>(h, stream) : launch_halo<2, 1>(h, stream);
  }
  if (d->N <= 32 && d->epi == EPI_PLAIN) return launch_conv<Cfg32>(a, d->epi, stream);
  // 33..64 outputs: half of a 64x128 split tile is padding, still ~2x faster than the exact 64-wide kernel
  if (d->N <= 64 && d->epi == EPI_PLAIN && !(g_conv_split && d->wpk_split && g_conv_buf)) return launch_conv<Cfg64>(a, d->epi, stream);
  // one workgroup per CU is not enough to keep the matrix pipe busy: when the 128x128 grid has
  // fewer than ~2 workgroups per CU, halve the tile height
  const int M = d->B * d->H * d->W;
  if (g_conv_split && d->wpk_split && d->N > 32) {
    a.wpk = d->wpk_split;
    // 64x128 tiles with the buffer-addressed loaders and the branch-free k-loop are the fastest variant on every
    // update-block shape at M = 28160 (scripts/conv_micro.py: zr 129 us, hd 142 us, q 65 us, m2 61 us, c1 41 us;
    // the 128x128 kernel needs 131 / 157 / 87 / 78 / 45 us); 128x128 stays selectable (key 3 = 4) and is the
    // fallback for shapes the k-tile table cannot describe.
    const bool narrow = g_conv_buf != 0 || (int64_t)ceil_div(d->N, 128) * ceil_div(M, 128) < 400;
    if (g_conv_patch && g_conv_split == 1 && d->KH * d->KW > 1 && d->N > 64 && M >= g_conv_patch_min_m) {
      ConvArgs p = a;
      if (want_stats) { p.st_sum = t_stat.sum; p.st_sq = t_stat.sq; p.st_slots = t_stat.slots; }
      const int rc = launch_conv_patch(p, d->epi, stream);
      if (rc >= 0) { t_stat.done = want_stats; return rc; }
    }
    if (g_conv_split == 5 || (g_conv_split == 1 && g_conv_n256 && d->N >= 256 &&
                              ceil_div(d->N, 256) * 256 <= ceil_div(d->N, 128) * 128))
      return launch_conv_split<SCfgN256>(a, d->epi, stream);
    if (g_conv_split == 1 && g_conv_n64 && d->N <= 64 && d->epi == EPI_PLAIN && M >= g_conv_n64_min_m) {
      const int rc = launch_conv_split_plain<SCfg256N64>(a, stream);
      if (rc >= 0) return rc;
    }
    // eight-wave 128x128 tiles where they fill the machine in one round (N >= 256 at M ~ 28 K): key 13
    // sixteen-wave 256x128 tiles: a further 3-6 % on the 192..512-output layers (zr 93 -> 87 us), slower on m2 (N = 576)
    if (g_conv_split == 1 && ((g_conv_w8 == 1 && d->N >= 192 && d->N <= 512 && M >= 16384) || g_conv_w8 == 2) && d->N >= 192) {
      const int rc = launch_conv_split<SCfg256W16>(a, d->epi, stream);
      if (rc >= 0) return rc;
    }
    if (g_conv_w8 && g_conv_split == 1 && (int64_t)ceil_div(d->N, 128) * ceil_div(M, 128) >= g_conv_w8_min) {
      const int rc = launch_conv_split<SCfg128W8>(a, d->epi, stream);
      if (rc >= 0) return rc;
    }
    if (g_conv_split == 3) return launch_conv_split<SCfgM64>(a, d->epi, stream);
    if (g_conv_split == 4) return launch_conv_split<SCfg128>(a, d->epi, stream);
    return narrow ? launch_conv_split<SCfgM64>(a, d->epi, stream) : launch_conv_split<SCfg128>(a, d->epi, stream);
  }
  if (g_conv_tile == 3) return launch_conv<Cfg6464>(a, d->epi, stream);
  if (g_conv_tile == 4) return launch_conv<Cfg6464K16>(a, d->epi, stream);
  if (g_conv_tile == 5) return launch_conv<CfgM64K16>(a, d->epi, stream);
  if (g_conv_tile == 2) return launch_conv<CfgM64>(a, d->epi, stream);
  if (g_conv_tile == 1) return launch_conv<Cfg128>(a, d->epi, stream);
  if ((int64_t)ceil_div(d->N, 128) * ceil_div(M, 128) < 512) return launch_conv<CfgM64>(a, d->epi, stream);
  return launch_conv<Cfg128>(a, d->epi, stream);
}

// fsraft_conv_forward whose kernel, where it can, also accumulates the per-image column sums of its result and of its squares
// (the statistics the InstanceNorm behind the convolution needs: fsraft_inorm_relu_cl_fwd with have_sums = 1 then skips its
// own pass over the tensor).  sum, sq: [B * slots][N] fp32, ZERO on entry (the rows of one image are added up by the consumer);
// *done = 1 if the launch carried them, 0 if the caller has to compute them (kernels whose tiles straddle images, epilogues
// with a bias / ReLU / mask).
extern "C" int fsraft_conv_forward_stats(const fsraft_conv_desc* d, float* sum, float* sq, int slots, int* done, hipStream_t stream) {
  if (!sum || !sq || slots < 1 || !done) return FS_ERR_ARG;
  t_stat = StatReq{sum, sq, slots, false};
  const int rc = fsraft_conv_forward(d, stream);
  *done = (rc == FS_OK && t_stat.done) ? 1 : 0;
  t_stat = StatReq{nullptr, nullptr, 0, false};
  return rc;
}


// The split-K route of the small-M convolutions needs a scratch buffer; the ABI allocates nothing, so the binding hands one
// over (and keeps it alive): per call in fsraft_conv_desc.ws, or -- this entry point -- for every later call of the CALLING THREAD
// whose descriptor carries none.  Calls that enqueue convolutions on DIFFERENT streams concurrently must not share a buffer.
extern "C" int fsraft_conv_workspace(float* ws, int64_t floats) {
  if (ws && (((uintptr_t)ws & 15) || floats < 0)) return FS_ERR_ARG;
  t_reg_ws = ws;
  t_reg_ws_floats = ws ? floats : 0;
  return FS_OK;
}

// Reads back the arithmetic-mode switches (key 3: forward / data-gradient convolutions, key 4: weight gradients); the
// host side uses it to skip packing the exact-fp32 weight matrices while the split-bf16 kernels are the ones that run.
extern "C" int fsraft_get_tuning(int key) {
  if (key == 3) return g_conv_split;
  if (key == 4) return g_wgrad_split;
  return -1;
}

// fsraft.h: one switch for the arithmetic of every GEMM-shaped kernel of the library
extern "C" int fsraft_set_build_split(int on);
extern "C" int fsraft_set_gemm_split(int on);
extern "C" int fsraft_set_arithmetic(int mode) {
  if (mode != 0 && mode != 1) return FS_ERR_ARG;
  g_conv_split = mode ? 1 : 0;
  g_wgrad_split = mode ? 2 : 0;
  fsraft_set_build_split(mode);
  fsraft_set_gemm_split(mode);
  return FS_OK;
}
extern "C" int fsraft_get_arithmetic(void) { return g_conv_split != 0 ? 1 : 0; }

extern "C" int fsraft_set_tuning(int key, int value) {
  if (key == 0) g_conv_tile = value;
  else if (key == 1) g_wgrad_tile = value;
  else if (key == 2) g_wgrad_blocks = value;
  else if (key == 3) g_conv_split = value;
  else if (key == 5) g_conv_buf = value;
  else if (key == 8) g_wgrad_buf = value;
  else if (key == 9) g_conv_n256 = value;
  else if (key == 10) g_wgrad_multi = value;
  else if (key == 11) g_wgrad_blocks_multi = value;
  else if (key == 18) g_conv_n64 = value;
  else if (key == 19) g_conv_n64_min_m = value;
  else if (key == 20) g_conv_halo = value;
  else if (key == 21) g_conv_halo_min_m = value;
  else if (key == 22) g_wgrad_xcd = value;
  else if (key == 26) g_conv_patch = value;
  else if (key == 31) g_conv_patch_min_m = value;
  else if (key == 32) g_conv_ksplit = value;
  else if (key == 27) g_wgrad_patch = value;
  else if (key == 28) g_conv_patch64 = value;
  else if (key == 29) g_wgrad_patch1 = value;
  else if (key == 16) g_wgrad_pack = value;
  else if (key == 17) g_wgrad_blocks_pack = value;
  else if (key == 12) g_conv_uniform = value;
  else if (key == 13) g_conv_w8 = value;
  else if (key == 14) g_conv_w8_min = value;
  else if (key == 15) g_wgrad_w8 = value;
  else if (key == 7) g_xcd_swizzle = value;
  else if (key == 4) g_wgrad_split = value;
  else return FS_ERR_ARG;
  return FS_OK;
}

// dwpk[Cout][Ktot] += dY^T * im2col(X)   (same packed layout as the forward weights)
extern "C" int fsraft_col_sum(const float* x, int ld, int64_t M, int C, float* out, float scale, hipStream_t s);

namespace { int launch_wgrad_patch(WgradArgsM m, hipStream_t s); }

extern "C" int fsraft_conv_wgrad(const float* dy, int ldy, int Cout, const float* const* src, const int* srcC,
                                 const int* srcld, int nsrc, float* dwpk, float* dbias, int B, int H, int W, int KH,
                                 int KW, const unsigned* dy_amax, const unsigned* const* src_amax, hipStream_t stream) {
  if (!dy || !src || !dwpk || nsrc < 1 || nsrc > 3 || ldy % 4 != 0) return FS_ERR_ARG;
  WgradArgs a{};
  a.dy = dy; a.ldy = ldy; a.Cout = Cout;
  a.dyamax = dy_amax;
  for (int s = 0; s < 3; ++s) a.samax[s] = (src_amax && s < nsrc) ? src_amax[s] : nullptr;
  const bool small_m = Cout <= 32;
  const bool t64 = !small_m && g_wgrad_tile == 3;
  const int bn = t64 ? 64 : 128, bm = small_m ? 32 : (t64 ? 64 : 128);
  int xt128 = 0;
  for (int s = 0; s < 3; ++s) {
    a.src[s] = Src{s < nsrc ? src[s] : src[0], s < nsrc ? srcC[s] : 0, s < nsrc ? srcld[s] : 4};
    if (s < nsrc) {
      if (!src[s] || srcld[s] % 4 != 0) return FS_ERR_ARG;
      xt128 += KH * KW * ceil_div(srcC[s], bn);
    }
  }
  a.nsrc = nsrc; a.dwpk = dwpk; a.Ktot = conv_ktot(srcC, nsrc, KH * KW);
  a.B = B; a.H = H; a.W = W; a.KH = KH; a.KW = KW;
  const int64_t M = (int64_t)B * H * W;
  // one segment of a 3x3 layer at encoder size: the resident-block kernel (wgrad_patch.inc) reads dY and X once instead of once
  // per tap group
  if (g_wgrad_patch && g_wgrad_patch1 && g_wgrad_split != 0 && KH == 3 && KW == 3 && Cout > 32 && M >= g_wgrad_patch1) {
    WgradArgsM m{};
    m.a = a; m.a.dbias = dbias;
    m.nseg = 1;
    m.dys[0] = dy; m.dyam[0] = dy_amax;
    for (int s = 0; s < nsrc; ++s) { m.srcs[s][0] = src[s]; m.sam[s][0] = a.samax[s]; }
    const int rc = launch_wgrad_patch(m, stream);
    if (rc >= 0) return rc;
  }
  if (g_wgrad_pack && g_wgrad_split != 0 && g_wgrad_buf && nsrc == 1 && srcC[0] <= 96 && KH * KW > 1 &&
      (int64_t)(32 * (WGRAD_PACK_WORDS - 2) + 2 * W + 2) * srcld[0] * 4 < 0x7fffffff) {
    // few input channels: several taps per x tile, 64-row dY tiles (conv_wgrad_pack_kernel)
    const int cpad = ceil_div(srcC[0], 32) * 32;
    int tp = SWCfgPack::BN / cpad;
    if (tp > WGRAD_PACK_SLOTS) tp = WGRAD_PACK_SLOTS;
    const int xt = ceil_div(KH * KW, tp), yt = ceil_div(Cout, SWCfgPack::BM);
    int64_t want = (g_wgrad_blocks_pack + (int64_t)xt * yt - 1) / ((int64_t)xt * yt);
    if (want < 1) want = 1;
    int64_t chunk = (M + want - 1) / want;
    if (chunk < 256) chunk = 256;
    if (chunk > 32 * (WGRAD_PACK_WORDS - 2)) chunk = 32 * (WGRAD_PACK_WORDS - 2);
    chunk = (chunk + 31) / 32 * 32;
    a.kchunk = (int)chunk;
    a.dbias = dbias;
    dim3 grid(xt, yt, (unsigned)((M + chunk - 1) / chunk));
    if (g_wgrad_xcd == 2) {      // measured slower here (64 -> 64 at 8x220x512: 406 vs 377 us): three tap groups per dY tile only
      a.xcd_xt = xt; a.xcd_yt = yt; a.xcd_groups = yt * (int)grid.z;
      grid = dim3((unsigned)(ceil_div(a.xcd_groups, 8) * 8 * xt), 1, 1);
    }
    hipLaunchKernelGGL((conv_wgrad_pack_kernel<SWCfgPack>), grid, dim3(SWCfgPack::NT), 0, stream, a);
    return fs_launch_status();
  }
  const int ytiles = ceil_div(Cout, bm);
  // aim for ~4 workgroups per CU; each split handles a multiple of 32 pixels, at least 256
  int64_t want = (g_wgrad_blocks + (int64_t)xt128 * ytiles - 1) / ((int64_t)xt128 * ytiles);
  if (want < 1) want = 1;
  int64_t chunk = (M + want - 1) / want;
  if (chunk < 256) chunk = 256;
  if (chunk > 32 * (WGRAD_MASK_WORDS - 2)) chunk = 32 * (WGRAD_MASK_WORDS - 2);   // pixel-mask capacity of the buffer-addressed kernel
  chunk = (chunk + 31) / 32 * 32;
  a.kchunk = (int)chunk;
  const int zs = (int)((M + chunk - 1) / chunk);
  dim3 grid(xt128, ytiles, zs);
  a.dbias = dbias;
  const bool wbuf = g_wgrad_buf && chunk <= 32 * (WGRAD_MASK_WORDS - 2) && (int64_t)chunk * 4 * 2048 < 0x7fffffff;
  if (!small_m && !t64 && g_wgrad_split == 1) {
    if (wbuf) hipLaunchKernelGGL((conv_wgrad_split_kernel<SWCfg128, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_wgrad_split_kernel<SWCfg128>), grid, dim3(256), 0, stream, a);
    return fs_launch_status();
  }
  if (!small_m && !t64 && g_wgrad_split == 2) {
    if (wbuf) hipLaunchKernelGGL((conv_wgrad_split_kernel<SWCfg128S, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_wgrad_split_kernel<SWCfg128S>), grid, dim3(256), 0, stream, a);
    return fs_launch_status();
  }
  // the exact-fp32 kernels do not fuse the bias gradient: separate column-sum pass
  if (dbias) { const int rc = fsraft_col_sum(dy, ldy, M, Cout, dbias, 1.0f, stream); if (rc) return rc; }
  if (small_m) hipLaunchKernelGGL((conv_wgrad_kernel<WCfg32>), grid, dim3(256), 0, stream, a);
  else if (t64) hipLaunchKernelGGL((conv_wgrad_kernel<WCfg6464>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<WCfg128>), grid, dim3(256), 0, stream, a);
  return fs_launch_status();
}

// nseg (dY, X) pairs of identical shape in one launch; src[seg * nsrc + s].  Falls back to one launch per segment when
// the buffer-addressed split kernel cannot take the shape.
#include "wgrad_patch.inc"

extern "C" int fsraft_conv_wgrad_multi(const float* const* dy, int nseg, int ldy, int Cout, const float* const* src,
                                       const int* srcC, const int* srcld, int nsrc, float* dwpk, float* dbias, int B,
                                       int H, int W, int KH, int KW, const unsigned* const* dy_amax,
                                       const unsigned* const* src_amax, hipStream_t stream) {
  if (!dy || !src || !dwpk || nseg < 1 || nsrc < 1 || nsrc > 3 || ldy % 4 != 0) return FS_ERR_ARG;
  const int64_t M = (int64_t)B * H * W;
  const bool fast = g_wgrad_multi && nseg > 1 && Cout > 32 && g_wgrad_tile != 3 && g_wgrad_split == 2 && g_wgrad_buf;
  for (int base = 0; base < nseg; base += WGRAD_MAX_SEG) {
    const int n = nseg - base < WGRAD_MAX_SEG ? nseg - base : WGRAD_MAX_SEG;
    if (!fast || n == 1) {
      for (int i = 0; i < n; ++i) {
        const int rc = fsraft_conv_wgrad(dy[base + i], ldy, Cout, src + (size_t)(base + i) * nsrc, srcC, srcld, nsrc, dwpk,
                                         dbias, B, H, W, KH, KW, dy_amax ? dy_amax[base + i] : nullptr,
                                         src_amax ? src_amax + (size_t)(base + i) * nsrc : nullptr, stream);
        if (rc) return rc;
      }
      continue;
    }
    WgradArgsM m{};
    WgradArgs& a = m.a;
    a.dy = dy[base]; a.ldy = ldy; a.Cout = Cout;
    int xt128 = 0;
    for (int s = 0; s < 3; ++s) {
      a.src[s] = Src{s < nsrc ? src[(size_t)base * nsrc + s] : src[(size_t)base * nsrc], s < nsrc ? srcC[s] : 0, s < nsrc ? srcld[s] : 4};
      if (s < nsrc) {
        if (srcld[s] % 4 != 0) return FS_ERR_ARG;
        xt128 += KH * KW * ceil_div(srcC[s], 128);
      }
    }
    for (int i = 0; i < n; ++i) {
      if (!dy[base + i]) return FS_ERR_ARG;
      m.dys[i] = dy[base + i];
      m.dyam[i] = dy_amax ? dy_amax[base + i] : nullptr;
      for (int s = 0; s < nsrc; ++s) {
        if (!src[(size_t)(base + i) * nsrc + s]) return FS_ERR_ARG;
        m.srcs[s][i] = src[(size_t)(base + i) * nsrc + s];
        m.sam[s][i] = src_amax ? src_amax[(size_t)(base + i) * nsrc + s] : nullptr;
      }
    }
    a.nsrc = nsrc; a.dwpk = dwpk; a.Ktot = conv_ktot(srcC, nsrc, KH * KW);
    a.B = B; a.H = H; a.W = W; a.KH = KH; a.KW = KW; a.dbias = dbias;
    if (g_wgrad_patch && KH * KW > 1) {
      m.nseg = n;
      const int rc = launch_wgrad_patch(m, stream);
      if (rc == 0) continue;
      if (rc > 0) return rc;
    }
    const int ytiles = ceil_div(Cout, 128);
    // ~g_wgrad_blocks workgroups in total, a whole number of pixel splits per segment
    int64_t zs = (g_wgrad_blocks_multi + (int64_t)xt128 * ytiles * n - 1) / ((int64_t)xt128 * ytiles * n);
    if (zs < 1) zs = 1;
    int64_t chunk = (M + zs - 1) / zs;
    if (chunk < 256) chunk = 256;
    if (chunk > 32 * (WGRAD_MASK_WORDS - 2)) chunk = 32 * (WGRAD_MASK_WORDS - 2);
    chunk = (chunk + 31) / 32 * 32;
    if ((int64_t)chunk * 4 * 2048 >= 0x7fffffff) return FS_ERR_ARG;
    a.kchunk = (int)chunk;
    m.zs = (int)((M + chunk - 1) / chunk);
    m.nseg = n;
    dim3 grid(xt128, ytiles, m.zs * n);
    if (g_wgrad_xcd == 3) {        // all tiles of one pixel split on one XCD (xcd_yt < 0 carries -x tiles)
      m.a.xcd_xt = xt128 * ytiles; m.a.xcd_yt = -xt128; m.a.xcd_groups = m.zs * n;
      grid = dim3((unsigned)(ceil_div(m.a.xcd_groups, 8) * 8 * xt128 * ytiles), 1, 1);
    } else if (g_wgrad_xcd) {
      m.a.xcd_xt = xt128; m.a.xcd_yt = ytiles; m.a.xcd_groups = ytiles * m.zs * n;
      grid = dim3((unsigned)(ceil_div(m.a.xcd_groups, 8) * 8 * xt128), 1, 1);
    }
    if (g_wgrad_w8) hipLaunchKernelGGL((conv_wgrad_split_kernel<SWCfg128W8, true, true>), grid, dim3(512), 0, stream, m);
    else hipLaunchKernelGGL((conv_wgrad_split_kernel<SWCfg128S, true, true>), grid, dim3(256), 0, stream, m);
    const int rc = fs_launch_status();
    if (rc) return rc;
  }
  return FS_OK;
}

// mode 0: OIHW -> forward packed; mode 1: OIHW -> data-gradient packed (rows = Cin);
// mode 2: packed (forward layout) -> OIHW, optionally accumulating.  srcC splits Cin for modes 0/2.
extern "C" int fsraft_pack_conv_weights(const PackJob* jobs, int njobs, hipStream_t stream) {
  if (njobs < 0 || (njobs && !jobs)) return FS_ERR_ARG;
  for (int i = 0; i < njobs; ++i) {
    const PackJob& j = jobs[i];
    const int m = j.mode;
    if (!(m == 0 || m == 1 || m == 2 || m == 3 || m == 10 || m == 11) || !j.wpk || j.npiece < 1 || j.npiece > 3) return FS_ERR_ARG;
    for (int p = 0; p < j.npiece; ++p) if (!j.w[p] || j.rows[p] < 1) return FS_ERR_ARG;
    if (m == 3) continue;
    if (j.nsrc < 1 || j.nsrc > 3 || j.kh < 1 || j.kw < 1 || j.cin_full < 1) return FS_ERR_ARG;
    if ((j.flags & 1) && m < 10) return FS_ERR_ARG;
    if ((j.flags & 2) && (j.kh != 2 || j.kw != 2 || j.nsrc != 1 || j.srcC[0] != 4 * j.cin_full || j.srcOff[0] != 0)) return FS_ERR_ARG;
    if (!(j.flags & 2)) for (int s = 0; s < j.nsrc; ++s) if (j.srcC[s] < 1 || j.srcOff[s] < 0 || j.srcOff[s] + j.srcC[s] > j.cin_full) return FS_ERR_ARG;
  }
  for (int i0 = 0; i0 < njobs; i0 += PACK_JOBS) {
    PackJobs tab{};
    const int n = njobs - i0 < PACK_JOBS ? njobs - i0 : PACK_JOBS;
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
      tab.j[i] = jobs[i0 + i];
      const PackJob& j = tab.j[i];
      int64_t cout = 0, cin = 0;
      for (int p = 0; p < j.npiece; ++p) cout += j.rows[p];
      for (int s = 0; s < j.nsrc; ++s) cin += (j.srcC[s] + 31) / 32 * 32;
      const int64_t tot = j.mode == 3 ? cout : ((cout + 31) / 32 * 32) * ((cin + 31) / 32 * 32) * j.kh * j.kw;
      most = tot > most ? tot : most;
    }
    int blocks = (int)((most + 1023) / 1024);
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(pack_jobs_kernel, dim3(blocks, n), dim3(256), 0, stream, tab);
  }
  return fs_launch_status();
}

extern "C" int fsraft_pack_conv_weight(float* w_oihw, float* wpk, int Cout, int Cin, int KH, int KW, const int* srcC,
                                       int nsrc, int mode, int accumulate, const unsigned* w_amax, hipStream_t stream) {
  if (!w_oihw || !wpk || mode < 0 || (mode > 2 && mode != 10 && mode != 11) || ((uintptr_t)w_amax & 3)) return FS_ERR_ARG;
  PackArgs a{};
  a.amax = w_amax;
  a.split = mode >= 10;                      // modes 10 / 11: split-bf16 variants of modes 0 / 1
  if (mode >= 10) mode -= 10;
  if (a.split && mode > 1) return FS_ERR_ARG;
  a.w = w_oihw; a.wpk = wpk; a.Cout = Cout; a.Cin = Cin; a.taps = KH * KW; a.mode = mode; a.accumulate = accumulate;
  if (mode == 1) {
    a.nsrc = 1; a.C[0] = Cout; a.C[1] = a.C[2] = 0;
    a.Ktot = a.taps * (((Cout + 31) / 32) * 32);
    a.rows = Cin;
  } else {
    if (!srcC || nsrc < 1 || nsrc > 3) return FS_ERR_ARG;
    int tot = 0;
    for (int s = 0; s < 3; ++s) { a.C[s] = s < nsrc ? srcC[s] : 0; tot += a.C[s]; }
    if (tot != Cin) return FS_ERR_ARG;
    a.nsrc = nsrc;
    a.Ktot = conv_ktot(srcC, nsrc, a.taps);
    a.rows = Cout;
  }
  const int64_t total = (int64_t)a.rows * a.Ktot;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return fs_launch_status();
}
