/* fsraft.h -- C ABI of libfsraft.so, the MI355X (gfx950) implementation of the RAFT
 * hot path of iwbn/flow-supervisor.
 *
 * Everything here takes plain device pointers, sizes and a hipStream_t; no torch types
 * cross this boundary.  All tensors are fp32.  Every function only enqueues work on
 * `stream` (no allocation, no synchronisation -> safe under hipGraph capture) and
 * returns 0 (FSRAFT_OK), 1 (bad argument) or 2 (launch failed).
 *
 * Threads and devices (the reference's multi-GPU caller is nn.DataParallel, pytorch/train.py:192: one host thread per device in
 * one process).  A call launches on the CALLER's current device -- bind the device of the pointers first (hipSetDevice) -- and
 * on the stream it is given.  Entry points may be called concurrently from several host threads: no per-call state is shared
 * (scratch travels in the call, fsraft_conv_desc.ws; fsraft_conv_workspace and the statistics request of
 * fsraft_conv_forward_stats are per calling thread).  What IS process-wide is configuration -- fsraft_set_arithmetic and the
 * switches of fsraft_tuning.h -- to be set once, before worker threads start.
 *
 * Each entry point names the reference interface it replaces, as file:line under
 * /root/reference.  The Python binding a maintainer would add is shown in
 * INTEGRATION.md; ours lives in flow_supervisor_amd/_lib.py.
 */
#ifndef FSRAFT_H
#define FSRAFT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define FSRAFT_OK 0
#define FSRAFT_ERR_ARG 1
#define FSRAFT_ERR_LAUNCH 2

/* Version of this header's structs and signatures.  Round 6 (version 6) added the amax words below: trailing fields of
 * fsraft_conv_desc / fsraft_pack_job and trailing `const unsigned*` arguments of the GEMM-shaped entry points.  A binding
 * compiled against another version must not call in (the descriptor structs carry no size field). */
#define FSRAFT_ABI_VERSION 6
int fsraft_abi_version(void);

/* ---- amax words: the scales of the split arithmetic ----------------------------------------------------------------------
 * The reference multiplies in fp32 (pytorch/core/corr.py:52-60 matmul, every nn.Conv2d of update.py:6-136).  Here a GEMM-shaped
 * kernel carries every fp32 operand x as two fp16 pieces of x * s (hi = fp16(x s), lo = fp16(x s - hi)) and evaluates a product
 * as (a_hi b_hi + a_hi b_lo + a_lo b_hi) / (s_a s_b) on v_mfma_f32_*_f16 with fp32 accumulation: ~2^-22 per product, the fp32
 * GEMM's accuracy class (csrc/split_arith.hpp; fsraft_set_arithmetic(0) selects exact fp32 MFMA instead).  s is a power of
 * two per TENSOR, derived on the device from the tensor's "amax word": ONE `unsigned` in device memory holding the bit pattern
 * of a magnitude w with max |x| < 2 w (an upper bound, or what producers raised it to: within [1/2, 4] of the maximum).  s maps
 * w into [2^13, 2^14): every element lands below 2^15, nothing overflows fp16, elements within 2^-14 of the largest keep all 22
 * bits, smaller ones degrade gracefully (fp16 subnormals) to a floor of 2^-36 of the largest.
 *   - entry points that READ a tensor through the matrix pipe take its word (`..._amax`, const unsigned*); NULL = "the caller
 *     vouches |x| < 2^15": scale 1;
 *   - entry points that WRITE one can raise a word (fsraft_conv_desc.dst_amax, `..._amax` outputs: an atomicMax, one binade
 *     up, whenever a workgroup stored a magnitude whose exponent exceeds the word's -- hence the factors), so that a chain of
 *     kernels needs no extra pass; the word must be zero (or an earlier bound) before the producer runs;
 *   - fsraft_amax / fsraft_amax_jobs compute words of tensors that come from elsewhere (images, parameters, gradients handed in
 *     by autograd); fsraft_amax_scaled derives a bound from another word (dst = max(dst, factor * src)).
 * Records (fsraft_to_records, the weight packs, the gradient volume) hold pieces of x * s: whoever reads them is given the word
 * they were split with.  Words are written by kernels and read by later kernels on the same stream: no host synchronisation. */
int fsraft_amax(const float* x, int64_t rows, int64_t C, int64_t ld, unsigned* word, hipStream_t stream);     /* element (r, c) at x[r * ld + c] */
int fsraft_amax_jobs(const float* const* ptrs, const int64_t* rows, const int64_t* C, const int64_t* ld, unsigned* const* words,
                     int n, hipStream_t stream);                        /* n tensors, 32 per launch; several may share a word */
int fsraft_amax_scaled(const unsigned* src, float factor, unsigned* dst, hipStream_t stream);

/* ---- all-pairs correlation volume + pyramid ---------------------------------------
 * Replaces CorrBlock.__init__ / CorrBlock.corr, pytorch/core/corr.py:13-27, 52-60
 * (torch.matmul + 3x avg_pool2d) and TF calc_all_field, raft/allfield.py:61-92.
 * fmap1, fmap2: [B,C,H,W].  levels[l]: [B*H*W, H>>l, W>>l] (floor), l < num_levels <= 4. */
int fsraft_corr_build(const float* fmap1, const float* fmap2, float* const* levels, int num_levels,
                      int B, int C, int H, int W, const unsigned* amax1, const unsigned* amax2, hipStream_t stream);

/* Backward of the pooling chain: dlevels[0] += unpool(dlevels[1..]) in place.
 * (autograd of F.avg_pool2d, pytorch/core/corr.py:25-27) */
/* Pyramid of an existing level-0 volume (raft/allfield.py:94-106 build_pyramid; the backward-flow pyramid of the transposed
 * volume at raft/semi.py:250-251): levels[0] = [rows][H2][W2] given, levels[l] = 2x2 averages of levels[l-1], floor sizes. */
int fsraft_corr_pool_pyramid(float* const* levels, int num_levels, int64_t rows, int H2, int W2, hipStream_t stream);
/* TensorFlow semantics of the same pyramid (raft/allfield.py:99-104, tf.nn.avg_pool2d(level 0, s, s, 'SAME'), s = 2, 4, ...):
 * levels[l] = [rows][ceil(H2/2^l)][ceil(W2/2^l)], partial edge windows averaged over their in-range elements, and the
 * forward lookup over a pyramid of those sizes (raft/allfield.py:109-135). */
int fsraft_corr_pool_pyramid_same(float* const* levels, int num_levels, int64_t rows, int H2, int W2, hipStream_t stream);
int fsraft_corr_lookup_fwd_same(float* const* levels, int num_levels, const float* coords, int64_t coords_bs,
                                int64_t coords_cs, int64_t coords_ps, float* out, int nhwc_out, int B, int H, int W,
                                int radius, hipStream_t stream);
int fsraft_corr_unpool_bwd(float* const* dlevels, int num_levels, int B, int H, int W, hipStream_t stream);

/* ---- the same volume in the tiled-row layout (what CorrBlock runs on) ------------------------------------------------
 * Every query i = (b, y, x) owns ONE row of P floats: its pyramid levels back to back, each level cut into 4x4-cell
 * tiles of 64 contiguous bytes (tiles x-fastest; ceil(h_l/4) x ceil(w_l/4) tiles, pad cells beyond the reference's
 * floor-halved h_l x w_l):
 *     cell (y, x) of level l  at  row[off[l] + ((y>>2) * tw[l] + (x>>2)) * 16 + (y&3) * 4 + (x&3)]
 * fsraft_vol_layout fills out[24] = {nlev, H, W, P, h[4], w[4], th[4], tw[4], off[4]}; P % 32 == 0.
 * A (2r+2)^2 lookup window touches ~0.8 KB of 128-byte lines here against ~1.7 KB in the reference's row-major
 * [B*N,1,h_l,w_l] tensors (pytorch/core/corr.py:19-27), which fsraft_corr_build above still produces for API twins. */
int fsraft_vol_layout(int H, int W, int num_levels, int* out);
/* vol [B*H*W][P] <- all-pairs volume + pyramid (pad cells of V: unspecified, never read by the lookup) */
int fsraft_corr_build_tiled(const float* fmap1, const float* fmap2, float* vol, int num_levels, int B, int C, int H, int W,
                            const unsigned* amax1, const unsigned* amax2, hipStream_t stream);
/* the same from pre-split feature maps f1r, f2r = [B][H*W][C/32] records (fsraft_to_records of the channels-last maps,
 * C % 32 == 0): both operands staged by LDS-DMA on the record GEMM core, 256 queries x an 8x16 target patch per workgroup */
int fsraft_corr_build_rec(const void* f1r, const void* f2r, float* vol, int num_levels, int B, int C, int H, int W,
                          const unsigned* amax1, const unsigned* amax2 /* the words f1r / f2r were split with */, hipStream_t stream);
/* CorrBlock.__call__ (pytorch/core/corr.py:29-50) on that layout; out [B,H,W,L*(2r+1)^2] channels-last; one wave per query.
 * add_grid != 0: `coords` holds the FLOW and the query position is pixel grid + flow (the caller's coords0 + flow,
 * pytorch/core/raft.py:121-131, never materialised) */
int fsraft_corr_lookup_tiled_fwd(const float* vol, int num_levels, const float* coords, int64_t coords_bs, int64_t coords_cs,
                                 int64_t coords_ps, float* out, int B, int H, int W, int radius, int add_grid,
                                 unsigned* out_amax /* nullable: word of `out`, raised */, hipStream_t stream);
/* Gradient volume of n lookups at once (grid_sampler_2d_backward w.r.t. the volume, pytorch/core/utils/utils.py:57-71, for
 * all iterations of a step): dvol [B*H*W][P] = (or +=, accumulate != 0) sum_t (d out_t / d V)^T dout_t, pad cells zero;
 * dout[t] is [B,H,W,CH] channels-last, coords[t] element (b, c, pix) at coords[t][b*s0 + c*s1 + pix*s2] with
 * (s0, s1, s2) = coords_str[3t .. 3t+2].  n <= 16 per call.  Each row is accumulated in LDS and written once; records != 0:
 * as [32 hi | 32 lo] fp16 records, the operand format of fsraft_gemm_rec_nt / _tn below.  Only queries [q0, q0 + nq)
 * (nq == 0: all from q0) are built, into dvol rows 0 .. nq-1 -- AlternateCorrBlock's backward walks the queries in chunks so
 * that no O(N^2) buffer exists.  qlist (nullable): caller-owned scratch of 1 + rows unsigned.  With it, one wave per query
 * builds the row from the bounding boxes of its lookups' windows (a few KB of LDS instead of the whole row) and queries whose
 * lookups spread beyond the box are listed there for the row-at-a-time kernel; without it every query takes that kernel.
 * wmask (nullable; needs qlist, records, no accumulate, all queries or a chunk inside one image): the record bitmap of
 * fsraft_corr_bwd_ktiles for the same queries -- only the
 * records either list GEMM reads are written, the rest of dvol stays untouched (it would be zero records nobody reads). */
int fsraft_corr_dvol_build(const float* const* dout, const float* const* coords, const int64_t* coords_str, int n, float* dvol,
                           int num_levels, int B, int H, int W, int radius, int accumulate, int records, int add_grid,
                           int64_t q0, int64_t nq, unsigned* qlist, const unsigned* wmask,
                           const unsigned* dvol_amax /* records: a word bounding |dvol| -- (lookups of the step) x max |dout| does,
                                                        a cell collects at most one unit of bilinear weight per lookup */,
                           hipStream_t stream);
/* Backward of matmul + avg_pool2d chain (pytorch/core/corr.py:21-27, 52-60) without un-pooling the volume gradient:
 *   f2cat [B][C][P]: level-l cell = mean of fmap2 over its 2^l x 2^l pixels (0 in pad cells), so that
 *   dF1[b][c][i] = s * sum_p f2cat[b][c][p] * dvol[b][i][p]   (one NT GEMM, K = P)  and
 *   d2cat[b][p][c] = s * sum_i dvol[b][i][p] * f1[b][i][c]    (one TN GEMM, M = P);
 *   fsraft_corr_dfmap2: d2 [B][H*W][C] = sum_l 4^-l * d2cat[b][cell_l(y>>l, x>>l)][c] over the levels whose cell exists. */
int fsraft_corr_f2cat(const float* fmap2, float* f2cat, int num_levels, int B, int C, int H, int W, hipStream_t stream);
int fsraft_corr_dfmap2(const float* d2cat, float* d2, int num_levels, int B, int C, int H, int W, hipStream_t stream);
/* f2cat directly as records [B][C][P / 32] x ([32 hi | 32 lo] fp16 pieces) -- fsraft_corr_f2cat followed by fsraft_to_records in one
 * pass (the plane pooled in LDS by the reference's recursion, corr.py:24-26).  Planes of at most 12288 pixels (H * W);
 * FS_ERR_ARG above that, the caller then takes the two calls. */
int fsraft_corr_f2cat_rec(const float* fmap2, void* f2r, int num_levels, int B, int C, int H, int W,
                          const unsigned* amax2 /* word of fmap2: bounds its pooled means too */, hipStream_t stream);

/* ---- radius-r pyramid lookup --------------------------------------------------------
 * Replaces CorrBlock.__call__, pytorch/core/corr.py:29-50 (+ bilinear_sampler,
 * core/utils/utils.py:57-71) and TF smurf_corr_block, raft/allfield.py:109-135.
 * coords element (b,c,pix) is read at coords[b*bs + c*cs + pix*ps] (c=0: x, c=1: y).
 * out: [B, 4*(2r+1)^2, H, W] (nhwc == 0) or [B, H, W, 4*(2r+1)^2] (nhwc != 0). radius in {3,4}. */
int fsraft_corr_lookup_fwd(float* const* levels, int num_levels, const float* coords, int64_t coords_bs,
                           int64_t coords_cs, int64_t coords_ps, float* out, int nhwc_out, int B, int H, int W,
                           int radius, hipStream_t stream);
/* dlevels[l] += (d out / d V_l)^T dout.  (grid_sampler_2d_backward w.r.t. the volume) */
int fsraft_corr_lookup_bwd(float* const* dlevels, int num_levels, const float* coords, int64_t coords_bs,
                           int64_t coords_cs, int64_t coords_ps, const float* dout, int nhwc_in, int B, int H, int W,
                           int radius, hipStream_t stream);

/* ---- memory-efficient correlation (no N x N volume) ----------------------------------
 * Replaces alt_cuda_corr.forward / .backward, pytorch/alt_cuda_corr/correlation.cpp:23-54
 * (kernels correlation_kernel.cu:18-119, 122-256).  Same argument meaning:
 * fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] channels-last, coords [B,N,H1,W1,2] (x,y; N coordinate sets per query
 * pixel, correlation_kernel.cu:34,59 -- the reference's caller passes N = 1, corr.py:84),
 * corr [B,N,(2r+1)^2,H1,W1], UNSCALED (caller divides by sqrt(C), corr.py:91). */
int fsraft_altcorr_fwd(const float* fmap1, const float* fmap2, const float* coords, float* corr, int B, int N, int H1,
                       int W1, int H2, int W2, int C, int radius, hipStream_t stream);
/* fmap1_grad [B,H1,W1,C] is overwritten; fmap2_grad [B,H2,W2,C] must be zeroed by the caller
 * (it is accumulated with atomics); coords get no gradient (the reference returns zeros). */
int fsraft_altcorr_bwd(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad,
                       float* fmap1_grad, float* fmap2_grad, int B, int N, int H1, int W1, int H2, int W2, int C,
                       int radius, hipStream_t stream);

/* AlternateCorrBlock.__call__ (pytorch/core/corr.py:74-91: four alt_cuda_corr.forward calls, stack, reshape, / sqrt(C)) as ONE
 * launch that writes the channels-last [B,H,W,L*(2r+1)^2] lookup: fmap1 [B,H,W,C], fmap2_levels[l] [B,H>>l,W>>l,C] (the
 * average-pooled maps), coords as in fsraft_corr_lookup_tiled_fwd. */
int fsraft_altcorr_fused_fwd(const float* fmap1, const float* const* fmap2_levels, int num_levels, const float* coords,
                             int64_t coords_bs, int64_t coords_cs, int64_t coords_ps, int add_grid, float* out, int B, int H, int W,
                             int C, int radius, hipStream_t stream);

/* The same lookup with the dot products of a tile of queries against a region of target rows as one small GEMM on the matrix
 * pipe (split arithmetic on pre-split operands, as the volume build): f1r [B][H*W][C/32 records] and f2r_levels[l]
 * [B][(H>>l)*(W>>l)][C/32 records] = fsraft_to_records of the channels-last maps, which are passed as well (window positions
 * outside a tile's region -- flow discontinuities -- are taken from them in fp32).  C % 32 == 0, C <= 256.
 * regime: NULL, or 8 ints of device memory, zero before the first call and private to the caller's stream.  With it the call
 * dispatches by flow regime: a statistic kernel counts the queries whose level-0 window leaves their tile's region, and of the
 * two lookup kernels enqueued behind it (this one and fsraft_altcorr_fused_fwd's) the one the decision names does the work, the
 * other returns at once.  After the call regime[0] = 1 if the fp32 kernel ran, regime[4] / regime[5] = uncovered / counted queries. */
int fsraft_altcorr_mfma_fwd(const void* f1r, const void* const* f2r_levels, const float* fmap1, const float* const* fmap2_levels,
                            int num_levels, const float* coords, int64_t coords_bs, int64_t coords_cs, int64_t coords_ps,
                            int add_grid, float* out, int B, int H, int W, int C, int radius,
                            const unsigned* amax1, const unsigned* const* amax2_levels /* the words f1r / each f2r level were split with */,
                            int* regime, hipStream_t stream);

/* ---- convex 8x upsampler -------------------------------------------------------------
 * Replaces RAFT.upsample_flow, pytorch/core/raft.py:72-83 and UpsampleConvexWithMask,
 * raft/upsample.py:11-41.  flow element (n,c,pix) at flow[n*bs + c*cs + pix*ps];
 * mask channels-last [N,H,W,576]; up [N,2,8H,8W]. */
int fsraft_upsample_fwd(const float* flow, int64_t flow_bs, int64_t flow_cs, int64_t flow_ps,
                        const float* mask_nhwc, float* up, int N, int H, int W, hipStream_t stream);
/* dmask_nhwc [N,H,W,576], dflow [N,2,H,W]; scratch: N*H*W*18 floats. */
int fsraft_upsample_bwd(const float* flow, int64_t flow_bs, int64_t flow_cs, int64_t flow_ps,
                        const float* mask_nhwc, const float* dup, float* dmask_nhwc, float* dflow, float* scratch,
                        int N, int H, int W, unsigned* dmask_amax /* nullable: word of dmask_nhwc, raised */, hipStream_t stream);
/* upflow8, pytorch/core/utils/utils.py:80-82 (raft-small). flow [N,C,H,W] -> up [N,C,8H,8W]. */
int fsraft_upflow8_fwd(const float* flow, float* up, int N, int C, int H, int W, hipStream_t stream);
int fsraft_upflow8_bwd(const float* dup, float* dflow, int N, int C, int H, int W, hipStream_t stream);

/* ---- update-block convolutions (implicit GEMM, fp32 MFMA) ----------------------------
 * Replace every nn.Conv2d of pytorch/core/update.py:6-136 together with the cat / ReLU /
 * sigmoid / tanh / GRU-gate elementwise ops around them.  Activations are channels-last
 * with pitch ld (ld % 4 == 0, padding channels zero). */
typedef struct fsraft_conv_desc {
  const float* src[3]; int srcC[3]; int srcld[3]; int nsrc;   /* concatenated inputs            */
  const float* wpk; const float* bias;                        /* packed weights, bias[N] or NULL */
  const float* wpk_split;                                     /* same, packed as [hi | lo] fp16 records (modes 10/11), or NULL */
  int B, H, W, KH, KW, N;                                     /* N = output channels             */
  float* dst[3]; int64_t dst_bs[3]; int64_t dst_ps[3]; int64_t dst_cs[3];
  int dst_n0[3]; int dst_acc[3]; int ndst;                    /* output channel ranges           */
  int relu; float alpha;                                      /* y = relu?(alpha*(acc+bias))     */
  int epi;                                                    /* 0 plain, 2 GRU z|r, 3 GRU q     */
  const float* h; int ldh; const float* z; int ldz;
  float* aux1; int ld1; float* aux2; int ld2; int hid;
  const float* pre; int ldpre;                                /* epi 2/3: [M][ldpre] addend to the pre-activation, or NULL */
  const float* rmask[3]; int ldmask[3]; int maskc[3];         /* epi 0, per destination: ReLU-backward mask (zero column
                                                                 j < maskc of the range where rmask[m*ldmask+j] <= 0), or NULL */
  const float* wpk_frag;                                      /* optional: wpk_split re-ordered for direct fragment loads --
                                                                 [k-tile][32-row block][hi k0-15, hi k16-31, lo k0-15, lo k16-31]
                                                                 [lane = 32 * (k half) + row][16 B], rows zero-padded to 32;
                                                                 enables the resident-patch 3x3 kernel (one source, 33..64
                                                                 channels in, N <= 64, large B*H*W).  NULL: never used */
  int pad_h1, pad_w1;                                         /* 0: taps centred (KH/2, KW/2 rows / columns above / left of
                                                                 the output pixel); else 1 + that count -- even kernel sizes:
                                                                 a 2x2 kernel has pad 1 forward and pad 0 in its data gradient */
  float* ws; int64_t ws_floats;                               /* scratch of THIS call for the split-K route at small pixel
                                                                 counts (16-byte aligned, on the device the call runs on, not
                                                                 shared with a call on another stream); NULL: the buffer the
                                                                 calling thread registered with fsraft_conv_workspace, if any */
  /* amax words (see the top of this file).  src_amax[s]: word of source s (one scale is used for all sources: the largest);
   * w_amax: the word wpk_split / wpk_frag were packed with; dst_amax[i] (nullable): raised to the largest magnitude stored
   * into destination i -- GRU epilogues: [0] the new hidden state (epi 3), [1] r*h (epi 2); the gates are bounded by 1 */
  const unsigned* src_amax[3]; const unsigned* w_amax; unsigned* dst_amax[3];
} fsraft_conv_desc;

int fsraft_conv_ktot(const int* srcC, int nsrc, int KH, int KW);
int fsraft_conv_forward(const fsraft_conv_desc* d, hipStream_t stream);
/* The same convolution; where the chosen kernel's tiles lie inside one image (the resident-patch kernels) and the epilogue is
 * the raw result (one destination, no bias / ReLU / scale / mask / accumulation) it also adds the per-image column sums of the
 * result and of its squares to sum / sq [B * slots][N] (fp32, ZERO on entry; workgroups spread over the `slots` rows of their
 * image) -- the statistics of the InstanceNorm2d that follows the convolution in pytorch/core/extractor.py:13-57, 118-170 --
 * and sets *done = 1; *done = 0: not carried, the caller computes them.  Pass the buffers to fsraft_inorm_relu_cl_fwd
 * (slots = 8, have_sums = *done). */
int fsraft_conv_forward_stats(const fsraft_conv_desc* d, float* sum, float* sq, int slots, int* done, hipStream_t stream);
/* dwpk[Cout][Ktot] += dY^T im2col(src);  dbias (nullable): dbias[co] += sum over pixels of dY[:, co] */
int fsraft_conv_wgrad(const float* dy, int ldy, int Cout, const float* const* src, const int* srcC,
                      const int* srcld, int nsrc, float* dwpk, float* dbias, int B, int H, int W, int KH, int KW,
                      const unsigned* dy_amax, const unsigned* const* src_amax /* [nsrc] words, or NULL */, hipStream_t stream);
/* The same reduction over nseg (dY, X) pairs of identical shape -- the iterations of one training step -- in one launch:
 * dwpk += sum_t dY_t^T im2col(X_t).  src[t * nsrc + s] is source s of pair t. */
int fsraft_conv_wgrad_multi(const float* const* dy, int nseg, int ldy, int Cout, const float* const* src,
                            const int* srcC, const int* srcld, int nsrc, float* dwpk, float* dbias, int B, int H, int W,
                            int KH, int KW, const unsigned* const* dy_amax /* [nseg] or NULL */,
                            const unsigned* const* src_amax /* [nseg * nsrc] or NULL */, hipStream_t stream);
/* mode 0: OIHW -> packed forward; 1: OIHW -> packed data-gradient; 2: packed -> OIHW (+=);
 * modes 10 / 11: as 0 / 1 but every 32-k run stored as [32 hi | 32 lo] fp16 pieces of w * scale(w_amax) (split GEMM core) */
int fsraft_pack_conv_weight(float* w_oihw, float* wpk, int Cout, int Cin, int KH, int KW, const int* srcC,
                            int nsrc, int mode, int accumulate, const unsigned* w_amax /* modes 10 / 11: word of w_oihw */,
                            hipStream_t stream);

/* Batched form: every packed matrix of a module (or, mode 2, every weight gradient) in one launch per 16 jobs, read in place
 * from / written in place to the parameter-shaped tensors -- no torch.cat of fused layers, no channel gather.
 *   w[p], rows[p]   up to three OIHW tensors stacked along the output channels (z|r gates, flow-head|mask-head), all with
 *                   cin_full input channels and kh x kw taps
 *   srcC, srcOff    GEMM source s = input channels [srcOff[s], srcOff[s] + srcC[s]) of those tensors
 *   mode            0 / 1 / 10 / 11 as above; 2: wpk -> w (w = scale * wpk, or += when accumulate); 3: wpk = the rows[p]-long
 *                   vectors w[p] concatenated (fused biases)
 *   flags           bit 0: split pack in fragment order (resident-patch 3x3 kernel, rows zero-padded to a multiple of 32);
 *                   bit 1: w is the [N][C][3][3] weight of a stride-2, pad-1 convolution and the packed matrix is its
 *                   stride-1 2x2 equivalent over the space-to-depth input ([N][4C][2][2], fsraft_space_to_depth2):
 *                   cin_full = C, kh = kw = 2, one source of 4C channels */
typedef struct {
  float* w[3]; int rows[3]; int npiece;
  float* wpk;
  int cin_full, kh, kw;
  int srcC[3], srcOff[3], nsrc;
  int mode, flags;
  float scale; int accumulate;
  const unsigned* amax;          /* modes 10 / 11: word bounding the job's weights (fsraft_amax_jobs over its pieces); NULL: scale 1 */
} fsraft_pack_job;
int fsraft_pack_conv_weights(const fsraft_pack_job* jobs, int njobs, hipStream_t stream);

/* Convolutions with 2 output channels (FlowHead.conv2, pytorch/core/update.py:10: 3x3, hidden -> 2) as per-pixel dot
 * products instead of a padded GEMM tile.  x channels-last [M][ld]; w_oihw [2][C][3][3]; out element (b,o,pix) at
 * out[b*obs + o*ocs + pix*ops].  C % 4 == 0, C <= 512. */
int fsraft_conv_small_fwd(const float* x, int ld, int C, const float* w_oihw, const float* bias, float* out, int64_t obs,
                          int64_t ocs, int64_t ops, int N, int B, int H, int W, int KH, int KW, hipStream_t stream);
/* dwpk[o][tap*ceil32(C) + c] += sum over nseg (dy_t, x_t) pairs and pixels of dy_t[pix][o] * x_t[pix+shift][c];
 * dbias[o] += sum dy (nullable) */
int fsraft_conv_small_wgrad(const float* const* dy, const float* const* x, int nseg, int ldy, int ldx, int C, float* dwpk,
                            float* dbias, int N, int B, int H, int W, int KH, int KW, hipStream_t stream);
/* Data gradient of fsraft_conv_small_fwd (the flow head's 256 -> 2 convolution, pytorch/core/update.py:6-14, as autograd's
 * conv2d backward w.r.t. its input): dx[pix][c] = sum_{o < 2, tap} w_oihw[o][c][tap] * dy[pix - shift(tap)][o], zero where
 * relu_src <= 0 (nullable: the ReLU in front of the convolution, pitch ldm).  dy channels-last with pitch ldy >= 2; dx
 * channels-last with pitch lddx, 16-byte aligned.  C % 4 == 0, C <= 256, N == 2, 3x3. */
int fsraft_conv_small_dgrad(const float* dy, int ldy, const float* w_oihw, float* dx, int lddx, const float* relu_src, int ldm,
                            int C, int N, int B, int H, int W, int KH, int KW, unsigned* dx_amax /* nullable: word of dx, raised */,
                            hipStream_t stream);

/* Scratch buffer for the split-K route of the convolutions at small pixel counts (one or two pairs per GPU: the layer's
 * k-tiles are dealt to several workgroups per tile, which park partial tiles here; a second kernel adds them and applies the
 * layer's epilogue).  The library allocates nothing: the caller owns `ws` (16-byte aligned, `floats` fp32) and keeps it alive.
 * Preferred: hand it over per call in fsraft_conv_desc.ws.  This entry point registers a buffer for the CALLING THREAD only
 * (thread-local; NULL clears it) and serves descriptors whose ws is NULL -- a worker thread per device, as in the reference's
 * nn.DataParallel caller (pytorch/train.py:192), registers its own device's buffer and never sees another thread's. */
int fsraft_conv_workspace(float* ws, int64_t floats);

/* ---- arithmetic of the dense contractions -------------------------------------------------------------------------
 * Storage and accumulation are fp32 everywhere.  mode 1 (default): every fp32 product of the GEMM-shaped kernels (volume
 * build and its backward, the update block's / encoders' convolutions, their weight gradients, the GMA GEMMs) is evaluated
 * as three fp16 MFMA products of scaled operands with fp32 accumulation (hi*hi + hi*lo + lo*hi, ~2^-22 relative error per
 * product: "amax words" at the top of this file; rounds 1-5 used bf16 pieces, 2^-17);
 * mode 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32, a pure fmaf chain) -- the test / reference mode.  Process-wide; not
 * to be flipped while kernels of another thread are being enqueued.  Replaces nothing in the reference (its
 * torch.backends.cuda.matmul.allow_tf32 default plays the same role on the reference's hardware). */
int fsraft_set_arithmetic(int mode);
int fsraft_get_arithmetic(void);   /* 1 / 0 as above */

/* ---- batched fp32 GEMM (volume backward: autograd of torch.matmul, corr.py:57) ------- */
int fsraft_gemm_f32(const float* A, int64_t lda, int64_t sA, const float* Bm, int64_t ldb, int64_t sB, float* C,
                    int64_t ldc, int64_t sC, int batch, int M, int N, int K, int trans_b, float alpha,
                    int accumulate, const unsigned* a_amax, const unsigned* b_amax /* trans_b: the split kernel's words */,
                    hipStream_t stream);

/* C[b][m][n] = alpha * sum_k A[b][k][m] * Bm[b][k][n] (both k-major) on the split (fp16x3) core. */
int fsraft_gemm_tn_split(const float* A, int64_t lda, int64_t sA, const float* Bm, int64_t ldb, int64_t sB, float* C,
                         int64_t ldc, int64_t sC, int batch, int M, int N, int K, float alpha, int accumulate,
                         const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream);

/* ---- GEMM on pre-split ("record") operands -------------------------------------------------------------------------
 * A record is 32 consecutive k of one row as [32 x fp16 hi | 32 x fp16 lo] (hi = fp16(x s), lo = fp16(x s - hi), s the scale of
 * the tensor's amax word): 128 bytes, the bytes the 32 floats took.  Producers split once; the GEMM stages by LDS-DMA (no
 * conversion, no VGPR round trip), evaluates a_hi b_hi + a_hi b_lo + a_lo b_hi on fp16 MFMA with fp32 accumulation
 * (csrc/gemm_rec.hpp) and divides s_a s_b out of the result: a_amax / b_amax are the words A / B were split with.
 * fsraft_to_records: src [rows][K] fp32 (pitch ld floats) -> dst [rows][ceil(K/32)] records (tail of the last record zero),
 * row pitch dst_ld floats (0 = dense; activation tensors use an ODD number of 128-byte lines per row so that the rows of a
 * k-tile spread over all L2 channels instead of every 4th / 8th line).
 * fsraft_gemm_rec_nt: C[b][m][n] = alpha * sum_k A[b][m][k] B[b][n][k]; A [batch][M] rows of K/32 records (row pitch lda
 * floats), B [batch][N] rows (pitch ldb; 0 = K), K % 32 == 0, sA / sB batch strides in BYTES.  ksplit > 1: K split over workgroups, partial tiles added with fp32
 * atomics (C zeroed first unless accumulate != 0). */
int fsraft_to_records(const float* src, int64_t ld, void* dst, int64_t dst_ld, int64_t rows, int K, const unsigned* amax,
                      hipStream_t stream);
int fsraft_gemm_rec_nt(const void* A, int64_t lda, int64_t sA, const void* Bm, int64_t ldb, int64_t sB, float* C, int64_t ldc,
                       int64_t sC, int batch, int M, int N, int K, float alpha, int ksplit, int accumulate,
                       const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream);
/* C[b][m][n] = alpha * sum_k A[b][k][m] B[b][k][n]: both operands k-major, A [batch][K][lda floats] with the records along m,
 * B [batch][K][ldb floats] with the records along n (lda, ldb multiples of 32; K arbitrary).  Fragments are gathered with the
 * transposed LDS read (ds_read_b64_tr_b16). */
int fsraft_gemm_rec_tn(const void* A, int64_t lda, int64_t sA, const void* Bm, int64_t ldb, int64_t sB, float* C, int64_t ldc,
                       int64_t sC, int batch, int M, int N, int K, float alpha, int ksplit, int accumulate,
                       const unsigned* a_amax, const unsigned* b_amax, hipStream_t stream);

/* The two contractions above over LISTED k-tiles only (32 k each, ascending): klist[(b * tiles + tile) * kl_stride + i],
 * i < kcount[b * tiles + tile], one list per tile of the sparse operand -- the 128-row tiles of B resp. 128-column tiles of B
 * (kl_by_n != 0) or the 256-row / 256-column tiles of A.  Everything the lists leave out must be zero records: the skipped
 * products would have added +-0.  Replaces nothing in the reference (its autograd contracts the dense gradient volume,
 * pytorch/core/corr.py:52-60); the lists come from fsraft_corr_bwd_ktiles. */
int fsraft_gemm_rec_nt_list(const void* A, int64_t lda, int64_t sA, const void* Bm, int64_t ldb, int64_t sB, float* C, int64_t ldc,
                            int64_t sC, int batch, int M, int N, int K, float alpha, int ksplit, int accumulate, const int* klist,
                            const int* kcount, int kl_stride, int kl_by_n, const unsigned* a_amax, const unsigned* b_amax,
                            hipStream_t stream);
int fsraft_gemm_rec_tn_list(const void* A, int64_t lda, int64_t sA, const void* Bm, int64_t ldb, int64_t sB, float* C, int64_t ldc,
                            int64_t sC, int batch, int M, int N, int K, float alpha, int ksplit, int accumulate, const int* klist,
                            const int* kcount, int kl_stride, int kl_by_n, const unsigned* a_amax, const unsigned* b_amax,
                            hipStream_t stream);
/* Which k-tiles of the volume-backward GEMMs the step's lookups can reach, from their coordinates alone (arguments as
 * fsraft_corr_dvol_build; n <= 16): nt_list [B][ceil(H*W / 128)][nt_stride >= P / 32] + nt_count = records per 128-query tile
 * (dF1 = s * F2cat . dV^T with dV as the B operand, kl_by_n = 1); tn_list [B][ceil(P / 256)][tn_stride >= ceil(H*W / 32)] +
 * tn_count = 32-query blocks per 256-cell tile (d2cat = s * dV^T . f1 with dV as the A operand); tn_bits: scratch of
 * B * ceil(H*W / 32) * ceil(ceil(P / 256) / 32) unsigned.  wmask (nullable): [B][ceil(H*W / 32)][ceil(P / 1024)] unsigned, bit r of
 * a 32-query block = record r of its rows is read by one of the two list GEMMs; handed to fsraft_corr_dvol_build, the gradient
 * volume is then written there and nowhere else.  P = the row length of fsraft_vol_layout.  nq > 0: one list set for the chunk
 * of queries [q0, q0 + nq) of one image (the chunked backward of AlternateCorrBlock): read B = 1 and H*W = nq above. */
int fsraft_corr_bwd_ktiles(const float* const* coords, const int64_t* coords_str, int n, int num_levels, int B, int H, int W,
                           int radius, int add_grid, int64_t q0, int64_t nq, int* nt_list, int* nt_count, int nt_stride, unsigned* tn_bits,
                           int* tn_list, int* tn_count, int tn_stride, unsigned* wmask, hipStream_t stream);

/* ---- GMA variant (config 5) --------------------------------------------------------------
 * Attention.forward, pytorch/core/gma.py:54-76: sim = scale * q k^T runs on fsraft_gemm_f32 (trans_b), then this
 * in-place row softmax over the last dimension (rows = B*heads*N, n = N; rows up to 16384 floats are staged in LDS). */
int fsraft_softmax_rows(float* S, int64_t rows, int n, hipStream_t stream);
/* dA <- A * (dA - rowsum(dA * A))  (softmax backward, in place over dA) */
int fsraft_softmax_rows_bwd(const float* A, float* dA, int64_t rows, int n, hipStream_t stream);
/* The same softmax (gma.py:71-74) with the probabilities written over the logits as RECORDS ([32 hi | 32 lo] fp16 per 32
 * columns: the operand form of fsraft_gemm_rec_nt / _tn): for n % 32 == 0 (n <= 16352) the record row is as long as the fp32
 * row, so the map exists once.  Probabilities are bounded by 1: their records are split with the fixed scale 2^13, the scale
 * of an amax word holding 1.0f -- the word to hand to the GEMMs that read them.  The backward reads those records (a = hi + lo)
 * and turns the fp32 gradient dA into the records of dS = A * (dA - rowsum(dA * A)) in place (n <= 8160), split with the scale
 * of ds_amax (|dS| <= 2 max |dA|). */
int fsraft_softmax_rows_rec(float* S, int64_t rows, int n, hipStream_t stream);
int fsraft_softmax_rows_bwd_rec(const void* A_records, float* dA, int64_t rows, int n, const unsigned* ds_amax, hipStream_t stream);
/* Aggregate.forward, gma.py:113: dst = x + gamma[0] * y with gamma a device scalar (the nn.Parameter). */
int fsraft_gma_mix_fwd(const float* x, int ldx, const float* y, int ldy, const float* gamma, float* dst, int ldd,
                       int64_t M, int C, unsigned* dst_amax /* nullable: raised */, hipStream_t stream);
/* d = dL/d dst:  dx += d;  dy = gamma * d;  dgamma[0] += sum(d * y) */
int fsraft_gma_mix_bwd(const float* d, int ldd, const float* y, int ldy, const float* gamma, float* dx, int ldx,
                       float* dy, int lddy, float* dgamma, int64_t M, int C, unsigned* dx_amax, unsigned* dy_amax /* nullable: raised */,
                       hipStream_t stream);

/* ---- normalisation + ReLU around the encoder convolutions (callers of the path) ------------------------------
 * pytorch/core/extractor.py:6-57: relu(norm(conv(x))) with norm = InstanceNorm2d (feature net) or a frozen
 * BatchNorm2d (context net).  The convolutions stay MIOpen; these fuse what surrounds them.  NCHW, one (n,c) plane
 * of HW floats per workgroup; stats[plane] = (mean, rstd). */
int fsraft_inorm_relu_fwd(const float* x, float* y, float* stats, int64_t planes, int HW, float eps, int relu, hipStream_t stream);
int fsraft_inorm_relu_bwd(const float* g, const float* x, const float* stats, float* dx, int64_t planes, int HW, int relu,
                          hipStream_t stream);
/* y = relu?(x * scale[c] + shift[c]);  backward: dx = g' * scale[c], dsum_g[c] += sum g', dsum_gx[c] += sum g' * x */
int fsraft_affine_relu_fwd(const float* x, const float* scale, const float* shift, float* y, int64_t planes, int C, int HW,
                           int relu, hipStream_t stream);
int fsraft_affine_relu_bwd(const float* g, const float* x, const float* scale, const float* shift, float* dx, float* dsum_g,
                           float* dsum_gx, int64_t planes, int C, int HW, int relu, hipStream_t stream);

/* ---- sequence loss (a step next to the path: pytorch/train.py:60-96) -----------------------------------------------
 * loss = sum_i weights[i] * mean(mask * sqrt((pred_i - gt)^2 + eps^2)), mask = valid >= 0.5 && |gt| < max_flow, over n
 * predictions [B,2,H,W] in one pass; dpred[i] (nullable) receives d loss / d pred_i; out[0] = loss, out[1..5] = EPE sum,
 * counts < 1 / 3 / 5 px and valid (> 0.5) count of prediction metric_idx.  gt / valid may be NULL (zero flow / all valid).
 * out[6] must be zero on entry. */
int fsraft_sequence_loss(const float* const* pred, float* const* dpred, const float* weights, int n, int metric_idx,
                         const float* gt, const float* valid, float max_flow, float eps, int B, int H, int W, float* out,
                         hipStream_t stream);

/* ---- warm start (a step next to the path: pytorch/core/utils/utils.py:26-54 forward_interpolate, called between the
 * frames of a sequence at pytorch/evaluate.py:43) -------------------------------------------------------------------
 * flow [2][H][W] (dx plane, dy plane) at the resolution the flow lives on: every vector is carried to (x + dx, y + dy),
 * landings outside the open rectangle (0, W) x (0, H) are dropped, and every grid node takes the vector of the nearest
 * landed point (float64 distances, as scipy's griddata(method="nearest")); all zero when nothing lands.  out != flow. */
int fsraft_forward_interpolate(const float* flow, float* out, int H, int W, hipStream_t stream);

/* Channels-last ([B][HW][C], C % 4 == 0, C <= 256) variants, for the encoder stages whose convolutions run on
 * fsraft_conv_forward.  sums/sumsq/s1/s2 and dsum_g/dsum_gx: [B * 8][C] partial rows (workgroups spread their atomics over
 * eight rows per sample: thousands of adds on the same C addresses serialise in L2 otherwise; dsum_*: the per-channel sums
 * are the column sums of the rows); all must be ZERO on entry.  stats: [B][C][2] = (mean, rstd).
 * Fused residual unit (pytorch/core/extractor.py:43-56, "return self.relu(x+y)"): res != NULL makes the forward write
 * y = relu(res + relu?(norm(x))); the backward then takes out = that y and writes the shortcut's gradient g * (out > 0)
 * to dres before continuing into the norm branch (out and dres both NULL: plain norm + ReLU). */
/* have_sums != 0: sums / sumsq already hold the partial rows -- accumulated by the convolution that produced x
 * (fsraft_conv_forward_stats, 8 slots) -- and the statistics pass over x is skipped. */
/* s2d_w != 0 (= the image width W; W and H = HW / W even): the result y is written in the space-to-depth layout of
 * fsraft_space_to_depth2 ([B][H/2][W/2][2][2][C]) -- what the stride-2 residual unit behind it reads (extractor.py:23-57 with
 * stride 2) -- and the backward reads the gradient g and the saved result `out` from that layout; x, res, dx, dres stay
 * [B][HW][C].  0: everything [B][HW][C].  (fsraft_affine_relu_cl_fwd takes HW for this; ignored when s2d_w == 0.) */
int fsraft_inorm_relu_cl_fwd(const float* x, const float* res, float* y, float* sums, float* sumsq, float* stats, int B, int HW,
                             int C, float eps, int relu, int have_sums, int s2d_w, unsigned* y_amax /* nullable: word of y, raised */,
                             hipStream_t stream);
int fsraft_inorm_relu_cl_bwd(const float* g, const float* x, const float* stats, const float* out, float* s1, float* s2, float* dx,
                             float* dres, int B, int HW, int C, int relu, int s2d_w, unsigned* dx_amax /* nullable: word of dx, raised */,
                             hipStream_t stream);
int fsraft_affine_relu_cl_fwd(const float* x, const float* res, const float* scale, const float* shift, float* y, int64_t M, int C,
                              int relu, int HW, int s2d_w, unsigned* y_amax, hipStream_t stream);
int fsraft_affine_relu_cl_bwd(const float* g, const float* x, const float* scale, const float* shift, const float* out, float* dx,
                              float* dres, float* dsum_g, float* dsum_gx, int B, int HW, int C, int relu, int s2d_w, unsigned* dx_amax,
                              hipStream_t stream);

/* Frozen-BatchNorm parameter folding in one launch per direction (instead of ~11 framework launches on [C] tensors per layer
 * and step): scale = weight * rsqrt(var + eps), shift = bias - (mean - cbias) * scale, rs = rsqrt(var + eps), rmc = mean - cbias
 * (cbias: bias of the preceding convolution, folded in; nullable).  Backward from the partial sums [2][R][C] that
 * fsraft_affine_relu_cl_bwd leaves: dweight = rs * (S1 - rmc * S0), dbias = S0, dcbias = scale * S0 (nullable). */
int fsraft_bn_fold(const float* weight, const float* bias, const float* rm, const float* rv, const float* cbias, float eps, int C,
                   float* scale, float* shift, float* rs, float* rmc, hipStream_t stream);
int fsraft_bn_fold_bwd(const float* part, int R, int C, const float* rs, const float* rmc, const float* scale, float* dweight,
                       float* dbias, float* dcbias, hipStream_t stream);

/* ---- the encoders' stem (next-row f3) ---------------------------------------------------
 * Replaces `self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3)` and its backward-weights
 * (pytorch/core/extractor.py:135 BasicEncoder, :212 SmallEncoder with 32 outputs).  x [B][3][H][W] planar fp32, w [N][3][7][7],
 * N = 32 or 64; out / dy channels-last [B][Ho][Wo][N] with Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1.  The image needs no
 * gradient (it is the network input).  fsraft_stem7x7s2_wgrad OVERWRITES dw [N][3][7][7]; scratch: fsraft_stem_slots() * 64 *
 * 192 floats of workspace (per-workgroup partial sums). */
int fsraft_stem_slots(void);
int fsraft_stem7x7s2_fwd(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int N,
                         const unsigned* x_amax /* word of the images; the weights' scale is worked out in the kernel */,
                         hipStream_t stream);
int fsraft_stem7x7s2_wgrad(const float* x, const float* dy, float* dw, float* scratch, int B, int H, int W, int N,
                           const unsigned* x_amax, const unsigned* dy_amax, hipStream_t stream);

/* ---- optimizer step of the train step (row e: what the DP step does after the all-reduce) ---------
 * `torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip)` + `optimizer.step()` of `optim.AdamW(model.parameters(),
 * lr, weight_decay, eps)` (pytorch/train.py:137, 280-282) on FLAT fp32 buffers: parameters p, gradients g (scaled in place
 * like clip_grad_norm_ does), first / second moments m, v, n elements each.  step: device fp32 step count, incremented here;
 * norm: device scalar holding the gradients' 2-norm (null: no clipping); lr: device scalar; state: 4 floats of scratch;
 * skip64 (nullable): ceil(n / 64) bytes, non-zero = the 64-element block belongs to a parameter that received no gradient
 * this step and is left untouched with its moments (torch's AdamW skips `p.grad is None`). */
int fsraft_adamw_flat(float* p, float* g, float* m, float* v, int64_t n, float* step, const float* norm, float max_norm,
                      const float* lr, float beta1, float beta2, float eps, float weight_decay, float* state,
                      const unsigned char* skip64, hipStream_t stream);
/* Host helper (no device work): *id = 0 when `stream` is not being captured into a hipGraph, else the capture's id.  The
 * binding keys per-capture scratch (zero-filled accumulation targets) on it. */
int fsraft_stream_capture_id(hipStream_t stream, unsigned long long* id);

/* ---- layout / elementwise helpers around the GEMMs ----------------------------------- */
int fsraft_nchw_to_nhwc(const float* src, float* dst, int B, int C, int HW, int ld, int coff, int accumulate, hipStream_t s);
/* Space-to-depth by 2 of a channels-last tensor: dst[b][y/2][x/2][(y%2)*2 + x%2][c] = src[b][y][x][c] (inverse != 0: back).
 * A stride-2 3x3 / 1x1 convolution over src (pytorch/core/extractor.py:13, 39: the first convolution and the shortcut of a
 * stride-2 ResidualBlock) is a stride-1 2x2 / 1x1 convolution of fsraft_conv_forward over dst viewed as [B][H/2][W/2][4C]. */
int fsraft_space_to_depth2(const float* src, float* dst, int B, int H, int W, int C, int inverse, hipStream_t s);
int fsraft_nhwc_to_nchw(const float* src, float* dst, int B, int C, int HW, int ld, int coff, int accumulate, hipStream_t s);
int fsraft_im2col7(const float* flow, int64_t bs, int64_t cs, int64_t ps, float* cols, int ld, int B, int H, int W,
                   unsigned* cols_amax /* nullable: raised */, hipStream_t s);
int fsraft_col2im7(const float* dcols, int ld, float* dflow, int B, int H, int W, int accumulate, hipStream_t s);
int fsraft_flow_to_nhwc(const float* flow, int64_t bs, int64_t cs, int64_t ps, float* dst, int ld, int coff, int B, int HW,
                        unsigned* dst_amax /* nullable: raised */, hipStream_t s);
int fsraft_nhwc_to_flow(const float* src, int ld, int coff, float* dflow, int B, int HW, int accumulate, hipStream_t s);
int fsraft_relu_bwd(float* g, int ldg, const float* y, int ldy, int64_t M, int C, hipStream_t s);
/* dzr_sum / dq_sum (nullable, same layouts as dzr / dq): running sums over the iterations of a step; dhn2 (nullable): a second
 * summand of the incoming gradient (dh' = dhn + dhn2: the heads' part and the part arriving from the next iteration) */
int fsraft_gru_bwd1(const float* dhn, const float* dhn2, const float* z, const float* q, const float* h, float* dzr, int ldzr, float* dq, float* dh, float* dzr_sum, float* dq_sum, int64_t M, int hid,
                    unsigned* dzr_amax, unsigned* dq_amax, unsigned* dh_amax /* nullable: words of the three outputs, raised */, hipStream_t s);
int fsraft_gru_bwd2(const float* drh, const float* r, const float* h, float* dzr, int ldzr, float* dh, float* dzr_sum, int64_t M, int hid,
                    unsigned* dzr_amax, unsigned* dh_amax, hipStream_t s);
int fsraft_col_sum(const float* x, int ld, int64_t M, int C, float* out, float scale, hipStream_t s);
int fsraft_axpby(const float* x, float* y, float a, float b, int64_t n, hipStream_t s);
/* out[0..count) = (accumulate ? out : 0) + src[0] + ... + src[n - 1], added in list order (count % 4 == 0, 16-byte aligned): the sum
 * of the GRU gate gradients over the iterations of a step (the context part's backward, update.py:16-60 with x = cat(inp, motion)
 * split into a per-step context convolution), formed once from the kept per-iteration gradients. */
int fsraft_sum_n(const float* const* src, int n, float* out, int64_t count, int accumulate, hipStream_t s);

#ifdef __cplusplus
}
#endif
#endif /* FSRAFT_H */
