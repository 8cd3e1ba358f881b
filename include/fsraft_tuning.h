/* fsraft_tuning.h -- measurement and test knobs of libfsraft.so.
 *
 * NOT part of the drop-in boundary (include/fsraft.h): nothing here corresponds to an interface of the reference, and a
 * maintainer integrating the library never calls these.  They exist so that tests can reach a kernel at sizes its
 * selection heuristic would not pick (e.g. the resident-patch 3x3 kernel at a 16x16 test grid), so that scripts/ can A/B
 * kernel variants inside one process, and so that the parity suite can run the GEMM families in either arithmetic mode
 * one by one (fsraft_set_arithmetic of fsraft.h switches them together).
 *
 * Kernels that lost their A/B twice (LDS-direct weight tiles, record-activation convolution, resident-weight 64 -> 64 kernel,
 * the transposed-role / persistent volume builds; keys 24 / 25 / 30 of earlier rounds) were removed from the tree in round 5: their
 * measurements are in docs/history/ and profiles/; the keys return FSRAFT_ERR_ARG.
 */
#ifndef FSRAFT_TUNING_H
#define FSRAFT_TUNING_H

#ifdef __cplusplus
extern "C" {
#endif

/* key 0: conv tile (0 auto, 1 128x128, 2 64x128, 3 64x64); 1: wgrad tile (0 128x128, 3 64x64); 2 / 11 / 17: target
 * workgroup counts of the weight-gradient pixel splits; 3: arithmetic of the forward / data-gradient convolutions (0 exact
 * fp32, 1 fp16x3; 3 / 4 / 5 force a tile shape); 4: arithmetic of the weight gradients (0 exact, 2 fp16x3); 5 / 8:
 * buffer-addressed loaders; 7: XCD swizzle; 9 / 13 / 14 / 15 / 18 / 19: tile-shape thresholds; 12: uniform k-tile table;
 * 16: few-channel weight-gradient kernel; 20 / 21: resident-patch 3x3 encoder kernel and its minimum pixel count; 22:
 * XCD-aware weight-gradient order; 26: resident-patch forward / data-gradient kernel; 27: resident-block weight gradient
 * (0 off, 1 the 3x3 layers, 2 the five-tap layers too); 28: 64-column patch tiles; 29: single-segment resident-block
 * weight gradient, minimum pixel count; 31: minimum pixel count of the resident-patch forward / data-gradient kernel; 32: split-K slices of the small-M convolutions
 * (-1 auto, 0 off, >= 2 forced). */
int fsraft_set_tuning(int key, int value);
int fsraft_get_tuning(int key);   /* keys 3 / 4 */
int fsraft_set_build_split(int on);   /* volume build: 1 fp16x3 (default), 0 exact fp32 MFMA */
int fsraft_set_build_kernel(int which); /* record build: bits 8..15 start-up stagger of odd workgroups (x 64 x 127 cycles), bits 16..18 store policy (0 auto, 1 plain, 2 sc1, 3 nt) */
/* cache policy of the tiled lookup's window loads: -1 auto (nt for volumes beyond the Infinity Cache; default), 0 plain, 2 nt,
 * 16 sc1, 18 nt + sc1 (A/B switch); 100 = measurement only: the nt window loads alone, no blends and no output (the gather floor of
 * the tiled layout, scripts/lookup_gather_floor.py) */
int fsraft_set_lookup_policy(int aux);
/* fsraft_corr_bwd_ktiles: 0 = per query and level the bounding rectangle of its lookups' windows is marked; 1 / 2 = on the
 * first one / two levels every lookup's window is marked on its own (tighter lists, a longer pre-pass). */
int fsraft_set_ktile_exact(int levels);
int fsraft_set_upsample_kernel(int v4);  /* convex upsampler: 1 (default) the 16-byte kernels for 16-byte aligned tensors, 0 the 4-byte ones */
int fsraft_set_dvol_box(int on);        /* gradient volume: 1 (default) one wave per query on its lookups' bounding boxes (corr_dvol_sep_kernel) + work list, 0 row-segment kernel only */
int fsraft_set_dvol_policy(int policy); /* cache policy of the gradient-volume stores: 0 plain, 1 sc1, 2 nt */
int fsraft_set_gemm_split(int on);    /* fsraft_gemm_f32 with trans_b: 1 fp16x3 when operands are 16-byte aligned */
int fsraft_set_lookup_qb(int qb);     /* queries per workgroup of the row-major lookup kernels: 0 auto, 8, 16 or 32 */
int fsraft_set_norm_blocks(int target_workgroups);   /* workgroups per launch of the channels-last norm kernels (default 4096) */
int fsraft_set_rec_mfma16(int on);    /* record GEMM (NT): 1 = v_mfma_f32_16x16x32_f16, 0 = 32x32x16 */
int fsraft_set_alt_tile(int on);      /* alt-corr forward: 1 (default) 4x4-query tile kernel, 0 wave per query */
int fsraft_set_alt_rough_pct(int pct); /* fsraft_altcorr_mfma_fwd with a regime buffer: the fp32 kernel takes the launch when more than pct per cent of the queries leave their tile's region (default 65; -1 never, i.e. always the matrix-pipe kernel) */

#ifdef __cplusplus
}
#endif
#endif
