"""Parity of the HIP path (through the C ABI in libfsraft.so) against the golden fixtures
generated from the reference and against the CPU oracle.  Needs an MI355X: -m gpu.

Tolerances: the path is fp32 end to end; kernels differ from the reference only in
summation order, so elementwise checks use 1e-4-level absolute tolerances on O(1..10)
values and the end-to-end gate is EPE <= 1e-3 (BASELINE.json), with ~1e-5 expected."""
import argparse
import math

import numpy as np
import pytest
import torch

from _util import T, close, decode_records, grad_digest_check, load, rel_check, shapes, word_scale
from oracle import raft_torch as O
from oracle.weights import procedural_state_dict, rand_tensor, rand_uniform, synthetic_pair

pytestmark = pytest.mark.gpu
DEV = "cuda"

# Limits of the train-step comparisons: loss rel, prediction abs [px], gradient-norm rel and gradient-head rel for everything
# except the feature encoder (gnorm / ghead), and the same pair for `fnet.*` (see grad_digest_check).  ONE table for both
# arithmetic modes (round 6: the split mode's fp16x3 products carry ~2^-22 each, the accuracy class of the exact mode's fp32
# MFMA -- rounds 1-5 split into bf16 pieces, 2^-17, and needed a second, looser table).  Set to ~4x the worst error measured on
# MI355X over the whole suite (profiles/r06_parity_margins.txt lists every comparison with the share of its limit it used);
# fnet: gnorm ~1e-3, ghead 1.9e-2 are the reference's own run-to-run noise on those gradients (round 2, docs/history).
_TOL = dict(loss=5e-6, pred=5e-4, gnorm=1.5e-3, ghead=1e-2, gnorm_fnet=5e-3, ghead_fnet=3e-2)
TRAIN_TOL = {"exact": _TOL, "split": _TOL}


@pytest.fixture(params=["exact", "split"])
def precision(request):
    """The update-block GEMMs have two arithmetic modes:
    exact  -- v_mfma_f32_32x32x2_f32, a pure fp32 fmaf chain (tolerances = fp32 summation-order noise);
    split  -- the default: every fp32 operand scaled by its tensor's power-of-two scale and split into fp16 hi + lo,
              three fp16 MFMAs per product with fp32 accumulation, relative error ~2^-22 per product (csrc/split_arith.hpp).
    Both modes are held to the SAME limits everywhere in this file."""
    from flow_supervisor_amd import ops as _ops
    _ops.set_arithmetic(request.param == "split")
    yield request.param
    _ops.set_arithmetic(True)


def _native():
    from flow_supervisor_amd import _lib
    _lib.load()


def ns(small):
    return argparse.Namespace(small=small, mixed_precision=False, alternate_corr=False, dropout=0,
                              corr_levels=4, corr_radius=3 if small else 4)


# ----------------------------------------------------------------------------- a1-a3
@pytest.mark.parametrize("name", ["corr_tiny", "corr_odd", "corr_mid"])
def test_corr_build_lookup_and_grads_vs_reference(name, precision):
    from flow_supervisor_amd.core.corr import CorrBlock
    g = load(name)
    B, C, H, W, r, seed = (int(g[k]) for k in ("B", "C", "H", "W", "radius", "seed"))
    f1 = rand_tensor((B, C, H, W), seed).to(DEV).requires_grad_(True)
    f2 = rand_tensor((B, C, H, W), seed + 1).to(DEV).requires_grad_(True)
    blk = CorrBlock(f1, f2, num_levels=4, radius=r)
    for l in range(4):
        close(blk.corr_pyramid[l], g[f"pyr{l}"], 2e-5, what=f"pyr{l}")
    coords = T(g["coords"]).to(DEV)
    out = blk(coords)
    assert out.is_contiguous() and out.shape == (B, 4 * (2 * r + 1) ** 2, H, W)
    close(out, g["out"], 5e-5, what="lookup")
    out_cl = blk(coords, channels_last=True)
    close(out_cl.permute(0, 3, 1, 2), g["out"], 5e-5, what="lookup channels-last")
    up = rand_tensor(tuple(out.shape), seed + 3).to(DEV)
    # two lookups feeding one loss: exercises the accumulate-in-place gradient pyramid
    (0.5 * (out * up).sum() + 0.5 * (out_cl.permute(0, 3, 1, 2) * up).sum()).backward()
    close(f1.grad, g["dfmap1"], 1e-4, what="dfmap1")
    close(f2.grad, g["dfmap2"], 1e-4, what="dfmap2")
    v = CorrBlock.corr(f1.detach(), f2.detach())
    close(v.reshape(-1), T(g["pyr0"]).reshape(-1), 2e-5, what="CorrBlock.corr")


def test_corr_lookup_matches_oracle_on_sintel_shape(precision):
    """Full 55x128 / C=256 shape against the oracle for a strip of queries (the oracle needs the
    whole volume, so B=1) plus an average-pool consistency property on every level."""
    from flow_supervisor_amd.core.corr import CorrBlock
    B, C, H, W, r = 1, 256, 55, 128, 4
    f1 = rand_tensor((B, C, H, W), 11)
    f2 = rand_tensor((B, C, H, W), 12)
    coords = O.coords_grid(B, H, W) + rand_uniform((B, 2, H, W), 13, -8, 8)
    pyr = O.corr_pyramid(f1, f2, 4)
    ref = O.corr_lookup(pyr, coords, r)
    blk = CorrBlock(f1.to(DEV), f2.to(DEV), radius=r)
    for l in range(4):
        close(blk.corr_pyramid[l], pyr[l], 5e-5, what=f"level {l}")
    close(blk(coords.to(DEV)), ref, 1e-4, what="lookup 55x128")


# ----------------------------------------------------------------------------- a4/a5
@pytest.mark.parametrize("name", ["corr_tiny", "corr_odd", "corr_mid"])
def test_alternate_corr_equals_corrblock(name):
    from flow_supervisor_amd.core.corr import AlternateCorrBlock
    g = load(name)
    B, C, H, W, r, seed = (int(g[k]) for k in ("B", "C", "H", "W", "radius", "seed"))
    f1 = rand_tensor((B, C, H, W), seed).to(DEV).requires_grad_(True)
    f2 = rand_tensor((B, C, H, W), seed + 1).to(DEV).requires_grad_(True)
    out = AlternateCorrBlock(f1, f2, num_levels=4, radius=r)(T(g["coords"]).to(DEV))
    close(out, g["out"], 1e-4, what="alt lookup")
    (out * rand_tensor(tuple(out.shape), seed + 3).to(DEV)).sum().backward()
    close(f1.grad, g["dfmap1"], 2e-4, what="alt dfmap1")
    close(f2.grad, g["dfmap2"], 2e-4, what="alt dfmap2")


def test_alt_cuda_corr_module_contract():
    import flow_supervisor_amd.alt_cuda_corr as acc
    f1 = torch.randn(1, 6, 8, 64, device=DEV)
    f2 = torch.randn(1, 6, 8, 64, device=DEV)
    co = torch.rand(1, 1, 6, 8, 2, device=DEV) * 5
    (corr,) = acc.forward(f1, f2, co, 4)
    assert corr.shape == (1, 1, 81, 6, 8)
    ref = O.alt_corr_level(f1.cpu(), f2.cpu(), co.cpu(), 4)
    close(corr, ref, 1e-4, what="alt_cuda_corr.forward")
    g1, g2, gc = acc.backward(f1, f2, co, torch.ones_like(corr), 4)
    assert g1.shape == f1.shape and g2.shape == f2.shape and gc.shape == co.shape and float(gc.abs().sum()) == 0
    with pytest.raises(RuntimeError):
        acc.forward(f1.cpu(), f2, co, 4)                       # CHECK_CUDA
    with pytest.raises(RuntimeError):
        acc.forward(f1.permute(0, 2, 1, 3), f2, co, 4)         # CHECK_CONTIGUOUS


@pytest.mark.parametrize("B,C,H,W,nlev", [(1, 256, 47, 156, 4), (2, 128, 17, 19, 4), (1, 64, 8, 12, 3), (1, 96, 5, 7, 2)])
def test_alt_lookup_on_the_matrix_pipe(B, C, H, W, nlev):
    """altcorr_mfma_fwd_kernel (tiles of 6 x 4 queries, the products with a region of target rows as a
    bf16x3 GEMM on records) against the oracle's lookup of the dense volume (what AlternateCorrBlock must equal,
    SURVEY.md 8c) and against the fp32 tile kernel: smooth flow, motion boundaries inside tiles (window positions outside a
    tile's region take the fp32 route), rough flow, flow that leaves the image; ragged tiles and 2..4 levels."""
    import torch.nn.functional as F
    from flow_supervisor_amd import ops
    torch.manual_seed(11)
    f1 = torch.randn(B, C, H, W, device=DEV)
    f2 = torch.randn(B, C, H, W, device=DEV)
    f1c = ops.nchw_to_nhwc(f1)
    lv, x = [], f2
    for _ in range(nlev):
        lv.append(ops.nchw_to_nhwc(x))
        x = F.avg_pool2d(x, 2, stride=2)
    recs = (ops.to_records(f1c.view(B, -1, C)), [ops.to_records(f.view(B, -1, C)) for f in lv])
    pyr = O.corr_pyramid(f1.cpu(), f2.cpu(), nlev)
    step = torch.zeros(B, 2, H, W, device=DEV)
    step[:, 0, :, W // 2:] = 11.0
    step[:, 1, H // 2:] -= 7.0
    cases = (("smooth", torch.tensor([3.3, -1.7], device=DEV).view(1, 2, 1, 1) + 0.3 * torch.randn(B, 2, H, W, device=DEV)),
             ("motion boundaries", step + 0.2 * torch.randn(B, 2, H, W, device=DEV)),
             ("rough", 8.0 * torch.randn(B, 2, H, W, device=DEV)),
             ("leaving the image", 200.0 * torch.randn(B, 2, H, W, device=DEV)))
    for name, flow in cases:
        got = ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True, recs=recs)
        ref = O.corr_lookup(pyr, (flow.cpu() + O.coords_grid(B, H, W)), 4)
        close(got.permute(0, 3, 1, 2), ref, 1e-4, what=f"matrix-pipe alt lookup vs oracle, {name}")
        fp32 = ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True)
        close(got, fp32, 1e-4, what=f"matrix-pipe alt lookup vs fp32 tile kernel, {name}")


@pytest.mark.parametrize("B,H,W,n,spread", [(2, 40, 48, 5, 2.0), (1, 37, 53, 12, 1.0), (1, 55, 128, 12, 3.0), (2, 24, 40, 3, 40.0)])
def test_volume_backward_over_listed_k_tiles_equals_the_dense_contraction(B, H, W, n, spread):
    """fsraft_corr_bwd_ktiles + fsraft_gemm_rec_nt_list / _tn_list against the dense record GEMMs on the same gradient volume:
    the lists must cover every non-zero record (checked on the raw records), dF1 -- one workgroup per tile walks its list in
    ascending order, the skipped k-tiles would have added +-0 -- is bit-equal, d2cat (two k-slices meeting in atomics, the slices
    cut differently) to summation-order noise.  Small flows, a ragged grid, the bench grid, and flows far beyond the image."""
    from flow_supervisor_amd import ops
    C, r = 64, 4
    lay = ops.VolLayout.get(H, W, 4)
    f1 = rand_tensor((B, C, H, W), 71, 1.0).to(DEV)
    f2 = rand_tensor((B, C, H, W), 72, 1.0).to(DEV)
    base = rand_tensor((B, 2, H, W), 73, spread).to(DEV)
    flows = [(base + rand_tensor((B, 2, H, W), 80 + t, 0.7).to(DEV)).contiguous() for t in range(n)]
    douts = [rand_tensor((B, H, W, 4 * (2 * r + 1) ** 2), 90 + t, 1.0).to(DEV) for t in range(n)]
    dvol = ops.corr_dvol_build(douts, flows, lay, B, r, records=True, is_flow=True)
    kt = ops.corr_bwd_ktiles(flows, lay, B, r, True)
    assert kt is not None
    # coverage: any record of a 128-query tile that holds a non-zero word must be listed
    N, P = H * W, lay.P
    nrec, ntiles = P // 32, -(-N // 128)
    nz = (dvol.view(torch.int32).view(B, N, nrec, 32) != 0).any(-1)                       # [B, N, nrec]
    pad = ntiles * 128 - N
    nzt = torch.nn.functional.pad(nz, (0, 0, 0, pad)).view(B, ntiles, 128, nrec).any(2).cpu()
    lists, counts = kt.nt_list.view(B, ntiles, kt.nt_stride).cpu(), kt.nt_count.view(B, ntiles).cpu()
    listed = torch.zeros(B, ntiles, nrec, dtype=torch.bool)
    for b in range(B):
        for t in range(ntiles):
            e = lists[b, t, :counts[b, t]].long()
            assert (e[1:] > e[:-1]).all(), "k-tile lists must ascend"
            listed[b, t, e] = True
    assert not (nzt & ~listed).any(), "a non-zero record is missing from the NT lists"
    # ... and every (256-cell tile, 32-query block) with a non-zero word from the TN lists
    mtiles, ktq = -(-P // 256), -(-N // 32)
    nzq = torch.nn.functional.pad(nz, (0, mtiles * 8 - nrec, 0, ktq * 32 - N)).view(B, ktq, 32, mtiles, 8).any(4).any(2).cpu()   # [B, ktq, mtiles]
    tl, tc = kt.tn_list.view(B, mtiles, kt.tn_stride).cpu(), kt.tn_count.view(B, mtiles).cpu()
    listed2 = torch.zeros(B, ktq, mtiles, dtype=torch.bool)
    for b in range(B):
        for m_ in range(mtiles):
            e = tl[b, m_, :tc[b, m_]].long()
            assert (e[1:] > e[:-1]).all()
            listed2[b, e, m_] = True
    assert not (nzq & ~listed2).any(), "a non-zero (query block, cell tile) pair is missing from the TN lists"
    f1r = ops.fmap_records(f1)
    a1, a2 = ops.corr_build_bwd_tiled(f1, f2, dvol, lay, records=True, f1r=f1r, ktiles=None)
    b1, b2 = ops.corr_build_bwd_tiled(f1, f2, dvol, lay, records=True, f1r=f1r, ktiles=kt)
    assert torch.equal(a1, b1)
    close(b2, a2, 1e-6, 1e-5, what="dfmap2 over listed k-tiles vs dense")
    # the gradient volume written only where the list GEMMs read (wmask), into a buffer poisoned with NaN bit patterns
    poison = torch.full((B * N, P), float("nan"), device=DEV)
    dv2 = ops.corr_dvol_build(douts, flows, lay, B, r, records=True, is_flow=True, wmask=kt.wmask, out=poison)
    c1, c2 = ops.corr_build_bwd_tiled(f1, f2, dv2, lay, records=True, f1r=f1r, ktiles=kt)
    assert torch.equal(c1, b1) and torch.equal(c2, b2)
    written = ~torch.isnan(dv2.view(B, N, nrec, 32)).all(-1)
    print(f"records written {written.float().mean().item():.2f}")
    frac = counts.sum().item() / (B * ntiles * nrec)
    print(f"k-tile fraction NT {frac:.2f}, TN {tc.sum().item() / (B * mtiles * ktq):.2f}")


def test_chunked_volume_backward_with_k_tile_lists_equals_without():
    """ops.corr_bwd_chunked (AlternateCorrBlock's backward: the gradient volume 2048 queries at a time) with the per-chunk k-tile
    lists and write masks (fsraft_corr_bwd_ktiles with q0 / nq; off by default: measured slower at the KITTI shape) against the
    dense chunks, on a grid whose last chunk is ragged."""
    from flow_supervisor_amd import ops
    B, C, H, W, r, n = 2, 64, 47, 61, 4, 4
    lay = ops.VolLayout.get(H, W, 4)
    f1 = rand_tensor((B, C, H, W), 171, 1.0).to(DEV)
    f2 = rand_tensor((B, C, H, W), 172, 1.0).to(DEV)
    flows = [(rand_tensor((B, 2, H, W), 173, 2.0) + rand_tensor((B, 2, H, W), 180 + t, 0.5)).to(DEV).contiguous() for t in range(n)]
    douts = [rand_tensor((B, H, W, 4 * (2 * r + 1) ** 2), 190 + t, 1.0).to(DEV) for t in range(n)]
    was = ops.CHUNK_KSKIP
    try:
        ops.CHUNK_KSKIP = False
        a1, a2 = ops.corr_bwd_chunked(f1, f2, douts, flows, lay, r, is_flow=True, chunk=1024)
        ops.CHUNK_KSKIP = True
        b1, b2 = ops.corr_bwd_chunked(f1, f2, douts, flows, lay, r, is_flow=True, chunk=1024)
    finally:
        ops.CHUNK_KSKIP = was
    close(b1, a1, 1e-6, 1e-5, what="dfmap1, chunked, with k-tile lists")          # (k-slices meet in atomics: summation order)
    close(b2, a2, 1e-6, 1e-5, what="dfmap2, chunked, with k-tile lists")


@pytest.mark.parametrize("B,H,W,nlev,n", [(2, 55, 128, 4, 12), (1, 17, 19, 4, 3), (2, 16, 24, 3, 16), (1, 46, 62, 4, 12)])
def test_gradient_volume_bounding_box_kernel_matches_the_row_kernel(B, H, W, nlev, n):
    """corr_dvol_sep_kernel (one wave per query, window gradients built separably in registers, only the bounding boxes of the
    lookups' windows in LDS, a work list for queries whose lookups spread further) against corr_dvol_kernel (the whole row segment
    in LDS), fp32 rows and records:
    smooth flows (every query on the fast route), independent noise, flows that jump +-40 px between lookups (every query
    through the work list), and a chunk of queries (AlternateCorrBlock's backward).  Same sums in the same order: equal up to
    the compilers' different fma contraction (<= 1e-6 relative to the row maximum); records within one unit of the low half."""
    from flow_supervisor_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(21)
    lay = ops.VolLayout.get(H, W, nlev)
    douts = [torch.randn(B, H, W, nlev * 81, device=DEV) for _ in range(n)]
    base = torch.randn(B, 2, H, W, device=DEV) * 3
    cases = (("smooth", [base + 0.3 * i for i in range(n)]),
             ("noise", [torch.randn(B, 2, H, W, device=DEV) * 3 for _ in range(n)]),
             ("jumping", [torch.randn(B, 2, H, W, device=DEV) * 40 for _ in range(n)]))
    try:
        for name, fl in cases:
            outs = {}
            for box in (0, 1):
                lib.fsraft_set_dvol_box(box)
                outs[box] = (ops.corr_dvol_build(douts, fl, lay, B, 4, records=False, is_flow=True),
                             ops.corr_dvol_build(douts, fl, lay, B, 4, records=True, is_flow=True))

            def dec(r):          # rows of [32 hi | 32 lo] fp16 records -> fp32 ((hi + lo) / scale of the word they carry)
                return decode_records(r, ops.amax_of(r))
            # (the wave-per-query kernel sums the lookups of one window origin in registers before they meet the others in the box: a
            #  different order of the same <= 4 n products per cell)
            close(outs[1][0], outs[0][0], 0.0, rtol=3e-6, what=f"gradient volume (fp32 rows), {name}")
            # a product with a zero weight is +-0 in either kernel; cells outside every window must be exactly zero
            assert int(((outs[1][0] != 0) & (outs[0][0] == 0)).sum()) == 0, "zero pattern"
            close(dec(outs[1][1]), dec(outs[0][1]), 0.0, rtol=3e-6, what=f"gradient volume (records, decoded: hi + lo carries 2^-22), {name}")
            close(dec(outs[1][1]), outs[1][0], 0.0, rtol=1e-6, what=f"records vs fp32 rows, {name}")
        nq0 = B * H * W
        q0, nq = nq0 // 3, min(100, nq0 - nq0 // 3)
        lib.fsraft_set_dvol_box(0)
        ref = ops.corr_dvol_build(douts, cases[1][1], lay, B, 4, records=False, is_flow=True, q0=q0, nq=nq)
        lib.fsraft_set_dvol_box(1)
        got = ops.corr_dvol_build(douts, cases[1][1], lay, B, 4, records=False, is_flow=True, q0=q0, nq=nq)
        close(got, ref, 0.0, rtol=3e-6, what="chunk of queries")
    finally:
        lib.fsraft_set_dvol_box(1)


@pytest.mark.parametrize("B,C,H,W,nlev", [(2, 32, 55, 128, 4), (1, 64, 47, 156, 4), (2, 32, 13, 17, 3), (1, 32, 6, 8, 2), (1, 32, 46, 62, 4)])
def test_pooled_target_operand_as_records_in_one_pass(B, C, H, W, nlev):
    """fsraft_corr_f2cat_rec (the plane pooled in LDS, records out) against the reference's own recursion (corr.py:24-26:
    avg_pool2d of the level above) laid out in the tiled rows, and against the two-kernel route it replaces: pad cells are
    zero records and the decoded records carry the value to 2^-16 (odd sizes: the plane is then not 16-byte aligned and takes
    the scalar loads).  The volume-backward tests above run through the one-pass route by default."""
    import torch.nn.functional as F
    from flow_supervisor_amd import ops
    torch.manual_seed(8)
    lay = ops.VolLayout.get(H, W, nlev)
    f2 = torch.randn(B, C, H, W, device=DEV)

    def dec(r):
        return decode_records(r.view(B * C, -1), ops.amax_of(r))
    old = ops.F2CAT_REC
    try:
        ops.F2CAT_REC = True
        one = dec(ops.f2cat_records(f2, lay))
        ops.F2CAT_REC = False
        two = dec(ops.f2cat_records(f2, lay))
    finally:
        ops.F2CAT_REC = old
    ref = torch.zeros(B * C, lay.P, device=DEV)
    lv = f2.reshape(B * C, 1, H, W)
    for l in range(nlev):
        if l:
            lv = F.avg_pool2d(lv, 2, stride=2)
        h, w = lv.shape[-2:]
        t = torch.zeros(B * C, lay.th[l] * 4, lay.tw[l] * 4, device=DEV)
        t[:, :h, :w] = lv[:, 0]
        t = t.view(B * C, lay.th[l], 4, lay.tw[l], 4).permute(0, 1, 3, 2, 4).reshape(B * C, -1)
        ref[:, lay.off[l]:lay.off[l] + t.shape[1]] = t
    close(one, ref, 1e-6, rtol=1e-6, what="one-pass records vs recursive avg_pool2d")
    close(two, ref, 1e-6, rtol=1e-6, what="two-kernel records vs recursive avg_pool2d")
    assert ((one == 0) == (ref == 0)).all(), "pad cells are zero records"


def test_alt_cuda_corr_several_coordinate_sets():
    """coords [B,N,H1,W1,2] with N > 1 (correlation_kernel.cu:34,59; the C ABI carries N): every set against the oracle's
    restatement of one extension call, and the backward against autograd of that restatement."""
    import flow_supervisor_amd.alt_cuda_corr as acc
    torch.manual_seed(5)
    B, N, H, W, C, r = 2, 3, 7, 9, 128, 4
    f1 = torch.randn(B, H, W, C, device=DEV)
    f2 = torch.randn(B, H, W, C, device=DEV)
    co = torch.rand(B, N, H, W, 2, device=DEV) * 12 - 2          # some windows leave the map
    (corr,) = acc.forward(f1, f2, co, r)
    assert corr.shape == (B, N, 81, H, W)
    dout = torch.randn_like(corr)
    g1, g2, gc = acc.backward(f1, f2, co, dout, r)
    assert gc.shape == co.shape and float(gc.abs().sum()) == 0
    f1c, f2c = f1.cpu().requires_grad_(True), f2.cpu().requires_grad_(True)
    ref = torch.cat([O.alt_corr_level(f1c, f2c, co.cpu()[:, n:n + 1], r) for n in range(N)], 1)
    close(corr, ref.detach(), 1e-4, what="alt_cuda_corr.forward N=3")
    (ref * dout.cpu()).sum().backward()
    close(g1, f1c.grad, 2e-4, what="alt_cuda_corr.backward N=3 fmap1_grad")
    close(g2, f2c.grad, 2e-4, what="alt_cuda_corr.backward N=3 fmap2_grad")


# ----------------------------------------------------------------------------- a9
def test_convex_upsample_vs_reference():
    from flow_supervisor_amd.core.raft import RAFT
    g = load("upsample")
    N, H, W = int(g["N"]), int(g["H"]), int(g["W"])
    flow = rand_tensor((N, 2, H, W), 401, 2.0).to(DEV).requires_grad_(True)
    mask = rand_tensor((N, 576, H, W), 402, 1.5).to(DEV).requires_grad_(True)
    model = RAFT(ns(False)).to(DEV)
    up = model.upsample_flow(flow, mask)
    close(up, g["up"], 1e-5, what="up")
    (up * rand_tensor(tuple(up.shape), 403).to(DEV)).sum().backward()
    close(flow.grad, g["dflow"], 1e-5, what="dflow")
    close(mask.grad, g["dmask"], 1e-5, what="dmask")


@pytest.mark.parametrize("N,H,W", [(1, 5, 7), (2, 6, 16), (1, 9, 37), (3, 4, 1)])
def test_convex_upsample_ragged_widths_both_kernels_vs_oracle(N, H, W):
    """raft.py:72-83 at widths that are not multiples of a workgroup's 16 (or 8) pixels, on the 16-byte kernels (default) and
    the 4-byte ones (fsraft_set_upsample_kernel(0)): forward and both gradients against the oracle's restatement."""
    from flow_supervisor_amd import _lib
    from flow_supervisor_amd.core.raft import RAFT
    lib = _lib.load()
    model = RAFT(ns(False)).to(DEV)
    fc = rand_tensor((N, 2, H, W), 421, 2.0); mc = rand_tensor((N, 576, H, W), 422, 1.5)
    fr, mr = fc.clone().requires_grad_(True), mc.clone().requires_grad_(True)
    ur = O.upsample_flow(fr, mr)
    w = rand_tensor(tuple(ur.shape), 423)
    (ur * w).sum().backward()
    try:
        for v4 in (1, 0):
            assert lib.fsraft_set_upsample_kernel(v4) == 0
            f = fc.to(DEV).requires_grad_(True); m = mc.to(DEV).requires_grad_(True)
            up = model.upsample_flow(f, m)
            close(up, ur, 1e-5, what=f"up (v4={v4})")
            (up * w.to(DEV)).sum().backward()
            close(f.grad, fr.grad, 1e-5, what=f"dflow (v4={v4})")
            close(m.grad, mr.grad, 1e-5, what=f"dmask (v4={v4})")
    finally:
        lib.fsraft_set_upsample_kernel(1)


def test_upflow8_and_helpers():
    from flow_supervisor_amd.core.utils.utils import InputPadder, coords_grid, upflow8
    h = load("helpers")
    f = rand_tensor((2, 2, 5, 7), 411, 2.0).to(DEV).requires_grad_(True)
    u = upflow8(f)
    close(u, h["upflow8"], 1e-5, what="upflow8")
    gup = rand_tensor(tuple(u.shape), 412).to(DEV)
    (u * gup).sum().backward()
    fr = f.detach().cpu().requires_grad_(True)
    (O.upflow8(fr) * gup.cpu()).sum().backward()
    close(f.grad, fr.grad, 1e-4, what="upflow8 grad")
    close(coords_grid(2, 3, 5, device=DEV), h["coords_grid"], 0)
    for k, v in h.items():
        if k.startswith("pad_"):
            _, mode, ht, wd = k.split("_")
            assert InputPadder((1, 3, int(ht), int(wd)), mode=mode)._pad == list(v)


# ----------------------------------------------------------------------------- a6-a8
@pytest.mark.parametrize("tag", ["basic", "small"])
def test_update_block_vs_reference(tag, precision):
    f = 1.0          # (one set of limits for both arithmetic modes)
    from flow_supervisor_amd.core.update import BasicUpdateBlock, SmallUpdateBlock
    g = load("update_" + tag)
    small = tag == "small"
    seed = int(g["seed"])
    blk = (SmallUpdateBlock(ns(True), hidden_dim=96) if small else BasicUpdateBlock(ns(False), hidden_dim=128))
    sh = shapes("update_" + tag)
    assert {k: list(v.shape) for k, v in blk.state_dict().items()} == sh
    blk.load_state_dict(procedural_state_dict(sh, seed))
    blk = blk.to(DEV)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    hd, cd, r = (96, 64, 3) if small else (128, 128, 4)
    cp = 4 * (2 * r + 1) ** 2
    net = torch.tanh(rand_tensor((B, hd, H, W), seed + 10)).to(DEV).requires_grad_(True)
    inp = torch.relu(rand_tensor((B, cd, H, W), seed + 11)).to(DEV).requires_grad_(True)
    corr = rand_tensor((B, cp, H, W), seed + 12, 2.0).to(DEV).requires_grad_(True)
    flow = rand_tensor((B, 2, H, W), seed + 13, 3.0).to(DEV).requires_grad_(True)
    net2, mask, delta = blk(net, inp, corr, flow)
    close(net2, g["net_out"], 2e-5 * f, what="net")
    close(delta, g["delta"], 2e-5 * f, what="delta")
    loss = (net2 * rand_tensor(tuple(net2.shape), seed + 20).to(DEV)).sum() + (delta * rand_tensor(tuple(delta.shape), seed + 21).to(DEV)).sum()
    if small:
        assert mask is None
    else:
        close(mask, g["mask"], 2e-5 * f, what="mask")
        loss = loss + (mask * rand_tensor(tuple(mask.shape), seed + 22).to(DEV)).sum()
    loss.backward()
    close(net.grad, g["dnet"], 2e-4 * f, what="dnet")
    close(inp.grad, g["dinp"], 2e-4 * f, what="dinp")
    close(corr.grad, g["dcorr"], 2e-4 * f, what="dcorr")
    close(flow.grad, g["dflow"], 2e-4 * f, what="dflow")
    for k, p in blk.named_parameters():
        gr = p.grad.reshape(-1)
        ref_n = float(g["dparam_norm." + k])
        assert abs(gr.norm().item() - ref_n) <= 2e-4 * f * ref_n + 1e-5, (k, gr.norm().item(), ref_n)
        samp = gr if gr.numel() <= 4096 else gr[:: gr.numel() // 4096][:4096]
        close(samp, g["dparam." + k], 2e-4, 1e-3, what="d" + k)


# ----------------------------------------------------------------------------- end to end
def _model(small, seed):
    from flow_supervisor_amd.core.raft import RAFT
    m = RAFT(ns(small))
    m.load_state_dict(procedural_state_dict(shapes("raft_small" if small else "raft_basic"), seed))
    return m.to(DEV)


def test_mixed_precision_flag_is_accepted_and_has_no_effect():
    """args.mixed_precision (raft.py:99-127): accepted, warned about once, and without effect -- the models do not enter autocast
    (it would send the encoders to the framework's half-precision convolutions) and the path stores fp32 either way."""
    import warnings
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.core.utils import utils as U
    sd = procedural_state_dict(shapes("raft_basic"), 77)
    im1, im2 = (t.to(DEV) for t in synthetic_pair(1, 128, 192, 78))
    outs = []
    for mp in (False, True):
        U._MIXED_WARNED[0] = False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            a = ns(False)
            a.mixed_precision = mp
            m = RAFT(a)
        assert any("mixed_precision" in str(x.message) for x in w) == mp
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        with torch.no_grad():
            outs.append(m(im1, im2, iters=4, test_mode=True)[1])
    # (the same kernels twice: atomics' summation order is the only difference; under autocast the encoders moved the flow by 3e-3)
    assert (outs[0] - outs[1]).abs().max().item() <= 1e-4


@pytest.mark.parametrize("name", ["e2e_small_128x256", "e2e_basic_368x496", "e2e_basic_440x1024"])
def test_end_to_end_flow_epe(name, precision):
    g = load(name)
    small, seed = bool(g["small"]), int(g["seed"])
    m = _model(small, seed).eval()
    im1, im2 = synthetic_pair(int(g["B"]), int(g["H"]), int(g["W"]), seed + 1)
    with torch.no_grad():
        low, up = m(im1.to(DEV), im2.to(DEV), iters=int(g["iters"]), test_mode=True)
    s = int(g["stride"])
    e_low = O.epe(low.cpu(), T(g["flow_low"])).item()
    e_up = O.epe(up[:, :, ::s, ::s].cpu(), T(g["flow_up_strided"])).item()
    print(name, precision, "EPE low", e_low, "EPE up", e_up)
    assert e_low <= 1e-3 and e_up <= 1e-3, (e_low, e_up)      # BASELINE.json gate
    assert e_up <= 5e-5, (precision, e_up)   # what both modes deliver


def test_end_to_end_alternate_corr_epe():
    g = load("e2e_basic_368x496")
    seed = int(g["seed"])
    m = _model(False, seed).eval()
    m.args.alternate_corr = True
    im1, im2 = synthetic_pair(1, 368, 496, seed + 1)
    with torch.no_grad():
        low, up = m(im1.to(DEV), im2.to(DEV), iters=12, test_mode=True)
    e = O.epe(up[:, :, ::4, ::4].cpu(), T(g["flow_up_strided"])).item()
    print("alt-corr EPE", e)
    assert e <= 1e-3


@pytest.mark.parametrize("alternate", [False, True])
def test_kitti_shape_evaluation(alternate):
    """evaluate.py:133-148: 375x1242 frames, InputPadder(mode='kitti') -> 376x1248 (47x156 features), 24 iters."""
    from flow_supervisor_amd.core.utils.utils import InputPadder
    g = load("e2e_basic_kitti_375x1242")
    seed, s = int(g["seed"]), int(g["stride"])
    m = _model(False, seed).eval()
    m.args.alternate_corr = alternate
    im1, im2 = synthetic_pair(1, 375, 1242, seed + 1)
    padder = InputPadder(im1.shape, mode="kitti")
    p1, p2 = padder.pad(im1.to(DEV), im2.to(DEV))
    assert tuple(p1.shape[-2:]) == tuple(int(v) for v in g["padded"])
    with torch.no_grad():
        low, up = m(p1, p2, iters=int(g["iters"]), test_mode=True)
    flow = padder.unpad(up)
    assert tuple(flow.shape) == (1, 2, 375, 1242)
    e_low = O.epe(low.cpu(), T(g["flow_low"])).item()
    e = O.epe(flow[:, :, ::s, ::s].cpu(), T(g["flow_strided"])).item()
    print("kitti", "alt" if alternate else "volume", "EPE low", e_low, "EPE", e)
    assert e_low <= 1e-3 and e <= 1e-3


@pytest.mark.parametrize("one_stream", [False, True])
@pytest.mark.parametrize("tag", ["basic", "small"])
def test_train_step_loss_and_grads(tag, precision, one_stream, monkeypatch):
    """(one_stream: core/streams.py switched off -- every branch of the forward pass on the caller's stream, the route every
    other golden test of this file runs with the switch on)"""
    from flow_supervisor_amd.core import streams
    monkeypatch.setattr(streams, "OVERLAP", not one_stream)
    g = load("train_step_" + tag)
    small, seed = tag == "small", int(g["seed"])
    m = _model(small, seed).train()
    m.freeze_bn()
    im1, im2 = synthetic_pair(2, int(g["H"]), int(g["W"]), seed + 1)
    preds = m(im1.to(DEV), im2.to(DEV), iters=int(g["iters"]))
    tol = TRAIN_TOL[precision]
    loss = O.sequence_loss_zero_gt(preds)
    rel_check(loss.item(), g["loss"], tol["loss"], "loss")
    loss.backward()
    close(preds[-1], g["last"], tol["pred"], rtol=0.0, what="last prediction")
    bad = grad_digest_check(m.named_parameters(), g, tol)
    assert not bad, bad[:8]


@pytest.mark.parametrize("tag", ["basic", "small"])
def test_gate_gradient_sums_of_a_step_in_one_pass(tag):
    """The context part of the GRU convolutions runs once per step (x = cat(inp, motion) of update.py:16-60 split), so its
    backward needs the gate gradients summed over the iterations.  Deferred (default): gru_bwd1 / gru_bwd2 only write the
    iteration's gradients and fsraft_sum_n adds the kept buffers once, in the order the running sums were formed: three list
    lengths (1, 3, 17 > one launch) of the kernel equal the sequential sum bit for bit, and the context encoder's gradients
    (everything behind `inp`) of a whole step agree with the running-sum route to the run-to-run spread of either."""
    from flow_supervisor_amd import ops
    from flow_supervisor_amd.core import update as U
    torch.manual_seed(3)
    for n in (1, 3, 17):
        ts = [torch.randn(2, 5, 7, 8, device=DEV) for _ in range(n)]
        ref = torch.zeros_like(ts[0])
        for t in ts:
            ref = ref + t
        out = torch.full((3, 5, 7, 8), 7.0, device=DEV)
        ops.sum_n_(ts, out)
        assert torch.equal(out[:2], ref) and float(out[2].min()) == 7.0
        ops.sum_n_(ts[:1], out, accumulate=True)
        assert torch.equal(out[:2], ref + ts[0])
    g = load("train_step_" + tag)
    small, seed = tag == "small", int(g["seed"])
    im1, im2 = (t.to(DEV) for t in synthetic_pair(2, int(g["H"]), int(g["W"]), seed + 1))
    res = {}
    old = U.CTX_SUM_DEFERRED
    try:
        for flag in (True, False):
            U.CTX_SUM_DEFERRED = flag
            m = _model(small, seed).train()
            m.freeze_bn()
            O.sequence_loss_zero_gt(m(im1, im2, iters=int(g["iters"]))).backward()
            res[flag] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    finally:
        U.CTX_SUM_DEFERRED = old
    assert res[True].keys() == res[False].keys()
    cn = [k for k in res[True] if k.startswith("cnet.")]
    assert cn
    for k in cn:
        # (two separate backward passes: the context encoder's norm sums and weight gradients are atomically accumulated, and
        #  fifteen layers amplify the order of those adds -- the same spread two runs of ONE route show, up to a few 1e-3 in
        #  relative L2 for the bottleneck encoder; what is exact is the sum kernel above)
        e = _rel_l2(res[True][k], res[False][k])
        assert e < 2e-2 or res[False][k].norm().item() < 1e-3, f"{k}: deferred vs running sums, relative L2 error {e:.3e}"


def _check_train_digest(m, preds, g, precision, skip=()):
    """loss, first / last prediction (strided) and every parameter-gradient norm + head against a `_train_digest` fixture."""
    tol = TRAIN_TOL[precision]
    loss = O.sequence_loss_zero_gt(preds)
    rel_check(loss.item(), g["loss"], tol["loss"], "loss")
    loss.backward()
    s = int(g["stride"])
    close(preds[0][:, :, ::s, ::s], g["first"], tol["pred"], rtol=0.0, what="first prediction")
    close(preds[-1][:, :, ::s, ::s], g["last"], tol["pred"], rtol=0.0, what="last prediction")
    bad = grad_digest_check(m.named_parameters(), g, tol, skip=skip)
    assert not bad, bad[:8]


@pytest.mark.parametrize("name,alternate", [("train_step_basic_440x1024", False), ("train_step_basic_376x1248", False),
                                            ("train_step_basic_376x1248", True)])
def test_train_step_at_bench_scale(name, alternate, precision):
    """One pair at the benchmark's own shapes, 12 iterations, forward + backward against the reference:
    the 12-segment batched weight gradient, the 12-deep operand stash and the once-per-step context backward run exactly as
    in bench.py.  376x1248 is config 4's padded KITTI shape; with alternate=True the same fixture (CorrBlock is the
    alt path's oracle) checks AlternateCorrBlock's wired backward at full size."""
    if alternate and precision == "exact":
        pytest.skip("the alt-corr kernels have one arithmetic mode; covered by the split run")
    g = load(name)
    seed = int(g["seed"])
    m = _model(False, seed).train()
    m.freeze_bn()
    m.args.alternate_corr = alternate
    im1, im2 = synthetic_pair(1, int(g["H"]), int(g["W"]), seed + 1)
    preds = m(im1.to(DEV), im2.to(DEV), iters=int(g["iters"]))
    assert len(preds) == 12
    _check_train_digest(m, preds, g, precision)


@pytest.mark.parametrize("one_stream", [False, True])
def test_l2l_two_phase_forward_and_grads(precision, one_stream, monkeypatch):
    """Flow-supervisor forward (core/l2l.py:29-133): student half on the crop, supervisor half on the uncropped
    pair with zero-padded detached state and a second correlation volume; golden from the reference L2L.
    one_stream: core/streams.py off (the uncropped frames are then encoded at the switch iteration, as the reference does)."""
    from flow_supervisor_amd.core import streams
    from flow_supervisor_amd.core.l2l import L2L
    monkeypatch.setattr(streams, "OVERLAP", not one_stream)
    g = load("l2l_basic")
    seed, B, iters = int(g["seed"]), int(g["B"]), int(g["iters"])
    H, W, h, w, oy, ox = (int(g[k]) for k in ("H", "W", "h", "w", "oy", "ox"))
    m = L2L(ns(False))
    m.load_state_dict(procedural_state_dict(shapes("l2l_basic"), seed))
    m = m.to(DEV).train()
    m.freeze_bn()
    ci1, ci2 = (t.to(DEV) for t in synthetic_pair(B, H, W, seed + 1))
    im1 = ci1[:, :, oy:oy + h, ox:ox + w].contiguous()
    im2 = ci2[:, :, oy:oy + h, ox:ox + w].contiguous()
    with pytest.raises(NameError):
        m(im1, im2, iters=iters)
    preds = m(im1, im2, ci1, ci2, torch.tensor([ox] * B), torch.tensor([oy] * B), iters=iters)
    assert len(preds) == iters and all(tuple(p.shape) == (B, 2, h, w) for p in preds)
    tol = TRAIN_TOL[precision]
    loss = O.sequence_loss_zero_gt(preds)
    rel_check(loss.item(), g["loss"], tol["loss"], "loss")
    loss.backward()
    close(preds[iters // 2 - 1][:, :, ::2, ::2], g["mid"], tol["pred"], rtol=0.0, what="last student prediction")
    close(preds[-1][:, :, ::2, ::2], g["last"], tol["pred"], rtol=0.0, what="last supervisor prediction")
    bad = grad_digest_check(m.named_parameters(), g, tol, hprefix=None)
    assert not bad, bad[:8]
    m.eval()
    with torch.no_grad():
        low, up = m(im1, im2, iters=iters, test_mode=True)
    assert O.epe(low.cpu(), T(g["test_low"])).item() <= 1e-3
    assert O.epe(up[:, :, ::2, ::2].cpu(), T(g["test_up"])).item() <= 1e-3



def _recipe_sample(g, tag, seed):
    """Inputs of tests/golden/make_golden.py::l2l_recipe_inputs, regenerated."""
    H, W, h, w = (int(g[k]) for k in ("H", "W", "h", "w"))
    sd = seed + (1 if tag == "sup" else 5)
    oy, ox = int(g[tag + "_oy"]), int(g[tag + "_ox"])
    ci1, ci2 = synthetic_pair(1, H, W, sd)
    im1 = (ci1[:, :, oy:oy + h, ox:ox + w] + rand_tensor((1, 3, h, w), sd + 1, 3.0)).clamp(0, 255).contiguous()
    im2 = (ci2[:, :, oy:oy + h, ox:ox + w] + rand_tensor((1, 3, h, w), sd + 2, 3.0)).clamp(0, 255).contiguous()
    flow = rand_tensor((1, 2, h, w), sd + 3, 4.0)
    valid = (rand_uniform((1, h, w), sd + 4, 0.0, 1.0) > 0.1).float()
    return tuple(t.to(DEV) for t in (im1, im2, ci1, ci2)) + (ox, oy, flow.to(DEV), valid.to(DEV))


def _recipe_model(tag):
    """L2L ("basic": Sintel recipe, "kitti": KITTI recipe -- the same network) or GMAL2L with the fixture's procedural weights."""
    g = load("l2l_recipe_" + tag)
    seed = int(g["seed"])
    if tag == "gma":
        from flow_supervisor_amd.core.gma_l2l import GMAL2L
        m = GMAL2L(gma_ns())
    else:
        from flow_supervisor_amd.core.l2l import L2L
        m = L2L(ns(False))
    sd = procedural_state_dict(shapes("l2l_recipe_" + ("gma" if tag == "gma" else "basic")), seed)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("rel_ind" in k for k in missing), (missing, unexpected)
    if tag == "gma":
        with torch.no_grad():
            m.update_block.aggregator.gamma.fill_(0.1)
    m = m.to(DEV).train()
    m.freeze_bn()
    return g, seed, m


@pytest.mark.parametrize("tag", ["basic", "gma", "kitti"])
def test_flow_supervisor_step_at_the_reference_recipe(tag, precision):
    """the optimisation step the reference repo exists for, at its own operating point
    (train_semi.sh:3-11: batch 1, crop 368x768 inside the 432x1024 frame, 12 student + 12 supervisor iterations), against
    the reference's L2L / GMAL2L run on the same inputs (tests/golden/l2l_recipe_*.npz): labelled pass with sequence_loss,
    unlabelled pass with sequence_loss_unsup, gradients of both passes accumulated through FlatGradients' two-pass
    buckets exactly as train.SemiTrainStep does.  Checked: both losses, student / supervisor predictions at iterations
    0, 11, 12, 23, every parameter-gradient norm and head after the labelled pass and after both (update_block AND
    grad_update_block; for GMAL2L the second half stays on update_block and grad_update_block gets none)."""
    from flow_supervisor_amd.parallel import FlatGradients
    from flow_supervisor_amd.train import sequence_loss, sequence_loss_unsup
    g, seed, m = _recipe_model(tag)
    hw = (int(g["h"]), int(g["w"]))                 # "kitti": train_semi.sh:14-17, crop 288x960 inside the 368x1240 frame
    tol = TRAIN_TOL[precision]
    named = list(m.named_parameters())
    grads = FlatGradients([p for _, p in named], [n for n, _ in named])
    grads.begin(backward_passes=2)
    skip = ("pos_emb",)
    for which in ("sup", "unsup"):
        im1, im2, ci1, ci2, ox, oy, flow, valid = _recipe_sample(g, which, seed)
        preds = m(im1, im2, ci1, ci2, ox, oy, iters=24, supervisor_grad=which == "sup")
        assert len(preds) == 24 and all(tuple(p.shape) == (1, 2) + hw for p in preds)
        if which == "sup":
            loss, metrics = sequence_loss(preds, flow, valid, float(g["gamma"]))
        else:
            loss, metrics = sequence_loss_unsup(preds, flow, valid, unsup_weight=float(g["unsup_lambda"]))
        rel_check(loss.item(), g[which + "_loss"], tol["loss"], which + " loss")
        rel_check(metrics["epe"], g[which + "_epe"], 1e-4, which + " epe metric")
        for i in (0, 11, 12, 23):
            close(preds[i][:, :, ::4, ::4], g[f"{which}_pred{i}"], tol["pred"], rtol=0.0, what=f"{which} prediction {i}")
        loss.backward()
        del preds
        if which == "sup":
            # gradients of the first pass alone (autograd's own tensors at this point: the buckets wait for the second pass)
            bad = grad_digest_check(named, g, tol, prefix="gnorm_sup.", hprefix="ghead_sup.", skip=skip)
            assert not bad, ("after the labelled pass", bad[:8])
    grads.finish()
    bad = grad_digest_check(named, g, tol, skip=skip)
    assert not bad, ("after both passes", bad[:8])
    if tag == "gma":
        miss = {id(p) for p in grads.missing}
        assert all(id(p) in miss for n, p in named if n.startswith("grad_update_block."))


@pytest.mark.parametrize("tag,batched", [("basic", True), ("basic", False), ("gma", True), ("kitti", True)])
def test_semi_train_step_gradients_at_the_reference_recipe(tag, batched, precision):
    """train.SemiTrainStep itself (the object bench.py --variant l2l / gma_l2l times) against the reference's two-pass step
    (tests/golden/l2l_recipe_*.npz): batched = the labelled and the unlabelled sample as one batch of two with per-sample crop
    offsets and ONE backward -- with everything the batch enables: the unlabelled sample's uncropped frames encoded without a
    graph, the supervisor phase's backward (update block and second volume) run on the labelled sample alone, the mask
    heads / upsamplers / losses of both samples in single launches; sequential = the reference's order, two forward /
    backward passes into the two-pass gradient buckets.  Both must reproduce the reference's losses and accumulated
    parameter gradients."""
    from flow_supervisor_amd.train import SemiTrainStep
    g, seed, m = _recipe_model(tag)
    step = SemiTrainStep(m, lr=0.0, wdecay=0.0, clip=None, iters=12, gamma=float(g["gamma"]), unsup_lambda=float(g["unsup_lambda"]),
                         batched=batched)
    sup, unsup = _recipe_sample(g, "sup", seed), _recipe_sample(g, "unsup", seed)
    ls, lu = step(sup, unsup)
    tol = TRAIN_TOL[precision]
    rel_check(float(ls), g["sup_loss"], tol["loss"], "sup loss")
    rel_check(float(lu), g["unsup_loss"], tol["loss"], "unsup loss")
    bad = grad_digest_check(list(m.named_parameters()), g, tol, skip=("pos_emb",))
    assert not bad, bad[:8]


@pytest.mark.parametrize("fixture", ["train_step_basic_368x496_b8", "train_step_basic_368x496_b8_it12"])
def test_chairs_batch8_train_step(precision, fixture):
    """BASELINE.json config 2 at its own batch size (8 pairs, 368x496), fwd + bwd: 3 iterations, and the
    configuration's own 12."""
    g = load(fixture)
    seed = int(g["seed"])
    m = _model(False, seed).train()
    m.freeze_bn()
    im1, im2 = synthetic_pair(int(g["B"]), int(g["H"]), int(g["W"]), seed + 1)
    preds = m(im1.to(DEV), im2.to(DEV), iters=int(g["iters"]))
    _check_train_digest(m, preds, g, precision)


def test_sequence_loss_unsup_vs_reference_function():
    """train.sequence_loss_unsup (the fused loss kernel with the supervisor's last prediction as target) against outputs of
    the reference's sequence_loss_unsup (pytorch/train.py:99-129): loss, metrics, d loss / d prediction (zero for the
    supervisor's half and for the pseudo label)."""
    from flow_supervisor_amd.train import sequence_loss_unsup
    g = load("sequence_loss_unsup")
    for name in ("a", "b"):
        B, H, W, n, seed = (int(v) for v in g[name + "_cfg"])
        gamma, lam = (float(v) for v in g[name + "_gamma"])
        preds = [rand_tensor((B, 2, H, W), seed + 10 + i, 3.0).to(DEV).requires_grad_(True) for i in range(n)]
        gt = rand_tensor((B, 2, H, W), seed + 1, 4.0).to(DEV)
        valid = (rand_uniform((B, H, W), seed + 2, 0.0, 1.0) > 0.2).float()
        valid[:, 2, 2] = 0.5
        loss, metrics = sequence_loss_unsup(preds, gt, valid.to(DEV), gamma, lam)
        loss.backward()
        rel_check(loss.item(), g[name + "_loss"], 2e-6, f"unsup loss {name}")
        for k, r in zip(("epe", "1px", "3px", "5px"), g[name + "_metrics"]):
            assert abs(metrics[k] - float(r)) <= 1e-5 + 1e-5 * abs(float(r)), (name, k, metrics[k], float(r))
        for i, p in enumerate(preds):
            got = p.grad if p.grad is not None else torch.zeros_like(p)
            close(got, g[f"{name}_dpred{i}"], 1e-9, 1e-5, what=f"unsup dpred{i}")


@pytest.mark.parametrize("bs", [1, 2])
def test_batched_flow_supervisor_losses_equal_the_two_functions_on_slices(bs):
    """train.semi_sequence_losses (both losses of the batched step on unsliced predictions, two kernel launches on pointer
    offsets) against train.sequence_loss on samples [0, bs) and train.sequence_loss_unsup on samples [bs, 2 bs) -- which are
    pinned to the reference's functions above: same loss values and bit-equal gradients (same kernel, same per-pixel order)
    for every prediction, with different upstream factors on the two losses."""
    from flow_supervisor_amd.train import semi_sequence_losses, sequence_loss, sequence_loss_unsup
    H, W, n = 24, 40, 6
    vals = [rand_tensor((2 * bs, 2, H, W), 300 + i, 3.0).to(DEV) for i in range(n)]
    gt = rand_tensor((bs, 2, H, W), 290, 4.0).to(DEV)
    valid = (rand_uniform((bs, H, W), 291, 0.0, 1.0) > 0.2).float().to(DEV)
    a = [v.clone().requires_grad_(True) for v in vals]
    ls, lu = semi_sequence_losses(a, bs, gt, valid, 0.85, unsup_weight=0.25, gamma_unsup=0.7)
    (2.0 * ls + 3.0 * lu).backward()
    b = [v.clone().requires_grad_(True) for v in vals]
    rs, _ = sequence_loss([p[:bs] for p in b], gt, valid, 0.85, metrics=False)
    ru, _ = sequence_loss_unsup([p[bs:] for p in b], gt, valid, 0.7, 0.25, metrics=False)
    (2.0 * rs + 3.0 * ru).backward()
    rel_check(ls.item(), rs.item(), 2e-6, "labelled loss")           # (block sums meet in float atomics: not bit-equal run to run)
    rel_check(lu.item(), ru.item(), 2e-6, "unlabelled loss")
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x.grad, y.grad), i
    assert float(a[-1].grad[bs:].abs().max()) == 0.0          # the supervisor's half of the unlabelled samples: no gradient


def test_mask_head_and_upsampler_of_all_iterations_as_one_launch():
    """update.HeadBatch (the mask convolution and the convex upsampler of every iteration deferred to one launch each, their
    backward likewise) against the per-iteration path on the same weights and inputs: predictions bit-equal (per-pixel /
    per-image kernels: the batch only changes how many rows a launch sees), every parameter gradient equal to summation-order
    noise (the mask head's weight gradient becomes one 12x longer segment, the encoders' statistics meet in atomics)."""
    from flow_supervisor_amd.core import update as U
    from flow_supervisor_amd.train import raft_sequence_loss
    im1, im2 = (t.to(DEV) for t in synthetic_pair(2, 128, 192, 91))
    out = {}
    was = U.HEAD_BATCH
    try:
        for on in (True, False):
            U.HEAD_BATCH = on
            torch.manual_seed(3)
            m = _model(False, 92).train()
            m.freeze_bn()
            preds = m(im1, im2, iters=5)
            raft_sequence_loss(preds).backward()
            out[on] = ([p.detach().clone() for p in preds], {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    finally:
        U.HEAD_BATCH = was
    for a, b in zip(out[True][0], out[False][0]):
        close(a, b, 2e-5, rtol=0.0, what="prediction with / without the head batch")      # (run-to-run noise of the InstanceNorm atomics)
    assert out[True][1].keys() == out[False][1].keys()
    for n, ga in out[True][1].items():
        gb = out[False][1][n]
        tol = 2e-2 if n.startswith("fnet.") else 2e-3
        assert (ga - gb).norm().item() <= tol * gb.norm().item() + 1e-6, (n, (ga - gb).norm().item(), gb.norm().item())


@pytest.mark.parametrize("B,H,W,C,ld", [(2, 13, 21, 256, 512), (1, 55, 128, 256, 512), (3, 7, 5, 128, 128)])
def test_flow_head_data_gradient_as_a_streaming_kernel(B, H, W, C, ld):
    """fsraft_conv_small_dgrad (data gradient of the flow head's C -> 2 3x3 convolution with the ReLU mask in front of it: 18
    multiply-adds per element) against autograd's conv2d backward on the same weights, ragged sizes and the bench grid, written
    into a channel slice of a wider buffer as the update block does.  fp32 multiply-adds in a fixed order: 1e-6 relative."""
    import torch.nn.functional as F
    from flow_supervisor_amd import ops
    from flow_supervisor_amd.ops import V
    torch.manual_seed(4)
    w = torch.randn(2, C, 3, 3, device=DEV) * 0.1
    x = torch.randn(B, C, H, W, device=DEV)
    dy = torch.randn(B, 2, H, W, device=DEV)
    xr = torch.relu(x).requires_grad_(True)
    F.conv2d(xr, w, padding=1).backward(dy)
    ref = (xr.grad * (x > 0)).permute(0, 2, 3, 1)                      # masked by the ReLU that produced the convolution's input
    dd = torch.zeros(B, H, W, 4, device=DEV)
    dd[..., :2] = dy.permute(0, 2, 3, 1)
    head = torch.zeros(B, H, W, ld, device=DEV)
    head[..., :C] = torch.relu(x).permute(0, 2, 3, 1)
    out = torch.full((B, H, W, ld), 7.0, device=DEV)
    ops.conv_small_dgrad(V(dd, 2), w, V(out, C, 0), V(head, C, 0), B, H, W)
    close(out[..., :C], ref, 0.0, rtol=2e-6, what="flow-head data gradient")
    assert bool((out[..., C:] == 7.0).all()), "channels beyond the slice must stay untouched"
    ops.conv_small_dgrad(V(dd, 2), w, V(out, C, 0), None, B, H, W)
    close(out[..., :C], xr.grad.permute(0, 2, 3, 1), 0.0, rtol=2e-6, what="flow-head data gradient, no mask")


@pytest.mark.parametrize("case", ["raft_b3_iters2", "raft_alt_iters3", "l2l_offsets_per_sample", "l2l_sup_grad_samples"])
def test_per_step_batches_against_the_per_iteration_path_on_odd_shapes(case):
    """update.HeadBatch + MotionBatch + the batched heads backward (everything outside the recurrence once per step) against one
    launch per iteration, beyond the shapes of the golden train steps: a batch of 3 on a 9x13 grid with two iterations,
    alt-corr, L2L with per-sample crop offsets, and L2L with sup_grad_samples=1 (uncropped frames of sample 1 encoded without
    a graph, supervisor-phase backward on sample 0 alone) under a loss that keeps that promise, against the plain forward.
    Predictions to 1e-4 px, every parameter gradient outside the feature encoder to 5e-3 relative (tests/debug_batches.py)."""
    import debug_batches as D
    from flow_supervisor_amd.core.l2l import L2L
    from flow_supervisor_amd.core.raft import RAFT
    torch.manual_seed(123)
    if case.startswith("raft"):
        B, H, W, it, alt = (3, 72, 104, 2, False) if case == "raft_b3_iters2" else (2, 128, 192, 3, True)
        im1, im2 = torch.rand(B, 3, H, W, device=DEV) * 255, torch.rand(B, 3, H, W, device=DEV) * 255
        assert D.compare(case, lambda: RAFT(D.ns(alt)), lambda m: m(im1, im2, iters=it))
    else:
        i1, i2, c1, c2, ox, oy = D.l2l_inputs(2, ([8, 24], [16, 0]))
        if case == "l2l_offsets_per_sample":
            assert D.compare(case, lambda: L2L(D.ns()), lambda m: m(i1, i2, c1, c2, ox, oy, iters=4))
        else:
            assert D.compare(case, lambda: L2L(D.ns()), lambda m: m(i1, i2, c1, c2, ox, oy, iters=5, sup_grad_samples=1), sup_k=1,
                             ref_call=lambda m: m(i1, i2, c1, c2, ox, oy, iters=5))


@pytest.mark.parametrize("alt", [False, True])
def test_eager_train_steps_leave_no_garbage_for_the_cyclic_collector(alt):
    """Device memory allocated after an eager train step must not depend on how many steps ran, WITHOUT the cyclic garbage
    collector: the once-per-step states of the update block (parameter arena, context part, GMA attention) used to sit in
    reference cycles (state -> anchor tensor -> grad_fn -> ctx -> state) that kept the context features and friends alive until a
    generation-2 collection happened to run -- ~27 MB per step at the bench shape, a creeping peak.  (tests/debug_leak.py)"""
    import gc
    from flow_supervisor_amd.train import TrainStep
    m = _model(False, 77).train()
    m.args.alternate_corr = alt            # (AlternateCorrBlock used to sit in a block -> anchor -> grad_fn -> ctx -> block cycle: 32 MB per step)
    m.freeze_bn()
    step = TrainStep(m, lr=1e-5, iters=4)
    im1, im2 = (t.to(DEV) for t in synthetic_pair(2, 128, 192, 78))
    gc.collect()
    gc.disable()
    try:
        seen = []
        for i in range(7):
            step(im1, im2)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()        # (blocks freed while a second stream still used them -- record_stream -- are only returned
            if i >= 2:                      #  once the allocator looks at their events again: without this the count depends on timing)
                seen.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert max(seen) - min(seen) <= 1 << 20, [v / 2 ** 20 for v in seen]


def test_test_mode_skips_the_dropped_upsamples_with_identical_outputs():
    """test_mode returns only the last flow_up (raft.py:141-142); the mask convolution and the upsampler of
    the other iterations are skipped.  Outputs must equal the last training-mode prediction of the same weights (same kernels,
    same inputs; to the run-to-run noise of the encoders' atomically accumulated InstanceNorm statistics, ~1e-5), and the
    update block must have produced no mask on the skipped iterations."""
    torch.manual_seed(1)
    m = _model(False, 55).eval()
    im1, im2 = (t.to(DEV) for t in synthetic_pair(1, 128, 192, 56))
    calls = []
    orig = m.update_block.forward_cl

    def spy(*a, **k):
        out = orig(*a, **k)
        calls.append(out[1] is not None)
        return out
    m.update_block.forward_cl = spy
    with torch.no_grad():
        low, up = m(im1, im2, iters=5, test_mode=True)
        assert calls == [False] * 4 + [True]
        calls.clear()
        preds = m(im1, im2, iters=5)
        assert calls == [True] * 5
    close(up, preds[-1], 1e-4, rtol=0.0, what="test_mode flow_up vs last training-mode prediction")
    assert tuple(low.shape) == (1, 2, 16, 24)


# ----------------------------------------------------------------------------- GMA (row a11, config 5)
def gma_ns():
    return argparse.Namespace(small=False, mixed_precision=False, dropout=0, num_heads=1, position_only=False,
                              position_and_content=False, corr_levels=4, corr_radius=4)


def _sample(gr):
    gr = gr.reshape(-1)
    return gr if gr.numel() <= 4096 else gr[:: gr.numel() // 4096][:4096]


def test_gma_attention_and_aggregate_vs_reference(precision):
    from flow_supervisor_amd.core.gma import Aggregate, Attention
    f = 1.0          # (one set of limits for both arithmetic modes)
    g = load("gma_ops")
    sh = shapes("gma_ops")
    seed, B, H, W = int(g["seed"]), int(g["B"]), int(g["H"]), int(g["W"])
    sd = procedural_state_dict(sh, seed)
    att = Attention(args=gma_ns(), dim=128, heads=1, max_pos_size=160, dim_head=128)
    agg = Aggregate(args=gma_ns(), dim=128, dim_head=128, heads=1)
    assert {"att." + k: list(v.shape) for k, v in att.state_dict().items()} | \
           {"agg." + k: list(v.shape) for k, v in agg.state_dict().items()} == sh
    att.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("att.")}, strict=False)
    agg.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("agg.")})
    att, agg = att.to(DEV), agg.to(DEV)
    ctx = torch.relu(rand_tensor((B, 128, H, W), seed + 1, 1.5)).to(DEV).requires_grad_(True)
    fm = rand_tensor((B, 128, H, W), seed + 2).to(DEV).requires_grad_(True)
    A = att(ctx)
    assert tuple(A.shape) == (B, 1, H * W, H * W)
    out = agg(A, fm)
    close(A, g["attn"], 2e-6 * f, what="attention")
    close(out, g["out"], 2e-5 * f, what="aggregate")
    (out * rand_tensor(tuple(out.shape), seed + 3).to(DEV)).sum().backward()
    close(ctx.grad, g["dctx"], 2e-5 * f, what="dctx")
    close(fm.grad, g["dfm"], 2e-5 * f, what="dfm")
    close(_sample(att.to_qk.weight.grad), g["dparam.att.to_qk.weight"], 2e-4 * f, 1e-3, what="dto_qk")
    close(_sample(agg.to_v.weight.grad), g["dparam.agg.to_v.weight"], 2e-4 * f, 1e-3, what="dto_v")
    close(agg.gamma.grad, g["dparam.agg.gamma"], 2e-4 * f, 1e-3, what="dgamma")
    # the general (multi-head / positional) formulation agrees with the kernels on the single-head case
    with torch.no_grad():
        close(att._forward_general(ctx), A, 1e-5, what="general attention")


def test_attention_map_kept_once_as_records():
    """gma.ATTN_RECORDS: the softmax writes the map over its logits as records (the one copy the training path keeps) and the
    softmax backward reads / writes records.  Forward: bit-identical to the dense map split by ops.to_records.  Backward: the
    gradients of the context features and of to_qk against the dense route (the records' 2^-17 is the only difference), with the
    gradient buffer handed over for in-place use (the `_fs_owned` protocol of update._AttnFn) and with a foreign one (copied)."""
    from flow_supervisor_amd import ops
    from flow_supervisor_amd.core import gma
    from flow_supervisor_amd.core.gma import Attention, is_records
    if not ops.SPLIT_VOLUME_BWD:
        pytest.skip("records are the split-arithmetic route")
    B, H, W, seed = 2, 8, 16, 4242
    att = Attention(args=gma_ns(), dim=128, heads=1, max_pos_size=160, dim_head=128).to(DEV)
    with torch.no_grad():
        att.to_qk.weight.copy_(rand_tensor(tuple(att.to_qk.weight.shape), seed, 0.08).to(DEV))
    x = torch.relu(rand_tensor((B, H, W, 128), seed + 1, 1.5)).to(DEV)
    G = rand_tensor((B, 1, H * W, H * W), seed + 2).to(DEV)
    res = {}
    for mode in ("dense", "records_owned", "records_foreign"):
        xa = x.clone().requires_grad_(True)
        att.to_qk.weight.grad = None
        A = att.forward_cl(xa, records=mode != "dense")
        assert is_records(A) == (mode != "dense") and tuple(A.shape) == (B, 1, H * W, H * W)
        g = G.clone()
        if mode == "records_owned":
            g._fs_owned = True
        A.backward(g)
        if mode == "records_foreign":
            assert torch.equal(g, G), "a gradient buffer that was not handed over must not be overwritten"
        res[mode] = (A.detach().clone(), xa.grad.clone(), att.to_qk.weight.grad.clone())
    dense = ops.to_records(res["dense"][0].view(B, H * W, H * W), amax=ops.amax_one(DEV))      # (probabilities: the scale of a word holding 1.0)
    for mode in ("records_owned", "records_foreign"):
        assert torch.equal(res[mode][0].view(B, H * W, H * W).view(torch.int32), dense.view(torch.int32)), mode
        for got, ref, what in ((res[mode][1], res["dense"][1], "dx"), (res[mode][2], res["dense"][2], "dto_qk")):
            err = (got - ref).abs().max().item()
            assert err <= 2e-5 * ref.abs().max().item() + 1e-9, (mode, what, err, ref.abs().max().item())
    # the reference-API Aggregate refuses a map that holds records (it would read them as probabilities)
    from flow_supervisor_amd.core.gma import Aggregate
    with pytest.raises(TypeError):
        Aggregate(args=gma_ns(), dim=128, dim_head=128, heads=1).to(DEV)(att.forward_cl(x, records=True), x.permute(0, 3, 1, 2))
    # the switch and the shapes the record pair does not cover fall back to the dense map
    assert not is_records(att.forward_cl(x[:, :, :15].contiguous(), records=True))         # N = 120: not a multiple of 32
    old = gma.ATTN_RECORDS
    try:
        gma.ATTN_RECORDS = False
        assert not is_records(att.forward_cl(x, records=True))
    finally:
        gma.ATTN_RECORDS = old


def test_gma_update_block_vs_reference(precision):
    from flow_supervisor_amd.core.gma_update import GMAUpdateBlock
    f = 1.0          # (one set of limits for both arithmetic modes)
    g = load("update_gma")
    sh = shapes("update_gma")
    seed, B, H, W = int(g["seed"]), int(g["B"]), int(g["H"]), int(g["W"])
    blk = GMAUpdateBlock(gma_ns(), hidden_dim=128)
    assert {k: list(v.shape) for k, v in blk.state_dict().items()} == sh
    blk.load_state_dict(procedural_state_dict(sh, seed))
    blk = blk.to(DEV)
    net = torch.tanh(rand_tensor((B, 128, H, W), seed + 10)).to(DEV).requires_grad_(True)
    inp = torch.relu(rand_tensor((B, 128, H, W), seed + 11)).to(DEV).requires_grad_(True)
    corr = rand_tensor((B, 324, H, W), seed + 12, 2.0).to(DEV).requires_grad_(True)
    flow = rand_tensor((B, 2, H, W), seed + 13, 3.0).to(DEV).requires_grad_(True)
    attn = torch.softmax(rand_tensor((B, 1, H * W, H * W), seed + 14, 2.0), -1).to(DEV).requires_grad_(True)
    net2, mask, delta = blk(net, inp, corr, flow, attn)
    close(net2, g["net_out"], 2e-5 * f, what="net"); close(delta, g["delta"], 2e-5 * f, what="delta")
    close(mask, g["mask"], 2e-5 * f, what="mask")
    loss = ((net2 * rand_tensor(tuple(net2.shape), seed + 20).to(DEV)).sum()
            + (delta * rand_tensor(tuple(delta.shape), seed + 21).to(DEV)).sum()
            + (mask * rand_tensor(tuple(mask.shape), seed + 22).to(DEV)).sum())
    loss.backward()
    close(net.grad, g["dnet"], 2e-4 * f, what="dnet"); close(inp.grad, g["dinp"], 2e-4 * f, what="dinp")
    close(corr.grad, g["dcorr"], 2e-4 * f, what="dcorr"); close(flow.grad, g["dflow"], 2e-4 * f, what="dflow")
    close(attn.grad[:, :, ::3, ::3], g["dattn"], 2e-4 * f, what="dattn")
    for k, p in blk.named_parameters():
        ref_n = float(g["dparam_norm." + k])
        assert abs(p.grad.norm().item() - ref_n) <= 2e-4 * f * ref_n + 1e-5, (k, p.grad.norm().item(), ref_n)
        ref = T(g["dparam." + k]).float()
        rel = float((_sample(p.grad).detach().cpu() - ref).norm() / (ref.norm() + 1e-12))
        assert rel <= 2e-4, ("d" + k, rel)


def _gma_model(seed, cls=None):
    from flow_supervisor_amd.core.gma_network import RAFTGMA
    m = (cls or RAFTGMA)(gma_ns())
    missing = m.load_state_dict(procedural_state_dict(shapes("raft_gma"), seed), strict=False)
    assert all(k.endswith("rel_ind") for k in missing.missing_keys) and not missing.unexpected_keys
    return m.to(DEV)


def test_gma_end_to_end_flow_epe(precision):
    g = load("e2e_gma_368x496")
    seed, s = int(g["seed"]), int(g["stride"])
    m = _gma_model(seed).eval()
    im1, im2 = synthetic_pair(1, int(g["H"]), int(g["W"]), seed + 1)
    with torch.no_grad():
        low, up = m(im1.to(DEV), im2.to(DEV), iters=int(g["iters"]), test_mode=True)
    e_low = O.epe(low.cpu(), T(g["flow_low"])).item()
    e_up = O.epe(up[:, :, ::s, ::s].cpu(), T(g["flow_up_strided"])).item()
    print("gma", precision, "EPE low", e_low, "EPE up", e_up)
    assert e_low <= 1e-3 and e_up <= 1e-3, (e_low, e_up)


def test_gma_train_step_loss_and_grads(precision):
    g = load("train_step_gma")
    seed = int(g["seed"])
    m = _gma_model(seed).train()
    m.freeze_bn()
    im1, im2 = synthetic_pair(2, int(g["H"]), int(g["W"]), seed + 1)
    preds = m(im1.to(DEV), im2.to(DEV), iters=int(g["iters"]))
    tol = TRAIN_TOL[precision]
    loss = O.sequence_loss_zero_gt(preds)
    rel_check(loss.item(), g["loss"], tol["loss"], "loss")
    loss.backward()
    close(preds[-1], g["last"], tol["pred"], rtol=0.0, what="last prediction")
    bad = grad_digest_check(m.named_parameters(), g, tol, hprefix=None, skip=("pos_emb",))
    assert not bad, bad[:8]


def test_gma_at_bench_scale(precision):
    """Config 5's own shape (440x1024, N = 7040 attention rows of 28 KB, K = 12 * 128 dattn GEMM), one pair, 12 iterations:
    evaluation EPE and the full train step against the reference's RAFTGMA with gamma = 0.1 (as bench.py sets it)."""
    g = load("e2e_gma_440x1024")
    seed, s = int(g["seed"]), int(g["stride"])
    m = _gma_model(seed)
    with torch.no_grad():
        m.update_block.aggregator.gamma.fill_(float(g["gamma"]))
    m.eval()
    im1, im2 = (t.to(DEV) for t in synthetic_pair(1, int(g["H"]), int(g["W"]), seed + 1))
    with torch.no_grad():
        low, up = m(im1, im2, iters=12, test_mode=True)
    e_low = O.epe(low.cpu(), T(g["flow_low"])).item()
    e_up = O.epe(up[:, :, ::s, ::s].cpu(), T(g["flow_up_strided"])).item()
    print("gma 440x1024", precision, "EPE low", e_low, "EPE up", e_up)
    assert e_low <= 1e-3 and e_up <= 1e-3, (e_low, e_up)
    g = load("train_step_gma_440x1024")
    m.train()
    m.freeze_bn()
    preds = m(im1, im2, iters=12)
    _check_train_digest(m, preds, g, precision, skip=("pos_emb",))


def test_gma_l2l_test_mode_is_the_plain_gma_forward():
    """GMAL2L in test mode runs the student alone (gma_l2l.py:56-124 with test_mode=True): with the same weights it must return what
    RAFTGMA returns.  (The two-phase training schedule itself is pinned by the reference-generated recipe fixture:
    test_flow_supervisor_step_at_the_reference_recipe[gma].)"""
    from flow_supervisor_amd.core.gma_l2l import GMAL2L
    from flow_supervisor_amd.core.gma_network import RAFTGMA
    torch.manual_seed(0)
    m = GMAL2L(gma_ns()).to(DEV).eval()
    ci1, ci2 = (t.to(DEV) for t in synthetic_pair(1, 160, 256, 77))
    im1, im2 = ci1[:, :, 16:144, 40:232].contiguous(), ci2[:, :, 16:144, 40:232].contiguous()
    ref = RAFTGMA(gma_ns()).to(DEV).eval()
    ref.load_state_dict({k: v for k, v in m.state_dict().items() if not k.startswith("grad_update_block.")})
    with torch.no_grad():
        a = m(im1, im2, iters=4, test_mode=True)[1]
        b = ref(im1, im2, iters=4, test_mode=True)[1]
    close(a, b, 1e-6, what="GMAL2L test mode")


# ----------------------------------------------------------------------------- building blocks
def test_gemm_and_layout_kernels():
    from flow_supervisor_amd import ops
    a = torch.randn(2, 70, 323, device=DEV)
    bt = torch.randn(2, 45, 323, device=DEV)
    bn = torch.randn(2, 323, 45, device=DEV)
    close(ops.gemm(a, bt, True, 0.5), 0.5 * a.cpu() @ bt.cpu().transpose(1, 2), 2e-4, what="gemm NT odd")
    close(ops.gemm(a, bn, False, 2.0), 2.0 * a.cpu() @ bn.cpu(), 2e-4, what="gemm NN odd")
    a = torch.randn(1, 256, 512, device=DEV)
    bt = torch.randn(1, 384, 512, device=DEV)
    close(ops.gemm(a, bt, True), a.cpu() @ bt.cpu().transpose(1, 2), 5e-4, what="gemm NT")
    x = torch.randn(2, 37, 5, 9, device=DEV)
    cl = ops.nchw_to_nhwc(x)
    assert cl.shape == (2, 5, 9, 40)
    close(cl[..., :37], x.permute(0, 2, 3, 1), 0)
    close(ops.nhwc_to_nchw(cl, 37), x, 0)


@pytest.mark.parametrize("kh,kw,cin,cout", [(1, 1, 324, 256), (3, 3, 256, 192), (1, 5, 384, 256), (5, 1, 384, 128),
                                            (3, 3, 256, 2), (3, 3, 128, 64), (3, 3, 256, 126), (3, 3, 242, 96)])
def test_conv_igemm_fwd_dgrad_wgrad(kh, kw, cin, cout, precision):
    """One convolution through the C ABI against torch's CPU conv2d (fwd, data grad, weight grad)."""
    import torch.nn.functional as F
    from flow_supervisor_amd import ops
    from flow_supervisor_amd.ops import Dst, V
    B, H, W = 2, 9, 13
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, kh, kw) / math.sqrt(cin * kh * kw)
    b = torch.randn(cout)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, b, padding=(kh // 2, kw // 2))
    gy = torch.randn_like(y)
    y.backward(gy)
    # split the input over two sources when it is big enough (exercises the concat path)
    split = [cin] if cin < 64 else [cin // 2 // 4 * 4, cin - cin // 2 // 4 * 4]
    xs, o = [], 0
    for c in split:
        xs.append(ops.nchw_to_nhwc(x[:, o:o + c].contiguous().to(DEV)))
        o += c
    srcs = [V(t, c) for t, c in zip(xs, split)]
    wd = w.to(DEV)
    wpk = ops.pack_weight(wd, split, 0)
    out = torch.zeros(B, H, W, (cout + 3) // 4 * 4, device=DEV)
    ops.conv_forward(srcs, wpk, b.to(DEV), B, H, W, kh, kw, cout, [Dst.nhwc(out)], wpk_split=ops.pack_weight(wd, split, 10))
    close(ops.nhwc_to_nchw(out, cout), y, 2e-4, what="conv fwd")
    gyc = ops.nchw_to_nhwc(gy.to(DEV))
    wpb = ops.pack_weight(wd, split, 1)
    dxs = [torch.zeros(B, H, W, (c + 3) // 4 * 4, device=DEV) for c in split]
    n0, dsts = 0, []
    for t, c in zip(dxs, split):
        dsts.append(Dst.nhwc(t, 0, n0))
        n0 += c
    ops.conv_forward([V(gyc, cout)], wpb, None, B, H, W, kh, kw, cin, dsts, wpk_split=ops.pack_weight(wd, split, 11))
    dx = torch.cat([ops.nhwc_to_nchw(t, c) for t, c in zip(dxs, split)], 1)
    close(dx, xr.grad, 2e-4, what="conv dgrad")
    dwpk = torch.zeros_like(wpk)
    dbias = torch.zeros(cout, device=DEV)
    ops.conv_wgrad(V(gyc, cout), srcs, dwpk, B, H, W, kh, kw, dbias=dbias)
    dw = ops.unpack_weight_grad(dwpk, tuple(w.shape), split)
    close(dw, wr.grad, 5e-4, what="conv wgrad")
    close(dbias, gy.sum(dim=(0, 2, 3)), 2e-4, what="bias grad (fused into the weight-gradient kernel)")


# ----------------------------------------------------------------------------- properties at full size
def test_full_size_properties_sintel_batch():
    """B=4, 55x128, C=256 (the bench workload): size-independent checks that need no oracle run."""
    from flow_supervisor_amd.core.corr import CorrBlock
    from flow_supervisor_amd.core.utils.utils import coords_grid
    B, C, H, W = 4, 256, 55, 128
    g = torch.Generator(device="cpu").manual_seed(5)
    f1 = torch.randn(B, C, H, W, generator=g).to(DEV)
    f2 = torch.randn(B, C, H, W, generator=g).to(DEV)
    blk = CorrBlock(f1, f2)
    # (1) pooling consistency: level l+1 == avg_pool(level l) (floor)
    for l in range(3):
        ref = torch.nn.functional.avg_pool2d(blk.corr_pyramid[l], 2, stride=2)
        close(blk.corr_pyramid[l + 1], ref, 1e-5, what=f"pool {l}")
    # (2) level 0 against random rows of the exact product
    idx = torch.randint(0, H * W, (64,))
    ref = torch.einsum("bcq,bcn->bqn", f1.view(B, C, -1)[:, :, idx.to(DEV)].double(), f2.view(B, C, -1).double()) / 16.0
    got = blk.corr_pyramid[0].view(B, H * W, H * W)[:, idx.to(DEV)]
    close(got, ref.float(), 2e-4, what="level-0 rows")
    # (3) lookup at integer coordinates returns the volume entries themselves (centre tap)
    coords = coords_grid(B, H, W, device=DEV)
    out = blk(coords)
    centre = out[:, 40]          # level 0, i=4, j=4  -> V[q, y, x]
    diag = blk.corr_pyramid[0].view(B, H * W, H * W).diagonal(dim1=1, dim2=2).reshape(B, H, W)
    close(centre, diag, 1e-5, what="centre tap == V[q,q]")
    # (4) linearity of the build in fmap2
    blk2 = CorrBlock(f1, 2.0 * f2)
    close(blk2.corr_pyramid[3], 2.0 * blk.corr_pyramid[3], 1e-4, what="linearity")


# ----------------------------------------------------------------------------- TF-shaped twins (API parity)
def test_tf_shaped_api_matches_pytorch_shaped_api():
    from flow_supervisor_amd import raft_tf
    from flow_supervisor_amd.core.corr import CorrBlock
    from flow_supervisor_amd.core.raft import convex_upsample
    from flow_supervisor_amd.core.update import BasicUpdateBlock
    B, C, H, W = 1, 64, 16, 24
    f1 = torch.randn(B, C, H, W, device=DEV)
    f2 = torch.randn(B, C, H, W, device=DEV)
    coords = O.coords_grid(B, H, W).to(DEV) + (torch.rand(B, 2, H, W, device=DEV) - 0.5) * 6
    ref = CorrBlock(f1, f2)
    pyr = raft_tf.calc_all_field(f1.permute(0, 2, 3, 1), f2.permute(0, 2, 3, 1), num_pool=3)
    assert [tuple(p.shape) for p in pyr] == [(B, H, W, H >> l, W >> l) for l in range(4)]
    out = raft_tf.CorrBlock(4, 4)(pyr, coords.permute(0, 2, 3, 1))
    close(out.permute(0, 3, 1, 2), ref(coords), 1e-6, what="TF-shaped lookup")
    flow = torch.randn(B, 2, H, W, device=DEV)
    mask = torch.randn(B, 576, H, W, device=DEV)
    up = raft_tf.UpsampleConvexWithMask(8)([flow.permute(0, 2, 3, 1), mask.permute(0, 2, 3, 1).contiguous(),
                                            torch.zeros(B, 8 * H - 3, 8 * W - 5, 2)])
    close(up.permute(0, 3, 1, 2) * 8, convex_upsample(flow, mask)[:, :, : 8 * H - 3, : 8 * W - 5], 1e-5, what="TF-shaped upsampler")
    a = ns(False)
    blk = BasicUpdateBlock(a).to(DEV)
    tfb = raft_tf.BasicUpdateBlock(a).to(DEV)
    tfb.load_state_dict(blk.state_dict())
    net = torch.tanh(torch.randn(B, 128, H, W, device=DEV)); inp = torch.relu(torch.randn(B, 128, H, W, device=DEV))
    corr = torch.randn(B, 324, H, W, device=DEV)
    with torch.no_grad():
        n1, m1, d1 = blk(net, inp, corr, flow)
        n2, m2, d2 = tfb.call([t.permute(0, 2, 3, 1).contiguous() for t in (net, inp, corr, flow)])
    close(n2.permute(0, 3, 1, 2), n1, 1e-6); close(m2.permute(0, 3, 1, 2), m1, 1e-6); close(d2.permute(0, 3, 1, 2), d1, 1e-6)


# ----------------------------------------------------------------------------- encoder-side fused norm + ReLU
@pytest.mark.parametrize("shape", [(2, 8, 20, 32), (1, 5, 7, 9)])
def test_fused_norm_relu_kernels_match_torch(shape):
    from flow_supervisor_amd.core.extractor import _FrozenBNRelu, _InstNormRelu
    torch.manual_seed(3)
    N, C, H, W = shape
    for relu in (True, False):
        x = (torch.randn(N, C, H, W, device=DEV) * 2 + 0.5).requires_grad_(True)
        g = torch.randn(N, C, H, W, device=DEV)
        y = _InstNormRelu.apply(x, 1e-5, relu)
        y.backward(g)
        xr = x.detach().clone().requires_grad_(True)
        yr = torch.nn.functional.instance_norm(xr, eps=1e-5)
        yr = torch.relu(yr) if relu else yr
        yr.backward(g)
        close(y, yr, 1e-5, what="instance norm fwd"); close(x.grad, xr.grad, 1e-5, what="instance norm bwd")
        w = (torch.rand(C, device=DEV) + 0.5).requires_grad_(True); b = torch.randn(C, device=DEV).requires_grad_(True)
        rm, rv = torch.randn(C, device=DEV), torch.rand(C, device=DEV) + 0.5
        x2 = x.detach().clone().requires_grad_(True)
        cb = torch.randn(C, device=DEV).requires_grad_(True)          # bias of the convolution in front, folded in
        y = _FrozenBNRelu.apply(x2, cb, w, b, rm, rv, 1e-5, relu)
        y.backward(g)
        x3 = x.detach().clone().requires_grad_(True); w3 = w.detach().clone().requires_grad_(True); b3 = b.detach().clone().requires_grad_(True)
        cb3 = cb.detach().clone().requires_grad_(True)
        yr = torch.nn.functional.batch_norm(x3 + cb3.view(1, C, 1, 1), rm, rv, w3, b3, False, 0.0, 1e-5)
        yr = torch.relu(yr) if relu else yr
        yr.backward(g)
        close(y, yr, 1e-5, what="frozen bn fwd"); close(x2.grad, x3.grad, 1e-5, what="frozen bn dx")
        close(w.grad, w3.grad, 1e-4, 1e-4, what="frozen bn dweight"); close(b.grad, b3.grad, 1e-4, 1e-4, what="frozen bn dbias")
        close(cb.grad, cb3.grad, 1e-4, 1e-4, what="folded conv bias grad")


@pytest.mark.parametrize("shape", [(2, 64, 20, 32), (1, 96, 7, 9), (3, 8, 5, 5), (2, 256, 9, 4)])
def test_channels_last_norm_relu_kernels_match_torch(shape):
    """csrc/norm_cl.hip: the [N][HW][C] twins of the fused norm + ReLU kernels, on channels_last tensors."""
    from flow_supervisor_amd.core.extractor import _FrozenBNReluCL, _InstNormReluCL
    torch.manual_seed(5)
    N, C, H, W = shape
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    for relu in (True, False):
        x = cl(torch.randn(N, C, H, W, device=DEV) * 2 + 0.5).requires_grad_(True)
        g = cl(torch.randn(N, C, H, W, device=DEV))
        y = _InstNormReluCL.apply(x, 1e-5, relu)
        assert y.is_contiguous(memory_format=torch.channels_last)
        y.backward(g)
        xr = x.detach().contiguous().requires_grad_(True)
        yr = torch.nn.functional.instance_norm(xr, eps=1e-5)
        yr = torch.relu(yr) if relu else yr
        yr.backward(g.contiguous())
        close(y, yr, 1e-5, what="instance norm fwd"); close(x.grad, xr.grad, 2e-5, what="instance norm bwd")
        w = (torch.rand(C, device=DEV) + 0.5).requires_grad_(True); b = torch.randn(C, device=DEV).requires_grad_(True)
        rm, rv = torch.randn(C, device=DEV), torch.rand(C, device=DEV) + 0.5
        x2 = x.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
        cb = torch.randn(C, device=DEV).requires_grad_(True)
        y = _FrozenBNReluCL.apply(x2, cb, w, b, rm, rv, 1e-5, relu)
        y.backward(g)
        x3 = x.detach().contiguous().requires_grad_(True); w3 = w.detach().clone().requires_grad_(True)
        b3 = b.detach().clone().requires_grad_(True); cb3 = cb.detach().clone().requires_grad_(True)
        yr = torch.nn.functional.batch_norm(x3 + cb3.view(1, C, 1, 1), rm, rv, w3, b3, False, 0.0, 1e-5)
        yr = torch.relu(yr) if relu else yr
        yr.backward(g.contiguous())
        close(y, yr, 1e-5, what="frozen bn fwd"); close(x2.grad, x3.grad, 1e-5, what="frozen bn dx")
        close(w.grad, w3.grad, 1e-4, 1e-4, what="frozen bn dweight"); close(b.grad, b3.grad, 1e-4, 1e-4, what="frozen bn dbias")
        close(cb.grad, cb3.grad, 1e-4, 1e-4, what="folded conv bias grad")
        # fused residual unit: relu(res + relu?(norm(x))), gradient to the shortcut included
        r1 = cl(torch.randn(N, C, H, W, device=DEV)).requires_grad_(True); r2 = r1.detach().contiguous().requires_grad_(True)
        x4 = x.detach().clone(memory_format=torch.preserve_format).requires_grad_(True); x5 = x.detach().contiguous().requires_grad_(True)
        y = _InstNormReluCL.apply(x4, 1e-5, relu, r1)
        y.backward(g)
        yr = torch.nn.functional.instance_norm(x5, eps=1e-5)
        yr = torch.relu(r2 + (torch.relu(yr) if relu else yr))
        yr.backward(g.contiguous())
        close(y, yr, 1e-5, what="residual instance norm fwd"); close(x4.grad, x5.grad, 2e-5, what="residual instance norm dx")
        close(r1.grad, r2.grad, 1e-6, what="residual instance norm dres")
        r1.grad = None; r2.grad = None
        x6 = x.detach().clone(memory_format=torch.preserve_format).requires_grad_(True); x7 = x.detach().contiguous().requires_grad_(True)
        y = _FrozenBNReluCL.apply(x6, None, w.detach(), b.detach(), rm, rv, 1e-5, relu, r1)
        y.backward(g)
        yr = torch.nn.functional.batch_norm(x7, rm, rv, w.detach(), b.detach(), False, 0.0, 1e-5)
        yr = torch.relu(r2 + (torch.relu(yr) if relu else yr))
        yr.backward(g.contiguous())
        close(y, yr, 1e-5, what="residual frozen bn fwd"); close(x6.grad, x7.grad, 1e-5, what="residual frozen bn dx")
        close(r1.grad, r2.grad, 1e-6, what="residual frozen bn dres")


@pytest.mark.parametrize("B,C,N,H,W,k", [(2, 64, 64, 20, 32, 3), (1, 96, 96, 7, 9, 3), (2, 8, 24, 13, 5, 3), (3, 32, 32, 40, 24, 3),
                                         (2, 128, 128, 9, 13, 3), (2, 128, 256, 9, 13, 1), (2, 64, 96, 33, 65, 3)])
def test_encoder_conv_channels_last_matches_torch(B, C, N, H, W, k, precision):
    """_ConvCL (forward / data gradient on fsraft_conv_forward, weight + bias gradient on fsraft_conv_wgrad -- the
    tap-packing few-channel kernel for C <= 96) against F.conv2d."""
    from flow_supervisor_amd.core.extractor import _ConvCL, _weight_packs
    f = 1.0          # (one set of limits for both arithmetic modes)
    torch.manual_seed(11)
    conv = torch.nn.Conv2d(C, N, k, padding=k // 2).to(DEV)
    x = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(B, N, H, W, device=DEV)
    y = _ConvCL.apply(x, conv.weight, conv.bias, _weight_packs(conv))
    assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(g)
    dw, db = conv.weight.grad.clone(), conv.bias.grad.clone()
    conv.weight.grad = None; conv.bias.grad = None
    xr = x.detach().contiguous().requires_grad_(True)
    yr = conv(xr)
    yr.backward(g)
    close(y, yr, 2e-5 * f, what="conv fwd"); close(x.grad, xr.grad, 2e-5 * f, what="conv dx")
    close(dw, conv.weight.grad, 1e-4 * f, 1e-4, what="conv dw"); close(db, conv.bias.grad, 1e-4 * f, 1e-4, what="conv db")
    # cache follows the weight: an in-place update must repack
    with torch.no_grad():
        conv.weight.mul_(0.5)
    y2 = _ConvCL.apply(x.detach(), conv.weight, conv.bias, _weight_packs(conv))
    close(y2, conv(xr.detach()), 2e-5 * f, what="conv fwd after weight update")


@pytest.mark.parametrize("B,C,N,H,W", [(2, 64, 64, 20, 32), (1, 64, 64, 13, 37), (2, 48, 40, 9, 70), (1, 36, 64, 4, 5), (3, 64, 36, 33, 31),
                                       (1, 64, 128, 5, 9), (2, 40, 100, 6, 34), (1, 128, 64, 7, 33)])
def test_conv3x3_resident_patch_kernel(B, C, N, H, W):
    """conv3x3_halo_kernel (csrc/conv_igemm.hip: the input patch of a 4 x 32 output tile stays in LDS for all nine taps),
    normally reserved for few-channel layers at encoder resolution, forced on for small ragged shapes: forward with bias +
    ReLU, and the data gradient (the same kernel on the flipped pack), against torch."""
    from flow_supervisor_amd import _lib, ops
    lib = _lib.load()
    lib.fsraft_set_tuning(3, 1); lib.fsraft_set_tuning(4, 2)
    lib.fsraft_set_tuning(21, 0)
    try:
        torch.manual_seed(7)
        w = torch.randn(N, C, 3, 3, device=DEV) * 0.1
        bias = torch.randn(N, device=DEV)
        x = torch.randn(B, H, W, C, device=DEV)
        out = torch.full((B, H, W, N), float("nan"), device=DEV)
        ops.conv_forward([ops.V(x, C)], ops.pack_weight(w, [C], 0), bias, B, H, W, 3, 3, N, [ops.Dst.nhwc(out)], relu=True,
                         wpk_split=ops.pack_weight(w, [C], 10), wpk_frag=ops.fragment_order(ops.pack_weight(w, [C], 10)))
        ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=1)).permute(0, 2, 3, 1)
        close(out, ref, 1e-4, what="resident-patch conv fwd")
        if N % 4 == 0:
            g = torch.randn(B, H, W, N, device=DEV)
            dx = torch.full((B, H, W, C), float("nan"), device=DEV)
            ops.conv_forward([ops.V(g, N)], ops.pack_weight(w, [C], 1), None, B, H, W, 3, 3, C, [ops.Dst.nhwc(dx)],
                             wpk_split=ops.pack_weight(w, [C], 11), wpk_frag=ops.fragment_order(ops.pack_weight(w, [C], 11)))
            dref = torch.nn.grad.conv2d_input((B, C, H, W), w, g.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
            close(dx, dref, 1e-4, what="resident-patch conv dgrad")
    finally:
        lib.fsraft_set_tuning(21, 65536)


@pytest.mark.parametrize("B,C,N,H,W", [(2, 64, 96, 20, 32), (1, 96, 128, 6, 10), (2, 8, 16, 14, 4)])
def test_encoder_strided_pair_space_to_depth(B, C, N, H, W, precision):
    """_StridedPairFn: the 3x3 stride-2 convolution and the 1x1 stride-2 shortcut of a stride-2 residual unit as 2x2 / 1x1
    stride-1 convolutions over the space-to-depth input (fsraft_space_to_depth2, pad override of fsraft_conv_forward),
    against F.conv2d: outputs, input gradient, both weight gradients."""
    from flow_supervisor_amd import ops
    from flow_supervisor_amd.core.extractor import ResidualBlock, _StridedPairFn, _pair_packs
    f = 1.0          # (one set of limits for both arithmetic modes)
    torch.manual_seed(13)
    blk = ResidualBlock(C, N, "instance", stride=2).to(DEV)
    x = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xs = ops.space_to_depth2(x.detach().permute(0, 2, 3, 1))
    assert torch.equal(ops.space_to_depth2(xs, inverse=True), x.detach().permute(0, 2, 3, 1))
    y1, ys = _StridedPairFn.apply(x, blk.conv1.weight, blk.downsample[0].weight, _pair_packs(blk))
    g1, gs = torch.randn_like(y1), torch.randn_like(ys)
    (y1 * g1).sum().add((ys * gs).sum()).backward()
    got = (x.grad.clone(), blk.conv1.weight.grad.clone(), blk.downsample[0].weight.grad.clone())
    blk.zero_grad(set_to_none=True)
    xr = x.detach().contiguous().requires_grad_(True)
    r1 = torch.nn.functional.conv2d(xr, blk.conv1.weight, None, 2, 1)
    rs = torch.nn.functional.conv2d(xr, blk.downsample[0].weight, None, 2, 0)
    (r1 * g1).sum().add((rs * gs).sum()).backward()
    close(y1, r1, 2e-5 * f, what="3x3 stride 2 fwd"); close(ys, rs, 2e-5 * f, what="shortcut fwd")
    close(got[0], xr.grad, 3e-5 * f, what="strided pair dx")
    close(got[1], blk.conv1.weight.grad, 1e-4 * f, 1e-4, what="3x3 stride 2 dw")
    close(got[2], blk.downsample[0].weight.grad, 1e-4 * f, 1e-4, what="shortcut dw")


@pytest.mark.parametrize("C,N,B,H,W,carried", [(64, 64, 2, 184, 250, True), (96, 96, 2, 184, 250, True), (128, 128, 8, 55, 128, True),
                                               (128, 128, 3, 55, 128, False), (64, 64, 1, 20, 24, False)])
def test_instance_norm_statistics_from_the_convolution_epilogue(C, N, B, H, W, carried):
    """fsraft_conv_forward_stats: the 3x3 encoder convolutions (extractor.py:13-57) on the halo / resident-patch kernels add the
    per-image column sums of their result and of its squares to the [2, B * 8, N] partial rows the InstanceNorm kernels read
    (ragged widths: pixels outside the image count as nothing); small grids take a kernel that does not and say so.  The sums
    against float64 sums of the convolution's own output, then norm(conv(x)) with and without the hand-over."""
    from flow_supervisor_amd.core import extractor as E
    torch.manual_seed(31)
    conv = torch.nn.Conv2d(C, N, 3, padding=1).to(DEV)
    x = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        h = E._NormSums()
        y = E._conv(conv, x, None, sums=h)
        assert (h.acc is not None) == carried, "which kernels carry the statistics changed"
        if carried:
            got = h.acc.view(2, B, 8, N).sum(2).double().cpu()
            yd = y.double()
            ref = torch.stack([yd.sum((2, 3)), yd.square().sum((2, 3))]).cpu()
            close(got[0], ref[0], 1e-3, rtol=1e-5, what="column sums")         # (sums of ~1e4 zero-mean terms: absolute part)
            close(got[1], ref[1], 0.0, rtol=1e-5, what="column sums of squares")
        norm = torch.nn.InstanceNorm2d(N)
        outs = {}
        old = E.STATS_IN_EPILOGUE
        try:
            for flag in (True, False):
                E.STATS_IN_EPILOGUE = flag
                outs[flag] = E._conv_norm(conv, norm, x, True)
        finally:
            E.STATS_IN_EPILOGUE = old
        close(outs[True], outs[False], 2e-6, rtol=2e-6, what="relu(norm(conv(x))) with the statistics from the epilogue")
        close(outs[True], torch.relu(norm(torch.nn.functional.conv2d(x, conv.weight, None, padding=1))), 2e-4, what="vs torch")


@pytest.mark.parametrize("norm,H,W", [("instance", 72, 104), ("batch", 72, 104), ("instance", 184, 248), ("instance", 88, 100)])
def test_norm_writes_the_space_to_depth_input_of_the_stride_two_unit(norm, H, W, monkeypatch):
    """The residual unit in front of a stride-2 unit (extractor.py:23-57, layer1 -> layer2 -> layer3) hands its output over AS
    the space-to-depth tensor the unit's two convolutions read: its last norm kernel writes that layout and its backward reads
    the gradient from it (fsraft_*_relu_cl_fwd/bwd, s2d_w).  Against the route with the two layout copies per unit
    (extractor.S2D_EMIT = False): forward values and gradients to the run-to-run spread of the atomically accumulated
    sums; the number of layout copies is counted.  88 x 100: the last stride-2 unit sees an odd size (22 x 25) and takes the
    framework's strided convolutions, with an ordinary tensor handed to it."""
    from flow_supervisor_amd.core import extractor as E
    torch.manual_seed(5)
    enc = E.BasicEncoder(output_dim=128, norm_fn=norm).to(DEV)
    if norm == "batch":
        enc.eval()
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(); m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 3, H, W, device=DEV)
    calls = []
    orig = E.ops.space_to_depth2
    monkeypatch.setattr(E.ops, "space_to_depth2", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    outs = {}
    old = E.S2D_EMIT
    try:
        for flag in (True, False):
            E.S2D_EMIT = flag
            calls.clear()
            enc.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(True)
            y = enc(xi)
            (y.square().sum()).backward()
            outs[flag] = (y.detach().clone(), xi.grad.clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}, len(calls))
    finally:
        E.S2D_EMIT = old
    # layout copies (forward + backward) of the stride-2 units that run on the space-to-depth route (even input size)
    n_ok = sum(1 for d in (2, 4) if (H // d) % 2 == 0 and (W // d) % 2 == 0)
    assert outs[False][3] == 2 * n_ok and outs[True][3] == 0, (outs[True][3], outs[False][3])
    # (same arithmetic, other addresses; the statistics themselves are atomically accumulated -- in the convolution epilogues at
    #  the larger sizes -- so two runs of ONE route already differ in the last digits)
    close(outs[True][0], outs[False][0], 1e-4, rtol=1e-4, what="encoder output")
    # gradients: fifteen normalisation backward passes amplify the last-digit differences of the sums (see
    # test_encoder_channels_last_path_matches_nchw_path: ~6e-3 in the image gradient between two runs in split mode)
    assert _rel_l2(outs[True][1], outs[False][1]) < 2e-2
    for k, v in outs[False][2].items():
        assert _rel_l2(outs[True][2][k], v) < 2e-2 or v.norm().item() < 1e-3, k


def test_context_encoder_output_stays_channels_last(monkeypatch):
    """The context encoder's output (raft.py:107-111: split, tanh, relu) is consumed channels-last by the update block: with
    `out_channels_last` the encoder hands it over in that layout and `to_channels_last` is a view -- three layout copies per
    direction less.  Same predictions as with the NCHW hand-over (the values never change, only where they live), and the
    copies are counted."""
    from flow_supervisor_amd import ops
    g = load("train_step_basic")
    seed = int(g["seed"])
    im1, im2 = (t.to(DEV) for t in synthetic_pair(2, int(g["H"]), int(g["W"]), seed + 1))
    calls = []
    orig = ops.nchw_to_nhwc
    monkeypatch.setattr(ops, "nchw_to_nhwc", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    res = {}
    for flag in (True, False):
        m = _model(False, seed).train()
        m.freeze_bn()
        m.cnet.out_channels_last = flag
        calls.clear()
        preds = m(im1, im2, iters=3)
        O.sequence_loss_zero_gt(preds).backward()
        res[flag] = (preds[-1].detach().clone(), len(calls), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert res[True][1] <= res[False][1] - 3, (res[True][1], res[False][1])
    close(res[True][0], res[False][0], 1e-5, rtol=1e-5, what="last prediction")
    for k, v in res[False][2].items():
        e = _rel_l2(res[True][2][k], v)
        assert e < 2e-2 or v.norm().item() < 1e-3, f"{k}: relative L2 error {e:.3e}"


def _rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("kind,norm,s2d", [("basic", "instance", "1"), ("basic", "batch", "1"), ("small", "instance", "1"),
                                           ("small", "none", "1"), ("basic", "instance", "0"), ("basic", "batch", "0")])
def test_encoder_channels_last_path_matches_nchw_path(kind, norm, s2d, precision, monkeypatch):
    """The channels_last encoder (FSRAFT_ENCODER_CL=1, default: fsraft convolutions + norm kernels) against the all-MIOpen
    NCHW encoder (=0): outputs, input gradient and every parameter gradient.  Gradients are compared in relative L2:
    fifteen ReLU layers deep, a pre-activation that sits within rounding of zero flips its mask and moves a handful of
    gradient entries by O(1) in either implementation, which a max-abs bound cannot tell from a real error."""
    from flow_supervisor_amd.core.extractor import BasicEncoder, SmallEncoder
    torch.manual_seed(21)
    # s2d "0": the stride-2 units fall back to MIOpen behind layout hops (the path odd-sized inputs take)
    import flow_supervisor_amd.core.extractor as X
    monkeypatch.setattr(X, "S2D_UNITS", s2d == "1")
    enc = (BasicEncoder if kind == "basic" else SmallEncoder)(output_dim=128, norm_fn=norm).to(DEV)
    if norm == "batch":
        enc.eval()
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(); m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 3, 72, 104, device=DEV)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("FSRAFT_ENCODER_CL", mode)
        enc.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        a, b = enc([xi[:1], xi[1:]])
        assert a.is_contiguous()
        (a.square().sum() + (b * 0.5).sum()).backward()
        outs[mode] = (torch.cat([a, b]).detach(), xi.grad.clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None})
    tol = 1e-5
    # Gradients: the exact-mode kernels agree with MIOpen to ~4e-6 (basic/instance) .. 3e-4 (mask flips).  In split mode
    # every layer's data gradient carries ~2^-17 relative rounding noise, and fifteen normalisation backward passes (each
    # subtracts the mean and the xhat-projection of the incoming gradient -- a difference of large numbers for this
    # random-init, squared-output objective) amplify it to ~6e-3 in the image gradient (measured in round 2, docs/history).
    # The bottleneck (small) encoder has half as many channels again per norm and measures 2.8e-2.  The wiring of the path is
    # what the exact-mode run pins down; the split arithmetic itself is bounded per layer by the convolution tests above.
    # (atomic accumulation order makes the flips differ from run to run: the exact-mode bound leaves room for them)
    gtol = 1e-2
    close(outs["1"][0], outs["0"][0], 2e-4, what="encoder out")
    assert _rel_l2(outs["1"][0], outs["0"][0]) < tol
    e = _rel_l2(outs["1"][1], outs["0"][1])
    assert e < gtol, f"encoder dx: relative L2 error {e:.3e}"
    assert outs["1"][2].keys() == outs["0"][2].keys()
    for k, v in outs["0"][2].items():
        e = _rel_l2(outs["1"][2][k], v)
        assert e < gtol or v.norm().item() < 1e-3, f"encoder grad {k}: relative L2 error {e:.3e}"


def test_residual_unit_input_with_a_third_consumer_and_a_hook(precision):
    """ADVICE r2 / VERDICT r3 #8: a stride-1 residual unit merges the shortcut's gradient and its first convolution's data gradient
    inside that convolution's epilogue (_ResLink).  The merged tensor is what the convolution's backward RETURNS, so autograd owns
    it like any gradient: an input x with a third consumer outside the unit, a tensor hook on x and retain_grad must all see
    the same numbers as plain torch ops on the same weights."""
    import torch.nn.functional as F
    from flow_supervisor_amd.core.extractor import ResidualBlock
    torch.manual_seed(5)
    blk = ResidualBlock(64, 64, "instance", stride=1).to(DEV)
    x0 = torch.randn(2, 64, 24, 40, device=DEV)
    w3 = torch.randn(2, 64, 24, 40, device=DEV)

    def run(fast):
        x = x0.clone().requires_grad_(True)
        xc = (x * 1.5).contiguous(memory_format=torch.channels_last) if fast else x * 1.5       # a non-leaf input, as inside the encoder
        seen = []
        xc.register_hook(lambda g: seen.append(g.detach().clone()))
        xc.retain_grad()
        if fast:
            out = blk(xc)
        else:
            y = F.relu(F.instance_norm(F.conv2d(xc, blk.conv1.weight, blk.conv1.bias, padding=1)))
            y = F.relu(F.instance_norm(F.conv2d(y, blk.conv2.weight, blk.conv2.bias, padding=1)))
            out = F.relu(xc + y)
        loss = (out * out).sum() + (xc * w3).sum()            # the third consumer of the unit's input
        blk.zero_grad(set_to_none=True)
        loss.backward()
        assert len(seen) == 1
        return out.detach(), x.grad.clone(), seen[0], xc.grad.clone(), blk.conv1.weight.grad.clone()

    of, gf, hf, rf, wf = run(True)
    orf, gr, hr, rr, wr = run(False)
    tol = 2e-4
    close(of, orf, tol, what="residual unit out")
    for a, b, nm in ((gf, gr, "dx"), (hf, hr, "gradient seen by the hook"), (rf, rr, "retained gradient"), (wf, wr, "dconv1.weight")):
        e = _rel_l2(a, b)
        # (split mode: two InstanceNorm backward passes amplify the ~2^-17 per-product noise, as in the encoder test above; the wiring is
        #  what the exact-mode run pins)
        assert e < 1e-4, f"{nm}: relative L2 error {e:.3e}"
    assert torch.equal(hf, rf)


# ----------------------------------------------------------------------------- ragged / odd shapes against the oracle
@pytest.mark.parametrize("B,H,W", [(3, 7, 9), (1, 9, 33), (2, 16, 8)])
def test_update_block_odd_shapes_vs_oracle(B, H, W, precision):
    """Shapes that are not multiples of any tile (M = 189, 297, 256 pixels; W < 32): forward and input gradients of the
    basic update block against the CPU oracle, which is pinned by the golden fixtures."""
    from flow_supervisor_amd.core.update import BasicUpdateBlock
    f = 1.0          # (one set of limits for both arithmetic modes)
    seed = 900 + H
    blk = BasicUpdateBlock(ns(False), hidden_dim=128)
    sd = procedural_state_dict(shapes("update_basic"), seed)
    blk.load_state_dict(sd)
    blk = blk.to(DEV)
    mk = lambda shp, s, sc=1.0: rand_tensor(shp, s, sc)
    net_c = torch.tanh(mk((B, 128, H, W), seed + 1)); inp_c = torch.relu(mk((B, 128, H, W), seed + 2))
    corr_c = mk((B, 324, H, W), seed + 3, 2.0); flow_c = mk((B, 2, H, W), seed + 4, 3.0)
    ins_c = [t.clone().requires_grad_(True) for t in (net_c, inp_c, corr_c, flow_c)]
    n_r, m_r, d_r = O.basic_update_block(sd, "", *ins_c)
    wn, wm, wd = mk(tuple(n_r.shape), seed + 5), mk(tuple(m_r.shape), seed + 6), mk(tuple(d_r.shape), seed + 7)
    ((n_r * wn).sum() + (m_r * wm).sum() + (d_r * wd).sum()).backward()
    ins_g = [t.to(DEV).requires_grad_(True) for t in (net_c, inp_c, corr_c, flow_c)]
    n_g, m_g, d_g = blk(*ins_g)
    ((n_g * wn.to(DEV)).sum() + (m_g * wm.to(DEV)).sum() + (d_g * wd.to(DEV)).sum()).backward()
    close(n_g, n_r, 2e-5 * f, what="net"); close(m_g, m_r, 2e-5 * f, what="mask"); close(d_g, d_r, 2e-5 * f, what="delta")
    for a, b, nm in zip(ins_g, ins_c, ("dnet", "dinp", "dcorr", "dflow")):
        close(a.grad, b.grad, 3e-4 * f, what=nm)
    pg = dict(blk.named_parameters())
    sd2 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    n2, m2, d2 = O.basic_update_block(sd2, "", net_c, inp_c, corr_c, flow_c)
    ((n2 * wn).sum() + (m2 * wm).sum() + (d2 * wd).sum()).backward()
    for k, p in pg.items():
        ref = sd2[k].grad
        rel = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12))
        # split mode: with only ~200 pixels one ReLU that flips at a near-zero pre-activation moves a weight gradient by
        # a few 1e-3 of its norm (see test_update_block_vs_reference)
        assert rel <= 2e-4, (k, rel)


def test_lookup_far_out_of_range_and_zero_volume():
    """Windows that lie entirely outside the map read zeros (grid_sample zero padding); a query on the border mixes."""
    from flow_supervisor_amd.core.corr import CorrBlock
    B, C, H, W = 1, 16, 9, 13
    f1 = rand_tensor((B, C, H, W), 41); f2 = rand_tensor((B, C, H, W), 42)
    coords = O.coords_grid(B, H, W)
    coords[:, 0] += 1000.0                        # every window far to the right of the image
    out = CorrBlock(f1.to(DEV), f2.to(DEV))(coords.to(DEV))
    assert float(out.abs().max()) == 0.0
    coords = O.coords_grid(B, H, W) + torch.tensor([-4.5, 3.25]).view(1, 2, 1, 1)
    ref = O.corr_lookup(O.corr_pyramid(f1, f2, 4), coords, 4)
    close(CorrBlock(f1.to(DEV), f2.to(DEV))(coords.to(DEV)), ref, 2e-5, what="border lookup")


def test_gma_update_block_unaligned_pixel_count(precision):
    """N = 7*9 = 63 is not a multiple of 4: the attention GEMMs take their transposed-copy paths."""
    from flow_supervisor_amd.core.gma import Attention
    from flow_supervisor_amd.core.gma_update import GMAUpdateBlock
    f = 1.0          # (one set of limits for both arithmetic modes)
    B, H, W, seed = 2, 7, 9, 950
    sh = shapes("update_gma")
    sd = procedural_state_dict(sh, seed)
    blk = GMAUpdateBlock(gma_ns(), hidden_dim=128); blk.load_state_dict(sd); blk = blk.to(DEV)
    att = Attention(args=gma_ns(), dim=128, heads=1, max_pos_size=160, dim_head=128).to(DEV)
    asd = {"att.to_qk.weight": att.to_qk.weight.detach().cpu()}
    mk = lambda shp, s, sc=1.0: rand_tensor(shp, s, sc)
    net_c = torch.tanh(mk((B, 128, H, W), seed + 1)); inp_c = torch.relu(mk((B, 128, H, W), seed + 2))
    corr_c = mk((B, 324, H, W), seed + 3, 2.0); flow_c = mk((B, 2, H, W), seed + 4, 3.0)
    ctx_c = inp_c.clone().requires_grad_(True)
    a_r = O.gma_attention(asd, "att.", ctx_c)
    n_r, m_r, d_r = O.gma_update_block(sd, "", net_c, inp_c, corr_c, flow_c, a_r)
    wn = mk(tuple(n_r.shape), seed + 5)
    (n_r * wn).sum().backward()
    ctx_g = inp_c.to(DEV).requires_grad_(True)
    a_g = att(ctx_g)
    n_g, m_g, d_g = blk(net_c.to(DEV), inp_c.to(DEV), corr_c.to(DEV), flow_c.to(DEV), a_g)
    (n_g * wn.to(DEV)).sum().backward()
    close(a_g, a_r, 2e-6 * f, what="attention"); close(n_g, n_r, 2e-5 * f, what="net"); close(d_g, d_r, 2e-5 * f, what="delta")
    close(ctx_g.grad, ctx_c.grad, 1e-4 * f, what="d context through attention")


# ----------------------------------------------------------------------------- fused sequence loss (step next to the path)
def test_sequence_loss_matches_restatement():
    """pytorch/train.py:60-96 on the fused kernel vs the oracle restatement on further random inputs (the restatement itself
    is pinned by tests/test_oracle_vs_golden.py::test_sequence_loss_restatement_vs_reference_function)."""
    from flow_supervisor_amd.train import raft_sequence_loss, sequence_loss
    torch.manual_seed(5)
    B, H, W, n = 2, 24, 40, 6
    preds_c = [(torch.randn(B, 2, H, W) * 3).requires_grad_(True) for _ in range(n)]
    gt = torch.randn(B, 2, H, W) * 4
    gt[0, :, :3, :5] = 500.0                                     # beyond max_flow: excluded
    valid = (torch.rand(B, H, W) > 0.2).float()
    l_r, m_r = O.sequence_loss(preds_c, gt, valid, 0.8, 1.0, 400.0)
    l_r.backward()
    preds_g = [p.detach().to(DEV).requires_grad_(True) for p in preds_c]
    l_g, m_g = sequence_loss(preds_g, gt.to(DEV), valid.to(DEV), 0.8, 1.0, 400.0)
    l_g.backward()
    assert abs(l_g.item() - l_r.item()) <= 1e-5 * abs(l_r.item())
    for k in m_r:
        assert abs(m_g[k] - m_r[k]) <= 1e-5 + 1e-5 * abs(m_r[k]), (k, m_g[k], m_r[k])
    for a, b in zip(preds_g, preds_c):
        close(a.grad, b.grad, 1e-8, 1e-4, what="d loss / d prediction")
    z = raft_sequence_loss([p.detach() for p in preds_g])
    assert abs(z.item() - O.sequence_loss_zero_gt([p.detach() for p in preds_c]).item()) <= 1e-5 * abs(z.item())


def _seq_loss_cases():
    g = load("sequence_loss")
    for name in ("a", "b", "c"):
        B, H, W, n, seed = (int(v) for v in g[name + "_cfg"])
        gamma, gamma2 = (float(v) for v in g[name + "_gamma"])
        preds = [rand_tensor((B, 2, H, W), seed + 10 + i, 3.0) for i in range(n)]
        gt = rand_tensor((B, 2, H, W), seed + 1, 4.0)
        gt[:, :, 0, :3] = 500.0
        gt[:, 0, 1, 1] = 300.0; gt[:, 1, 1, 1] = 300.0
        valid = (rand_uniform((B, H, W), seed + 2, 0.0, 1.0) > 0.2).float()
        valid[:, 2, 2] = 0.5
        yield name, g, preds, gt, valid, gamma, gamma2


def test_sequence_loss_vs_reference_function():
    """csrc/loss.hip against outputs of the reference's own sequence_loss (pytorch/train.py:60-96, extracted from the module's
    syntax tree by tests/golden/make_golden.py): loss, metrics, d loss / d prediction; invalid pixels, |gt| >= max_flow, the
    valid == 0.5 edge and the gamma / gamma2 halves."""
    from flow_supervisor_amd.train import sequence_loss
    for name, g, preds, gt, valid, gamma, gamma2 in _seq_loss_cases():
        pg = [p.to(DEV).requires_grad_(True) for p in preds]
        loss, metrics = sequence_loss(pg, gt.to(DEV), valid.to(DEV), gamma, gamma2)
        loss.backward()
        ref = float(g[name + "_loss"])
        assert abs(loss.item() - ref) <= 2e-6 * abs(ref), (name, loss.item(), ref)
        for k, r in zip(("epe", "1px", "3px", "5px"), g[name + "_metrics"]):
            assert abs(metrics[k] - float(r)) <= 1e-5 + 1e-5 * abs(float(r)), (name, k, metrics[k], float(r))
        for i, p in enumerate(pg):
            close(p.grad, g[f"{name}_dpred{i}"], 1e-9, 1e-4, what=f"{name}: d loss / d pred {i}")


# ----------------------------------------------------------------------------- warm start (section 8f rank 4)
def test_forward_interpolate_matches_reference_outputs_and_oracle():
    """fsraft_forward_interpolate (csrc/warm_start.hip) against (a) outputs of the reference's forward_interpolate stored in
    tests/golden/warm_start.npz and (b) the oracle restatement on fresh seeded flows, bit for bit: the result is a copy of
    input vectors, so there is no rounding to allow for."""
    from flow_supervisor_amd.core.utils.utils import forward_interpolate
    g = load("warm_start")
    for name in ("a", "b", "c", "shift"):
        out = forward_interpolate(T(g["in_" + name]).to(DEV))
        assert out.is_cuda and torch.equal(out.cpu(), T(g["out_" + name])), name
    for h, w, scale, seed in ((55, 128, 10.0, 901), (33, 47, 3.0, 902), (1, 9, 2.0, 903), (8, 8, 100.0, 904)):
        flow = rand_tensor((2, h, w), seed, scale)
        ref = O.forward_interpolate(flow) if h * w > 1 and _lands(flow) else torch.zeros_like(flow)
        assert torch.equal(forward_interpolate(flow.to(DEV)).cpu(), ref), (h, w)
    with pytest.raises(ValueError):
        forward_interpolate(torch.zeros(1, 2, 4, 4, device=DEV))


def _lands(flow):
    _, h, w = flow.shape
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    x1, y1 = xs + flow[0].double(), ys + flow[1].double()
    return bool(((x1 > 0) & (x1 < w) & (y1 > 0) & (y1 < h)).any())


def test_tf_backward_flow_pyramid_from_transposed_volume():
    """raft/semi.py:250-251: the backward-flow pyramid is build_pyramid(transpose(forward volume)).  Against a fresh all-pairs
    build with the feature maps swapped (same dot products) and against torch's own transpose + avg_pool2d."""
    from flow_supervisor_amd import raft_tf
    torch.manual_seed(31)
    B, C, H, W = 2, 64, 16, 24
    f1 = torch.randn(B, H, W, C, device=DEV); f2 = torch.randn(B, H, W, C, device=DEV)
    fw = raft_tf.calc_all_field(f1, f2, num_pool=3)
    vt = raft_tf.transpose_volume(fw[0])
    assert torch.equal(vt, fw[0].permute(0, 3, 4, 1, 2).contiguous())
    bw = raft_tf.build_pyramid(vt, num_pool=3)
    swapped = raft_tf.calc_all_field(f2, f1, num_pool=3)
    assert [tuple(p.shape) for p in bw] == [tuple(p.shape) for p in swapped]
    lv = vt.reshape(B * H * W, 1, H, W)
    for l in range(4):
        close(bw[l], swapped[l], 2e-5, what=f"backward pyramid level {l} vs swapped build")
        close(bw[l].reshape(lv.shape), lv, 1e-6, what=f"backward pyramid level {l} vs avg_pool2d")
        lv = torch.nn.functional.avg_pool2d(lv, 2, 2)


def _tf_same_avg_pool(x, k):
    """tf.nn.avg_pool2d(x, k, k, 'SAME') restated: out = ceil(n / k), padding out * k - n split floor / ceil (leading /
    trailing), padded cells excluded from the average.  x: [R, H, W] on the CPU."""
    R, H, W = x.shape
    h2, w2 = -(-H // k), -(-W // k)
    py, px = (h2 * k - H) // 2, (w2 * k - W) // 2
    out = torch.empty(R, h2, w2, dtype=x.dtype)
    for y in range(h2):
        y0, y1 = max(y * k - py, 0), min(y * k - py + k, H)
        for xx in range(w2):
            x0, x1 = max(xx * k - px, 0), min(xx * k - px + k, W)
            out[:, y, xx] = x[:, y0:y1, x0:x1].mean(dim=(1, 2))
    return out


@pytest.mark.parametrize("H,W", [(11, 16), (7, 13), (9, 10)])
def test_tf_same_pooling_pyramid_and_lookup(H, W):
    """Odd pooled sizes (55 rows at 1/8 of Sintel): the TF twins build the pyramid with TF's 'SAME' pooling (ceil sizes) and
    look it up on those sizes.  PARITY UNPINNED (no TensorFlow here): checked against a restatement of the documented
    tf.nn.avg_pool2d semantics and against an explicit bilinear gather (the oracle's lookup on the SAME pyramid)."""
    from flow_supervisor_amd import raft_tf
    torch.manual_seed(41)
    B, C = 2, 32
    f1 = torch.randn(B, H, W, C, device=DEV); f2 = torch.randn(B, H, W, C, device=DEV)
    pyr = raft_tf.calc_all_field(f1, f2, num_pool=3)
    assert [tuple(p.shape[-2:]) for p in pyr] == [(-(-H // (1 << l)), -(-W // (1 << l))) for l in range(4)]
    v0 = pyr[0].reshape(B * H * W, H, W).cpu()
    ref_levels = [v0] + [_tf_same_avg_pool(v0, 1 << l) for l in range(1, 4)]
    for l in range(4):
        close(pyr[l].reshape(ref_levels[l].shape), ref_levels[l], 2e-6, what=f"SAME level {l}")
    assert [tuple(p.shape) for p in raft_tf.build_pyramid(pyr[0], 3)] == [tuple(p.shape) for p in pyr]
    coords = (O.coords_grid(B, H, W) + (torch.rand(B, 2, H, W) - 0.5) * 5).to(DEV)
    out = raft_tf.CorrBlock(4, 4)(pyr, coords.permute(0, 2, 3, 1))
    ref = O.corr_lookup([lv.reshape(B * H * W, 1, lv.shape[-2], lv.shape[-1]) for lv in ref_levels], coords.cpu(), 4)
    close(out.permute(0, 3, 1, 2), ref, 2e-5, what="lookup on the SAME pyramid")


@pytest.mark.parametrize("B,H,W,cs,N,kh,kw,nseg", [(2, 13, 37, [128, 128, 128], 256, 1, 5, 3), (1, 21, 40, [256], 192, 3, 3, 2),
                                                    (2, 9, 33, [128, 128], 128, 5, 1, 2), (1, 7, 70, [126], 96, 3, 3, 4),
                                                    (3, 55, 128, [64], 126, 3, 3, 2), (1, 3, 5, [48, 20], 40, 1, 5, 2)])
def test_resident_block_weight_gradient(B, H, W, cs, N, kh, kw, nseg):
    """conv_wgrad_patch_kernel (csrc/wgrad_patch.inc, fsraft_set_tuning key 27): the multi-segment weight gradient of the
    3x3 / 1x5 / 5x1 layers over resident pixel blocks, against an fp64 convolution weight gradient and against the per-tap
    kernel it replaces.  Ragged H / W (partial 4 x 32 blocks, halo clipping), channel counts that are not multiples of 64 or
    of 4, several sources, several segments, bias gradient."""
    from flow_supervisor_amd import _lib, ops
    lib = _lib.load()
    lib.fsraft_set_tuning(3, 1); lib.fsraft_set_tuning(4, 2)
    torch.manual_seed(B * 1000 + H * 10 + kh)
    pad4 = lambda c: (c + 3) // 4 * 4
    xs = [[torch.randn(B, H, W, pad4(c), device=DEV) for c in cs] for _ in range(nseg)]
    dys = [torch.randn(B, H, W, pad4(N), device=DEV) for _ in range(nseg)]
    cin = sum(cs)
    res = []
    for flag in (1, 0):
        lib.fsraft_set_tuning(27, flag)
        dwpk = torch.zeros_like(ops.pack_weight(torch.zeros(N, cin, kh, kw, device=DEV), cs, 0))
        dbias = torch.zeros(N, device=DEV)
        ops.conv_wgrad_multi([ops.V(t, N) for t in dys], [[ops.V(t, c) for t, c in zip(x, cs)] for x in xs], dwpk, B, H, W, kh, kw,
                             dbias=dbias)
        res.append((dwpk, dbias))
    lib.fsraft_set_tuning(27, 1)
    # fp64 reference: the weight gradient of the same-padded convolution, packed like the weights
    w = torch.zeros(N, cin, kh, kw, dtype=torch.float64, device=DEV, requires_grad=True)
    ref_b = torch.zeros(N, dtype=torch.float64, device=DEV)
    for x, dy in zip(xs, dys):
        xin = torch.cat([t[..., :c] for t, c in zip(x, cs)], -1).permute(0, 3, 1, 2).double()
        y = torch.nn.functional.conv2d(xin, w, padding=(kh // 2, kw // 2))
        y.backward(dy[..., :N].permute(0, 3, 1, 2).double())
        ref_b += dy[..., :N].double().sum((0, 1, 2))
    ref = ops.pack_weight(w.grad.float(), cs, 0)
    real = ops.pack_weight(torch.ones(N, cin, kh, kw, device=DEV), cs, 0)     # 0 in the pad columns of the packed layout: nobody reads those
    scale = ref.abs().max().item()
    for (dwpk, dbias), name in zip(res, ("resident blocks", "per tap")):
        assert ((dwpk - ref) * real).abs().max().item() <= 3e-5 * scale, name
        close(dbias, ref_b.float(), 1e-5, what="bias gradient, " + name)
    assert ((res[0][0] - res[1][0]) * real).abs().max().item() <= 2e-5 * scale


def test_record_gemms_against_fp64():
    """fsraft_to_records / fsraft_gemm_rec_nt / fsraft_gemm_rec_tn (csrc/gemm_rec.hip, the LDS-DMA record core): ragged shapes,
    split-K with atomics, explicit pitches, accumulate.  Split products (fp16 pieces of scaled operands): relative error ~2^-22 per product."""
    from flow_supervisor_amd import ops
    torch.manual_seed(31)
    x = torch.randn(3, 50, 77, device=DEV)
    r = ops.to_records(x)                                   # [.., 96]: three records per row, the tail of the last one zero
    sc = word_scale(ops.amax_of(r))
    assert 0.5 * x.abs().max().item() < ops.amax_of(r).item() <= 4.0 * x.abs().max().item() and 2.0 ** 11 <= sc * x.abs().max().item() < 2.0 ** 15
    raw = r.view(torch.float16).view(3, 50, 3, 2, 32)       # (record, hi / lo, 32 fp16 pieces of x * scale)
    hi = raw[..., 0, :].double().reshape(3, 50, 96)
    lo = raw[..., 1, :].double().reshape(3, 50, 96)
    assert torch.equal(hi[..., :77], (x.double() * sc).to(torch.float16).double())
    assert (((hi + lo)[..., :77] / sc - x.double()).abs() <= 2.0 ** -22 * x.double().abs() + 2.0 ** -25 / sc).all()
    assert (hi[..., 77:] == 0).all() and (lo[..., 77:] == 0).all()
    for (b, M, N, K, ks) in ((1, 256, 128, 32, 1), (2, 300, 200, 96, 1), (3, 70, 530, 1000, 1), (2, 257, 129, 640, 3)):
        A, B = torch.randn(b, M, K, device=DEV), torch.randn(b, N, K, device=DEV)
        ref = 0.5 * torch.bmm(A.double(), B.double().transpose(1, 2))
        got = ops.gemm_rec_nt(ops.to_records(A), ops.to_records(B), 0.5, ksplit=ks)
        assert (got.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item(), (b, M, N, K, ks)
        got2 = ops.gemm_rec_nt(ops.to_records(A), ops.to_records(B), 0.5, ksplit=ks, out=got.clone(), accumulate=True)
        assert (got2.double() - 2 * ref).abs().max().item() < 4e-6 * ref.abs().max().item()
    for (b, K, M, N, ks) in ((1, 32, 256, 128, 1), (2, 100, 300, 200, 1), (1, 77, 64, 40, 1), (3, 1000, 530, 70, 2)):
        A, B = torch.randn(b, K, M, device=DEV), torch.randn(b, K, N, device=DEV)
        ref = 0.5 * torch.bmm(A.double().transpose(1, 2), B.double())
        got = ops.gemm_rec_tn(ops.to_records(A), ops.to_records(B), M, N, 0.5, ksplit=ks)
        assert (got.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item(), (b, K, M, N, ks)
    # explicit pitches: q and k as record slices of one [N][2D] tensor (the GMA attention call, core/gma.py)
    qk = torch.randn(2, 150, 256, device=DEV)
    qkr = ops.to_records(qk)
    out = torch.full((2, 150, 150), float("nan"), device=DEV)
    ops.gemm_rec_nt_raw(qkr.data_ptr(), 256, 150 * 256, qkr.data_ptr() + 4 * 128, 256, 150 * 256, out.data_ptr(), 150, 150 * 150, 2, 150, 150, 128, 0.25,
                        a_amax=ops.amax_of(qkr), b_amax=ops.amax_of(qkr))
    ref = 0.25 * torch.bmm(qk[..., :128].double(), qk[..., 128:].double().transpose(1, 2))
    assert (out.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    # operands far from 1: the amax words carry the range (bf16 pieces needed none; fp16 pieces would over- / underflow)
    for sa, sb in ((3e4, 2e-7), (1e-9, 5e3), (7e5, 1e4)):
        A, B = torch.randn(2, 200, 320, device=DEV) * sa, torch.randn(2, 130, 320, device=DEV) * sb
        ref = torch.bmm(A.double(), B.double().transpose(1, 2))
        got = ops.gemm_rec_nt(ops.to_records(A), ops.to_records(B))
        assert torch.isfinite(got).all() and (got.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item(), (sa, sb)


@pytest.mark.parametrize("B,H,W,nlev", [(2, 13, 22, 4), (1, 16, 24, 2), (1, 9, 33, 1), (2, 40, 48, 4)])
def test_tiled_row_volume_kernels_match_the_row_major_ones(B, H, W, nlev):
    """The tiled-row layout (csrc/corr_layout.hpp) end to end against round 1's row-major kernels, which the golden fixtures
    pin: build (fp32-operand and record kernels), lookup forward (coords and flow input), the one-pass gradient volume of a
    whole step of lookups (fp32 rows and records), and the build backward on both GEMM paths; pad cells of the rows are zero."""
    from flow_supervisor_amd import ops
    from flow_supervisor_amd.core.utils.utils import coords_grid
    torch.manual_seed(41)
    C, r, T = 64, 4, 3
    f1, f2 = torch.randn(B, C, H, W, device=DEV), torch.randn(B, C, H, W, device=DEV)
    levels4 = ops.corr_build(f1, f2, 4)                     # (the row-major kernels exist for four levels only)
    levels = levels4[:nlev]
    vol, lay = ops.corr_build_tiled(f1, f2, nlev)
    recs = (ops.fmap_records(f1), ops.fmap_records(f2))
    vol_r, _ = ops.corr_build_tiled(f1, f2, nlev, recs=recs)
    for l in range(nlev):
        close(lay.level_view(vol, l), levels[l], 2e-5, what=f"tiled build level {l}")
        close(lay.level_view(vol_r, l), levels[l], 2e-5, what=f"record build level {l}")
    nq = B * H * W
    valid = torch.zeros(lay.P, dtype=torch.bool, device=DEV)          # positions of a row that hold a cell of the reference pyramid
    for l in range(nlev):
        y, x = torch.meshgrid(torch.arange(lay.h[l], device=DEV), torch.arange(lay.w[l], device=DEV), indexing="ij")
        valid[lay.off[l] + ((y >> 2) * lay.tw[l] + (x >> 2)) * 16 + (y & 3) * 4 + (x & 3)] = True
    assert int(valid.sum()) == sum(h * w for h, w in zip(lay.h, lay.w))
    cells = valid.unsqueeze(0).expand(nq, lay.P)
    # (pad cells of the FORWARD volume are never read -- the lookup masks rows / columns beyond the floor sizes -- and are
    # not all written; the gradient volume's pad cells are contracted over by the backward GEMMs and must be zero: below)
    flows = [(torch.rand(B, 2, H, W, device=DEV) - 0.5) * 14 for _ in range(T)]
    coords = [coords_grid(B, H, W, device=DEV) + f for f in flows]
    for c, f in zip(coords, flows):
        ref = ops.corr_lookup_fwd(levels4, c, r, nhwc=True)[..., :nlev * 81].contiguous()
        close(ops.corr_lookup_tiled_fwd(vol, lay, c, r), ref, 1e-5, what="tiled lookup")
        close(ops.corr_lookup_tiled_fwd(vol, lay, f, r, is_flow=True), ref, 1e-5, what="tiled lookup, flow input")
    douts = [torch.randn(B, H, W, nlev * 81, device=DEV) for _ in range(T)]
    dlv = [torch.zeros_like(l) for l in levels4]
    for c, g in zip(coords, douts):
        g4 = torch.zeros(B, H, W, 4 * 81, device=DEV)
        g4[..., :nlev * 81] = g
        ops.corr_lookup_bwd_(dlv, c, g4, r, nhwc=True)
    dvol = ops.corr_dvol_build(douts, coords, lay, B, r)
    for l in range(nlev):
        close(lay.level_view(dvol, l), dlv[l], 2e-4, what=f"gradient volume level {l}")
    assert (dvol[~cells] == 0).all(), "pad cells of the gradient rows"
    dvol_f = ops.corr_dvol_build(douts, flows, lay, B, r, is_flow=True)
    close(dvol_f, dvol, 1e-6, what="gradient volume from flow input")
    d1o, d2o = ops.corr_build_bwd(f1, f2, [d.clone() for d in dlv])
    d1n, d2n = ops.corr_build_bwd_tiled(f1, f2, dvol, lay)
    dvol_r = ops.corr_dvol_build(douts, coords, lay, B, r, records=True)
    d1r, d2r = ops.corr_build_bwd_tiled(f1, f2, dvol_r, lay, records=True, f1r=recs[0])
    for got, ref, what in ((d1n, d1o, "dfmap1"), (d2n, d2o, "dfmap2"), (d1r, d1o, "dfmap1 (records)"), (d2r, d2o, "dfmap2 (records)")):
        assert ((got - ref).norm() / ref.norm()).item() < 5e-5, what


def test_frozen_batchnorm_fold_kernels():
    """fsraft_bn_fold / fsraft_bn_fold_bwd (csrc/norm_cl.hip): scale = w * rsqrt(rv + eps), shift = b - (rm - cbias) * scale and,
    from the partial sums [2][R][C] of the affine backward, dweight = rs * (S1 - rmc * S0), dbias = S0, dcbias = scale * S0."""
    from flow_supervisor_amd import _lib as L
    lib = L.load()
    torch.manual_seed(47)
    C, R = 96, 24
    w, b, rm, cb = (torch.randn(C, device=DEV) for _ in range(4))
    rv = torch.rand(C, device=DEV) + 0.1
    eps = 1e-5
    for cbias in (cb, None):
        out = torch.full((4, C), float("nan"), device=DEV)
        L.check(lib.fsraft_bn_fold(L.ptr(w), L.ptr(b), L.ptr(rm), L.ptr(rv), L.ptr(cbias) if cbias is not None else None, eps, C,
                                   L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), L.ptr(out[3]), L.stream()), "bn_fold")
        rs = torch.rsqrt(rv + eps)
        rmc = rm - (cbias if cbias is not None else 0)
        close(out[0], w * rs, 1e-5, what="scale")
        close(out[1], b - rmc * w * rs, 1e-5, what="shift")
        close(out[2], rs, 1e-5, what="rs")
        close(out[3], rmc, 1e-6, what="rmc")
        part = torch.randn(2, R, C, device=DEV)
        dpar = torch.full((3, C), float("nan"), device=DEV)
        L.check(lib.fsraft_bn_fold_bwd(L.ptr(part), R, C, L.ptr(out[2]), L.ptr(out[3]), L.ptr(out[0]), L.ptr(dpar[0]), L.ptr(dpar[1]),
                                       L.ptr(dpar[2]) if cbias is not None else None, L.stream()), "bn_fold_bwd")
        s0, s1 = part[0].sum(0), part[1].sum(0)
        close(dpar[0], rs * (s1 - rmc * s0), 1e-4, what="dweight")
        close(dpar[1], s0, 1e-4, what="dbias")
        if cbias is not None:
            close(dpar[2], w * rs * s0, 1e-4, what="dcbias")


def test_batched_pack_jobs_match_the_single_matrix_packer():
    """fsraft_pack_conv_weights (one launch per 16 matrices, parameters read in place) against fsraft_pack_conv_weight on
    torch-assembled weights: fused layers (cat along Cout), channel selections (cat of slices along Cin), the space-to-depth
    rewrite of a stride-2 weight (core/extractor.py::_s2d_weight), the fragment-order permutation (ops.fragment_order), fused
    biases, and the reverse direction (packed gradient -> parameter-shaped gradients, scaled)."""
    from flow_supervisor_amd import ops
    from flow_supervisor_amd.core.extractor import _s2d_weight, _s2d_weight_grad
    torch.manual_seed(23)
    wz, wr = torch.randn(40, 100, 1, 5, device=DEV), torch.randn(24, 100, 1, 5, device=DEV)
    sel, src_c = [(0, 36), (60, 100)], [36, 40]
    wcat = torch.cat([torch.cat([wz, wr], 0)[:, a:b] for a, b in sel], 1).contiguous()
    w64 = torch.randn(48, 64, 3, 3, device=DEV)
    w3 = torch.randn(20, 12, 3, 3, device=DEV)
    b1, b2 = torch.randn(40, device=DEV), torch.randn(24, device=DEV)
    plan = ops.PackPlan(DEV)
    hs = {}
    for mode in (0, 1, 10, 11):
        hs["cat", mode] = plan.pack([wz, wr], src_c, mode, srcOff=[a for a, _ in sel])
        hs["s2d", mode] = plan.pack([w3], [48], mode, cin_full=12, s2d=True)
        hs["plain", mode] = plan.pack([w64], [64], mode)
    hs["frag", 10] = plan.pack([w64], [64], 10, frag=True)
    hs["frag", 11] = plan.pack([w64], [64], 11, frag=True)
    hb = plan.bias([b1, b2])
    out = plan.run()
    for mode in (0, 1, 10, 11):
        assert torch.equal(out[hs["cat", mode]], ops.pack_weight(wcat, src_c, mode)), ("cat", mode)
        assert torch.equal(out[hs["s2d", mode]], ops.pack_weight(_s2d_weight(w3).contiguous(), [48], mode)), ("s2d", mode)
        assert torch.equal(out[hs["plain", mode]], ops.pack_weight(w64, [64], mode)), ("plain", mode)
    for mode in (10, 11):
        ref = ops.fragment_order(ops.pack_weight(w64, [64], mode))
        assert torch.equal(out[hs["frag", mode]].view(torch.int32).flatten(), ref.view(torch.int32).flatten()), ("frag", mode)
    assert torch.equal(out[hb], torch.cat([b1, b2]))
    # reverse: packed gradients -> parameter-shaped gradients
    gcat = torch.randn_like(out[hs["cat", 0]])
    gz, gr = torch.full_like(wz, float("nan")), torch.full_like(wr, float("nan"))
    g3p = torch.randn_like(out[hs["s2d", 0]])
    g3 = torch.full_like(w3, float("nan"))
    ops.unpack_weight_grads([(gcat, [gz, gr], src_c, [a for a, _ in sel], 100, 1, 5, 0.25, False),
                             (g3p, [g3], [48], [0], 12, 2, 2, 1.0, True)], DEV)
    ref = ops.unpack_weight_grad(gcat, tuple(wcat.shape), src_c) * 0.25
    full = torch.cat([gz, gr], 0)
    c = 0
    for a, b in sel:
        assert torch.equal(full[:, a:b], ref[:, c:c + b - a])
        c += b - a
    assert torch.isnan(full[:, 36:60]).all()             # channels no source covers are not touched
    assert torch.equal(g3, _s2d_weight_grad(ops.unpack_weight_grad(g3p, (20, 48, 2, 2), [48]), 12))


def test_weight_packs_follow_a_fused_optimizer_step():
    """`torch.optim.AdamW(fused=True)` updates parameters without bumping `Parameter._version`, which the GEMM-ready weight
    packs are keyed on (ops.parameters_updated): after two TrainStep steps the stepped model must predict exactly what a fresh
    model loaded from its state_dict predicts -- stale packs would still hold the initial weights."""
    import argparse
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.train import TrainStep
    torch.manual_seed(5)
    args = argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)
    model = RAFT(args).to(DEV).train()
    model.freeze_bn()
    step = TrainStep(model, lr=1e-3, iters=3)
    im1 = torch.rand(1, 3, 128, 192, device=DEV) * 255
    im2 = torch.rand(1, 3, 128, 192, device=DEV) * 255
    v0 = model.update_block.gru.convz1.weight._version
    w0 = model.update_block.gru.convz1.weight.detach().clone()
    for _ in range(2):
        step(im1, im2)
    assert model.update_block.gru.convz1.weight._version > v0
    assert not torch.equal(w0, model.update_block.gru.convz1.weight)
    fresh = RAFT(args).to(DEV).train()
    fresh.load_state_dict(model.state_dict())
    fresh.freeze_bn()
    with torch.no_grad():
        a = model(im1, im2, iters=3)[-1]
        b = fresh(im1, im2, iters=3)[-1]
    # (not bit-equal: the InstanceNorm statistics are summed with float atomics; stale packs give differences of order 1)
    assert (a - b).abs().max().item() < 1e-3, (a - b).abs().max().item()
    fresh.load_state_dict(RAFT(args).state_dict())
    with torch.no_grad():
        assert (fresh(im1, im2, iters=3)[-1] - b).abs().max().item() > 1e-2


@pytest.mark.parametrize("B,H,W,N", [(2, 440, 1024, 64), (1, 61, 75, 64), (3, 40, 70, 32), (1, 7, 9, 64), (2, 128, 192, 32)])
def test_stem_convolution_and_weight_gradient(B, H, W, N):
    """csrc/stem.hip: the encoders' 7x7 stride-2 stem (pytorch/core/extractor.py:135, :212) and its weight gradient against the
    fp64 convolution: odd sizes (partial tiles, clipped halo on every side), both output widths, bias."""
    from flow_supervisor_amd import ops
    torch.manual_seed(H * 7 + W)
    x = torch.rand(B, 3, H, W, device=DEV) * 2 - 1
    w = torch.randn(N, 3, 7, 7, device=DEV) * 0.1
    bias = torch.randn(N, device=DEV)
    y = ops.stem_fwd(x, w, bias)
    wd = w.double().requires_grad_()
    ref = torch.nn.functional.conv2d(x.double(), wd, bias.double(), 2, 3)
    close(y.permute(0, 3, 1, 2), ref.float(), 1e-6, what="stem forward")
    dy = torch.randn_like(y)
    ref.backward(dy.permute(0, 3, 1, 2).double())
    dw = ops.stem_wgrad(x, dy)
    close(dw, wd.grad.float(), 1e-6, rtol=3e-5, what="stem weight gradient")


def test_flat_adamw_matches_torch_adamw_with_clipping():
    """parallel.FlatAdamW (csrc/optim.hip: clip_grad_norm_ + AdamW as one kernel over flat buffers, pytorch/train.py:137, 280-282)
    against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW over six steps: parameters, clipped gradients, returned norm.
    Tensor sizes that are not multiples of four (the flat layout pads every tensor to 64 floats), a learning-rate change."""
    from flow_supervisor_amd.parallel import FlatAdamW, FlatGradients
    torch.manual_seed(3)
    shapes = [(64, 3, 7, 7), (2,), (17, 5), (1,), (256, 128, 3, 3), (96,), (33,)]
    ours = [torch.nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    grads = FlatGradients(ours)
    opt = FlatAdamW(grads, lr=3e-3, weight_decay=1e-2, eps=1e-8)
    topt = torch.optim.AdamW(ref, lr=3e-3, weight_decay=1e-2, eps=1e-8)
    for p, sh in zip(ours, shapes):
        assert tuple(p.shape) == sh and p.data_ptr() % 256 == 0
    for it in range(6):
        gs = [torch.randn(*sh, device=DEV) * (10.0 if it % 2 else 0.01) for sh in shapes]     # clipped / not clipped
        for p, q, g in zip(ours, ref, gs):
            grads.views[p].copy_(g)
            q.grad = g.clone()
        if it == 3:
            opt.set_lr(1e-3)
            for grp in topt.param_groups:
                grp["lr"] = 1e-3
        tn = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        topt.step()
        n = opt.step(clip=1.0)
        close(n, tn, 0.0, rtol=1e-6, what="gradient norm")
        for p, q in zip(ours, ref):
            close(p, q, 1e-7, rtol=1e-6, what=f"parameters after step {it}")
            close(grads.views[p], q.grad, 1e-9, rtol=1e-6, what="clipped gradient")


def test_flat_adamw_is_a_torch_optimizer():
    """the reference drives its optimizer with StepLR(optimizer, num_steps // 5, 0.5) + scheduler.step() and
    checkpoints it (pytorch/train.py:134-141, 283).  FlatAdamW must take a torch lr_scheduler, round-trip its state through
    state_dict() / load_state_dict(), leave parameters without a gradient (and their moments) alone like torch's AdamW
    skips `grad is None`, and refuse to step once a parameter was re-bound away from its flat buffer."""
    from flow_supervisor_amd.parallel import FlatAdamW, FlatGradients
    torch.manual_seed(4)
    shapes = [(32, 16, 3, 3), (32,), (7, 5), (130,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
    ours = mk()
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    grads = FlatGradients(ours)
    opt = FlatAdamW(grads, lr=2e-3, weight_decay=1e-2)
    topt = torch.optim.AdamW(ref, lr=2e-3, weight_decay=1e-2)
    assert isinstance(opt, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.StepLR(opt, 2, gamma=0.5)
    tsched = torch.optim.lr_scheduler.StepLR(topt, 2, gamma=0.5)

    def one_step(o, ps, gr, sc, skip=()):
        gs = [torch.randn(*sh, device=DEV) for sh in shapes]
        if o is opt:
            gr.begin()
            for i, (p, g) in enumerate(zip(ps, gs)):
                if i not in skip:
                    p.grad = g.clone()
            gr.finish()
            o.step()
        else:
            for i, (p, g) in enumerate(zip(ps, gs)):
                p.grad = None if i in skip else g.clone()
            o.step()
        sc.step()

    for it in range(5):
        skip = (1, 3) if it in (1, 2) else ()
        torch.manual_seed(100 + it); one_step(opt, ours, grads, sched, skip)
        torch.manual_seed(100 + it); one_step(topt, ref, None, tsched, skip)
        assert abs(sched.get_last_lr()[0] - tsched.get_last_lr()[0]) < 1e-12
        if it == 2:
            # torch counts steps per parameter; ours has one counter: parameters that skipped steps 1 and 2 differ from
            # torch's in their bias correction afterwards, so the comparison of THOSE stops here (untouched while skipped)
            for i in (1, 3):
                close(ours[i], ref[i], 1e-7, rtol=1e-6, what="a parameter without gradient is left alone")
        for i, (p, q) in enumerate(zip(ours, ref)):
            if i in (1, 3) and it >= 3:
                continue
            close(p, q, 1e-7, rtol=2e-6, what=f"parameter {i} after step {it} (StepLR lr {sched.get_last_lr()[0]:g})")

    # checkpoint / resume: a fresh optimizer loaded from state_dict() continues identically
    sd = opt.state_dict()
    twins = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    g2 = FlatGradients(twins)
    opt2 = FlatAdamW(g2, lr=1.0)
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"] and float(opt2.lr) == float(opt.lr)
    gs = [torch.randn(*sh, device=DEV) for sh in shapes]
    for o, gr, ps in ((opt, grads, ours), (opt2, g2, twins)):
        gr.begin()
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        gr.finish()
        o.step(clip=1.0)
    for p, q in zip(ours, twins):
        assert torch.equal(p, q), "resumed optimizer diverged"
    with pytest.raises(ValueError):
        FlatAdamW(FlatGradients(mk()[:2])).load_state_dict(sd)

    # a re-bound parameter (model.float() / load_state_dict(assign=True) style) must not be trained silently
    ours[0].data = ours[0].data.clone()
    with pytest.raises(RuntimeError, match="no longer lives"):
        opt.step()


def test_hipgraph_replays_of_the_train_step_follow_the_eager_steps():
    """bench.py times hipGraph replays of the whole train step (one rank).  Replays reuse every buffer of the capture, so
    anything zeroed "once" or by a node the graph drops shows up from the second replay on: ops._ZeroPool (chunks filled once
    per capture, not once per process) and the split-K record GEMMs (their zero fill is a kernel, the hipMemsetAsync node was
    not replayed).  Six replays against six eager steps from the same start: the losses must follow each other."""
    import argparse
    import copy
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.train import TrainStep
    torch.manual_seed(0)
    model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(DEV).train()
    model.freeze_bn()
    twin = copy.deepcopy(model)
    g = torch.Generator(device=DEV).manual_seed(1)
    im1 = torch.rand(2, 3, 184, 320, device=DEV, generator=g) * 255
    im2 = torch.rand(2, 3, 184, 320, device=DEV, generator=g) * 255
    eager = TrainStep(twin, lr=1e-4, iters=4, capturable=True)
    le = [float(eager(im1, im2)) for _ in range(8)]
    step = TrainStep(model, lr=1e-4, iters=4, capturable=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step(im1, im2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        loss = step(im1, im2)
    lg = []
    for _ in range(6):
        graph.replay()
        torch.cuda.synchronize()
        lg.append(float(loss))
    del graph
    # (the first steps of a random-init model on random images move the loss by factors -- 1.2, 12.7, 9.4, 5.1, 1.5, 2.8 ... --
    #  so a wrong update shows as a different sequence, while summation-order noise stays below 1e-3 relative)
    for a, b in zip(le[2:], lg):
        assert abs(a - b) <= 2e-3 * abs(a), (le, lg)


def test_step_replayed_as_two_hipgraphs_with_the_exchange_between_them():
    """bench.py at N > 1: forward + loss + backward as one hipGraph, the all-reduce of the flat gradient buffer issued eagerly,
    clip + AdamW + re-pack as a second hipGraph (TrainStep.forward_backward / exchange / update).  At world size 1 the exchange
    is the identity, so six replayed steps must follow six eager steps of the plain __call__ from the same start."""
    import argparse
    import copy
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.train import TrainStep
    torch.manual_seed(0)
    model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(DEV).train()
    model.freeze_bn()
    twin = copy.deepcopy(model)
    g = torch.Generator(device=DEV).manual_seed(1)
    im1 = torch.rand(2, 3, 184, 320, device=DEV, generator=g) * 255
    im2 = torch.rand(2, 3, 184, 320, device=DEV, generator=g) * 255
    eager = TrainStep(twin, lr=1e-4, iters=4, capturable=True)
    le = [float(eager(im1, im2)) for _ in range(8)]
    step = TrainStep(model, lr=1e-4, iters=4, capturable=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step(im1, im2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g_fb, g_up = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_fb, stream=side):
        loss = step.forward_backward(im1, im2)
    step.exchange()
    with torch.cuda.graph(g_up, stream=side, pool=g_fb.pool()):
        step.update()
    lg = []
    for _ in range(6):
        g_fb.replay()
        step.exchange()
        g_up.replay()
        torch.cuda.synchronize()
        lg.append(float(loss))
    del g_fb, g_up
    for a, b in zip(le[2:], lg):
        assert abs(a - b) <= 2e-3 * abs(a), (le, lg)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus 2 ...` typed as is, no torchrun and no WORLD_SIZE around it, must start its
    two ranks itself (children, before the parent touches the GPU), run the data-parallel step on both and print ONE JSON line
    with n_gpus = 2.  On a one-GPU box the ranks share cuda:0 and exchange over gloo through host memory; on a box with two
    devices the same command runs on RCCL.  The line must prove that the collective saw both ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--height", "128",
                        "--width", "192", "--iters", "3", "--batch-per-gpu", "1", "--no-cpu-baseline", "--no-extra"],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    print("bench --gpus 2:", {k: out[k] for k in ("value", "n_gpus", "ms_per_step", "rccl")})
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2
    assert out["rccl"]["world_size"] == 2 and out["rccl"]["ranks_seen_by_all_reduce"] == 2
    assert out["rccl"]["graph"] == "captured", out["rccl"]
    assert out["value"] > 0 and out["config"]["loss"] == out["config"]["loss"]     # finite loss on the replayed steps
    # the self-diagnosing part of the line: one entry per rank for the wall time and for each of the three parts
    # of the two-graph route, and they add up to the step
    for k in ("per_rank_ms_per_step", "per_rank_graph_fb_ms", "per_rank_exchange_ms", "per_rank_graph_up_ms"):
        assert len(out["rccl"][k]) == 2 and all(v > 0 for v in out["rccl"][k]), (k, out["rccl"])
    parts = [sum(out["rccl"][k][r] for k in ("per_rank_graph_fb_ms", "per_rank_exchange_ms", "per_rank_graph_up_ms")) for r in range(2)]
    assert all(p <= 1.15 * max(out["rccl"]["per_rank_ms_per_step"]) for p in parts), (parts, out["rccl"])


def test_bench_rank_that_cannot_rendezvous_exits_with_a_message(tmp_path):
    """the first multi-GPU run must not hang.  A rank whose peers never arrive leaves the rendezvous after
    FSRAFT_DIST_TIMEOUT_S with exit code 3 and one line on stderr that names the rank and the rendezvous address."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               FSRAFT_DIST_TIMEOUT_S="8", FSRAFT_BENCH_SHARED_GPUS="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--height", "128",
                        "--width", "192", "--iters", "2", "--batch-per-gpu", "1", "--no-cpu-baseline", "--no-extra"],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert "could not join the process group" in r.stderr and "rank 1" in r.stderr, r.stderr[-1500:]


def test_semi_step_reads_inputs_refreshed_in_place():
    """SemiTrainStep cached the concatenation of the labelled and the unlabelled inputs keyed on id() alone; a loop
    that refreshes preallocated input buffers in place (the pattern of hipGraph replays) trained on the first batch forever.
    Eager: the second step on refreshed buffers must see the new data; captured: the replay must re-read the buffers."""
    import argparse
    from flow_supervisor_amd.core.l2l import L2L
    from flow_supervisor_amd.train import SemiTrainStep
    torch.manual_seed(0)
    model = L2L(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(DEV).train()
    model.freeze_bn()
    g = torch.Generator(device=DEV).manual_seed(3)
    H, W, h, w = 128, 192, 96, 128

    def sample(oy, ox):
        f1 = torch.rand(1, 3, H, W, device=DEV, generator=g) * 255
        f2 = torch.rand(1, 3, H, W, device=DEV, generator=g) * 255
        return [f1[:, :, oy:oy + h, ox:ox + w].contiguous(), f2[:, :, oy:oy + h, ox:ox + w].contiguous(), f1, f2, ox, oy,
                torch.randn(1, 2, h, w, device=DEV, generator=g), torch.ones(1, h, w, device=DEV)]

    sup, unsup = sample(8, 16), sample(16, 32)
    fresh = [torch.rand_like(t) * 255 for t in sup[:4]]
    step = SemiTrainStep(model, lr=0.0, iters=2, capturable=True)       # lr 0: the weights stay, only the data moves the loss
    l0 = float(step(sup, unsup)[0])
    assert abs(float(step(sup, unsup)[0]) - l0) <= 1e-4 * abs(l0)
    keep = [t.clone() for t in sup[:4]]
    for t, f in zip(sup[:4], fresh):
        t.copy_(f)
    l1 = float(step(sup, unsup)[0])
    assert abs(l1 - l0) > 1e-3 * abs(l0), (l0, l1)
    # captured: replays follow the buffers
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(sup, unsup)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        ls, _ = step(sup, unsup)
    graph.replay(); torch.cuda.synchronize()
    assert abs(float(ls) - l1) <= 2e-3 * abs(l1), (float(ls), l1)
    for t, k in zip(sup[:4], keep):
        t.copy_(k)
    graph.replay(); torch.cuda.synchronize()
    assert abs(float(ls) - l0) <= 2e-3 * abs(l0), (float(ls), l0)
    del graph


# ----------------------------------------------------------------------------- data parallelism on the real step (row e)
@pytest.mark.parametrize("global_batch,H,W,iters", [(4, 128, 192, 3), (3, 128, 192, 3), (2, 440, 1024, 12)])
def test_two_process_train_step_matches_single_process(global_batch, H, W, iters, tmp_path):
    """Two fresh processes (tests/_dp_worker.py), each running the real TrainStep on its shard of `global_batch` pairs at
    128x192 x 3 iterations and exchanging the flat gradient (gloo staged through the host: both ranks sit on cuda:0),
    against one process on the whole batch: reduced + clipped flat gradient and post-AdamW weights.  global_batch = 3
    gives shards of 2 and 1 (gradients weighted by local / global batch); the third case is the benchmark's own shape and
    iteration count with one pair per rank."""
    import os
    import socket
    import subprocess
    import sys
    from flow_supervisor_amd.train import TrainStep
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    out = str(tmp_path / "r0.pt")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(global_batch), out, str(H), str(W), str(iters)])
             for r in range(2)]
    rcs = [p.wait(timeout=900) for p in procs]
    assert rcs == [0, 0], rcs
    got = torch.load(out)
    m = _model(False, 650).train()
    m.freeze_bn()
    im1, im2 = (t.to(DEV) for t in synthetic_pair(global_batch, H, W, 651))
    step = TrainStep(m, lr=1e-4, iters=iters)
    loss = step(im1, im2)
    ref_g = step.grads.flat.cpu()
    ref_p = torch.cat([p.detach().reshape(-1).cpu() for p in step.grads.params])
    rel_g = float((got["flat"] - ref_g).norm() / ref_g.norm())
    rel_p = float((got["params"] - ref_p).norm() / ref_p.norm())
    print("dp2 vs single: grad rel", rel_g, "param rel", rel_p, "loss(rank 0 shard)", got["loss"], "loss(all)", float(loss))
    assert rel_g <= 2e-3, rel_g          # split-bf16 products + a different summation order over the batch
    assert rel_p <= 2e-4, rel_p          # one AdamW step of lr 1e-4: where a gradient is ~0 its sign, hence the update, can differ


def test_rccl_exchange_at_world_size_one(tmp_path):
    """VERDICT r2 next #4 / ADVICE r2: the asynchronous bucket all-reduces issued from the backward hooks had only ever run
    through gloo.  A fresh process (tests/_rccl_worker.py) initialises the nccl (= RCCL) backend with one rank, forces the
    collectives on (FSRAFT_DP_FORCE_COLLECTIVE=1) and runs three real TrainSteps: RCCL's stream ordering against the
    hook-time copies and against clip + AdamW is what N > 1 ranks execute, and a sum over one rank is the identity, so
    gradients and weights must follow the no-collective run (to the run-to-run noise of the atomics in the weight
    gradients).  What one rank cannot show is a data race that only corrupts values when a peer contributes; what it does
    show is that the API sequence (async work handles from autograd's hook thread, wait() before the optimizer, capture)
    runs on RCCL.  The same worker captures a step WITH the
    collectives in a hipGraph and replays it: bench.py enables graphs at N > 1 only because this passes (if RCCL refuses
    capture on some stack the worker records the failure mode and the test reports it)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    out = str(tmp_path / "rccl.json")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    rc = subprocess.run([sys.executable, worker, str(port), out], env=env, timeout=900).returncode
    assert rc == 0, rc
    res = json.load(open(out))
    print("rccl world-1:", json.dumps(res))
    assert res["backend"] == "nccl"
    # (not bit-equal even without a collective: the weight gradients add with fp32 atomics, so two runs of the same three
    #  steps differ in the last bits and AdamW amplifies sign flips of ~0 gradients)
    assert res["eager"]["grad_rel"] <= 5e-3 and res["eager"]["param_rel"] <= 5e-4, res["eager"]
    assert all(abs(a - b) <= 1e-3 * abs(a) for a, b in zip(res["eager"]["losses_plain"], res["eager"]["losses_rccl"])), res["eager"]
    assert res["eager_unbucketed"]["grad_rel"] <= 5e-3 and res["eager_unbucketed"]["param_rel"] <= 5e-4, res["eager_unbucketed"]
    log = os.environ.get("FSRAFT_RCCL_LOG")
    if log:
        with open(log, "w") as f:
            json.dump(res, f, indent=1)
    if res["graph"]["ok"]:
        # replays re-run the same kernels on the same buffers; atomics in the weight gradients reorder sums
        assert res["graph"]["param_rel_vs_eager"] <= 5e-4, res["graph"]
        assert all(r <= 3e-3 for r in res["graph"]["loss_rel_vs_eager"]), res["graph"]
    else:
        pytest.xfail("hipGraph capture of a step containing RCCL all-reduces failed: " + res["graph"]["error"])


def test_nchw_entry_does_not_reuse_context_of_a_freed_tensor():
    """BasicUpdateBlock.forward (the NCHW drop-in entry INTEGRATION.md hands to the reference's raft.py)
    caches the channels-last copy of `inp`.  Under no_grad every pair's `inp = relu(...)` is a fresh tensor that the caching
    allocator places at the address of the previous pair's (freed) one: the cache must key on identity, not address."""
    from flow_supervisor_amd.core.update import BasicUpdateBlock
    sh = shapes("update_basic")
    blk = BasicUpdateBlock(ns(False), hidden_dim=128)
    blk.load_state_dict(procedural_state_dict(sh, 300))
    blk = blk.to(DEV).eval()
    sd = {"update_block." + k: v for k, v in blk.state_dict().items()}
    B, H, W = 1, 12, 16
    net = torch.tanh(rand_tensor((B, 128, H, W), 310)).to(DEV)
    corr = rand_tensor((B, 324, H, W), 312, 2.0).to(DEV)
    flow = rand_tensor((B, 2, H, W), 313, 3.0).to(DEV)
    outs, ptrs = [], []
    with torch.no_grad():
        for seed in (311, 411):
            inp = torch.relu(rand_tensor((B, 128, H, W), seed).to(DEV))      # fresh tensor, version 0, same shape
            ptrs.append(inp.data_ptr())
            n2, mask, delta = blk(net, inp, corr, flow)
            outs.append((n2.cpu(), delta.cpu(), inp.cpu()))
            del inp, n2, mask, delta
    for n2, delta, inp in outs:
        rn, _, rd = O.basic_update_block({k: v.cpu() for k, v in sd.items()}, "update_block.", net.cpu(), inp, corr.cpu(), flow.cpu())
        close(n2, rn, 2e-4, what="net (pair %d)" % len(ptrs)); close(delta, rd, 2e-4, what="delta")
    print("inp addresses of the two pairs:", ptrs, "(equal = the allocator reused the block)")


# ---------------------------------------------------------------- several host threads (the reference's nn.DataParallel caller)
def _small_grid_conv(seed, B=1, H=46, W=96, cs=(256,), N=126, kh=3, kw=3):
    """One small-grid convolution (fewer tiles than CUs: the split-K route with its scratch buffer) with seeded operands."""
    from flow_supervisor_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    srcs = [torch.randn(B, H, W, c, generator=g).to(DEV) for c in cs]
    w = (torch.randn(N, sum(cs), kh, kw, generator=g) * 0.05).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    packs = (ops.pack_weight(w, list(cs), 0), ops.pack_weight(w, list(cs), 10))

    def run():
        out = torch.full((B, H, W, N), float("nan"), device=DEV)
        ops.conv_forward([ops.V(t, c) for t, c in zip(srcs, cs)], packs[0], bias, B, H, W, kh, kw, N, [ops.Dst.nhwc(out)], relu=True,
                         wpk_split=packs[1])
        return out
    return run


def test_two_host_threads_on_two_streams_use_their_own_split_k_scratch():
    """VERDICT r4 next #5 / SURVEY 8b "Threading": the reference's multi-GPU caller is nn.DataParallel (pytorch/train.py:192) --
    one host thread per replica, all inside one process -- and ctypes releases the GIL around every libfsraft call, so two threads'
    calls interleave freely.  Round 4 registered ONE process-wide split-K scratch pointer (fsraft_conv_workspace) and switched it
    on the host: thread A's launch could pick up thread B's buffer.  Now the buffer travels in each call's descriptor.  Two threads,
    each on its own stream, run different small-grid convolutions (the route that parks partial tiles in the scratch) 40 times
    each, concurrently; every result must be bit-equal to the same convolution run alone."""
    import threading
    from flow_supervisor_amd import ops
    runs = [_small_grid_conv(11), _small_grid_conv(12, H=47, W=156, cs=(128, 128), N=128, kh=1, kw=5)]
    alone = [r() for r in runs]
    torch.cuda.synchronize()
    errors, results = [], [None, None]
    barrier = threading.Barrier(2)

    def worker(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                barrier.wait()
                outs = [runs[i]() for _ in range(40)]
                s.synchronize()
            results[i] = outs
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    keys = [k for k in ops._CONV_WS if k[0] == torch.cuda.current_device()]
    assert len(keys) >= 3, keys                  # the main thread's stream and one per worker: nothing shared
    for i in range(2):
        for k, o in enumerate(results[i]):
            assert torch.equal(o, alone[i]), (i, k, float((o - alone[i]).abs().max()))


def test_legacy_workspace_registration_is_per_thread():
    """fsraft_conv_workspace (kept for bindings written against the round-3 header) registers a buffer for the CALLING THREAD only:
    a descriptor without `ws` enqueued by another thread must not touch it (that thread simply gets no split-K route), while the
    registering thread's own call does use it.  Checked through a sentinel pattern in the buffer."""
    import ctypes
    import threading
    from flow_supervisor_amd import _lib, ops
    lib = _lib.load()
    run = _small_grid_conv(13)
    ref = run()
    ws = torch.full((ops.CONV_WS_FLOATS,), -7.0, device=DEV)
    saved = dict(ops._CONV_WS)
    real = ops._conv_workspace
    ops._conv_workspace = lambda device, pixels: None          # descriptors without ws: the registration decides
    try:
        assert lib.fsraft_conv_workspace(ctypes.c_void_p(ws.data_ptr()), ws.numel()) == 0       # this (main) thread
        box = {}

        def other():
            box["out"] = run()
            torch.cuda.synchronize()
        t = threading.Thread(target=other)
        t.start(); t.join()
        assert bool((ws == -7.0).all()), "another thread's launch wrote into this thread's registered scratch"
        close(box["out"], ref, 2e-5, what="convolution without a scratch buffer (no split-K route)")
        mine = run()
        torch.cuda.synchronize()
        assert not bool((ws == -7.0).all()), "the registering thread's own call did not use its scratch"
        assert torch.equal(mine, ref)
    finally:
        lib.fsraft_conv_workspace(ctypes.c_void_p(0), 0)
        ops._conv_workspace = real
        ops._CONV_WS.clear(); ops._CONV_WS.update(saved)


def test_two_host_threads_train_two_models_concurrently():
    """The whole path from two host threads at once (one model replica and one stream per thread, as a DataParallel-style caller
    would drive two devices; here both on the box's one GPU): forward + loss + backward of a small RAFT, three steps each, must give
    the losses and gradients of the same steps run one thread after the other."""
    import threading
    from flow_supervisor_amd.train import raft_sequence_loss

    def steps(seed, out):
        m = _model(False, seed).train()
        m.freeze_bn()
        im1, im2 = (t.to(DEV) for t in synthetic_pair(1, 128, 192, seed + 1))
        for _ in range(3):
            for p in m.parameters():
                p.grad = None
            loss = raft_sequence_loss(m(im1, im2, iters=3))
            loss.backward()
        torch.cuda.current_stream().synchronize()
        out.append((float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))

    serial = [[], []]
    for i in range(2):
        steps(50 + 10 * i, serial[i])
    errors, conc = [], [[], []]

    def worker(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                steps(50 + 10 * i, conc[i])
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for i in range(2):
        (l0, g0), (l1, g1) = serial[i][0], conc[i][0]
        assert abs(l0 - l1) <= 2e-5 * abs(l0), (l0, l1)
        for k in g0:
            if k.startswith("fnet."):
                continue            # (atomically accumulated InstanceNorm statistics: run-to-run noise of its own, TRAIN_TOL)
            close(g1[k], g0[k], 1e-6, rtol=2e-3, what=f"thread {i} {k}")


def test_ops_refuse_tensors_of_another_device():
    """"use the current device + current stream, re-entrant across devices".  The kernels are enqueued on the current
    device's current stream, so tensors living on another device are refused with a RuntimeError (require_cuda_f32) instead of
    being dereferenced by the wrong GPU; the module-level entry points make their input's device current themselves
    (_lib.on_tensor_device).  Needs two devices for the cross-device half."""
    from flow_supervisor_amd import _lib, ops
    from flow_supervisor_amd.core.corr import CorrBlock
    a = torch.randn(1, 64, 8, 12, device=DEV)
    _lib.require_cuda_f32(a, None, a)
    with pytest.raises(RuntimeError):
        _lib.require_cuda_f32(a, a.cpu())
    if torch.cuda.device_count() < 2:
        pytest.skip("one device: the cross-device refusal needs two")
    b = a.to("cuda:1")
    with pytest.raises(RuntimeError, match="one device"):
        _lib.require_cuda_f32(a, b)
    with pytest.raises(RuntimeError, match="current device"):
        ops.to_records(b.permute(0, 2, 3, 1).contiguous())          # current device is cuda:0
    blk = CorrBlock(b, b)                                             # the entry point switches to the tensors' device
    out = blk(torch.zeros(1, 2, 8, 12, device="cuda:1"))
    ref = CorrBlock(a, a)(torch.zeros(1, 2, 8, 12, device=DEV))
    assert out.device == b.device and torch.cuda.current_device() == 0
    close(out.cpu(), ref.cpu(), 1e-6, what="CorrBlock on cuda:1 from a thread whose current device is cuda:0")


def test_reference_shaped_shell_matches_the_package_shell():
    """INTEGRATION.md section 1 / `bench.py --variant dropin`: the reference's own model shell (core/raft_dropin.py restates
    pytorch/core/raft.py:99-144: NCHW tensors, `corr_fn(coords1)` -> `update_block(net, inp, corr, flow)` -> `upsample_flow` every
    iteration, `coords1` carried and detached) over the swapped blocks must give the predictions and parameter gradients of this
    package's own shell (flow-carrying channels-last loop, once-per-step head / motion-encoder batches, second stream) -- and both are
    held to the reference-generated fixture by the train-step goldens."""
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.core.raft_dropin import ReferenceShapedRAFT
    from flow_supervisor_amd.train import raft_sequence_loss
    ns = argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)
    a = RAFT(ns).to(DEV).train()
    b = ReferenceShapedRAFT(ns).to(DEV).train()
    b.load_state_dict(a.state_dict())
    a.freeze_bn(); b.freeze_bn()
    im1, im2 = (t.to(DEV) for t in synthetic_pair(2, 128, 192, 33))
    outs = []
    for m in (a, b):
        preds = m(im1, im2, iters=4)
        raft_sequence_loss(preds).backward()
        outs.append((preds, {k: p.grad for k, p in m.named_parameters() if p.grad is not None}))
    (pa, ga), (pb, gb) = outs
    assert len(pa) == len(pb) == 4 and pb[0].shape == (2, 2, 128, 192)
    for i in range(4):
        close(pb[i], pa[i], 2e-4, what=f"prediction {i}")
    assert set(ga) == set(gb)
    for k in ga:
        if k.startswith("fnet."):
            continue                # (its own run-to-run noise: atomically accumulated InstanceNorm statistics, TRAIN_TOL)
        close(gb[k], ga[k], 1e-6, rtol=3e-3, what=f"gradient {k}")
    # test_mode contract of the shell: (flow at 1/8 resolution, last upsampled flow)
    with torch.no_grad():
        lo, up = b.eval()(im1, im2, iters=3, test_mode=True)
    assert lo.shape == (2, 2, 16, 24) and up.shape == (2, 2, 128, 192)


def test_tf_twins_train_through_volume_pyramid_and_lookup():
    """VERDICT r4 missing #4: the TF tree reuses the transposed volume under a GradientTape (raft/semi.py:198-303: forward pyramid from
    calc_all_field, backward-flow pyramid from build_pyramid(transpose(volume)), a lookup on each, gradients into both feature maps).
    Round 4's twins were forward-only.  Floor-sized pyramids (every pooled size even) are differentiable now: the gradients of a
    scalar through calc_all_field -> CorrBlock lookup and through transpose_volume -> build_pyramid -> lookup must match torch
    autograd on the plain-torch restatement (oracle.corr_pyramid / corr_lookup on CPU, fp32).  'SAME' pyramids stay forward-only."""
    from flow_supervisor_amd import raft_tf
    torch.manual_seed(7)
    B, C, H, W = 2, 64, 16, 24
    f1c, f2c = torch.randn(B, H, W, C), torch.randn(B, H, W, C)
    coords = (O.coords_grid(B, H, W) + (torch.rand(B, 2, H, W) - 0.5) * 6).permute(0, 2, 3, 1).contiguous()
    wf, wb = torch.randn(B, H, W, 324), torch.randn(B, H, W, 324)

    def run(f1, f2, dev):
        if dev == "cpu":          # restatement in plain torch: the reference's own ops (matmul, avg_pool2d, grid_sample semantics)
            a, b = f1.permute(0, 3, 1, 2), f2.permute(0, 3, 1, 2)
            pyr = O.corr_pyramid(a, b, 4)
            fw = O.corr_lookup(pyr, coords.permute(0, 3, 1, 2), 4).permute(0, 2, 3, 1)
            vt = pyr[0].view(B, H, W, H, W).permute(0, 3, 4, 1, 2).reshape(B * H * W, 1, H, W)
            bpyr = [vt]
            for _ in range(3):
                bpyr.append(torch.nn.functional.avg_pool2d(bpyr[-1], 2, 2))
            bw = O.corr_lookup(bpyr, coords.permute(0, 3, 1, 2), 4).permute(0, 2, 3, 1)
            return (fw * wf).sum() + (bw * wb).sum()
        pyr = raft_tf.calc_all_field(f1, f2, num_pool=3)
        look = raft_tf.CorrBlock(4, 4)
        fw = look(pyr, coords.to(dev))
        bpyr = raft_tf.build_pyramid(raft_tf.transpose_volume(pyr[0]), num_pool=3)
        bw = look(bpyr, coords.to(dev))
        return (fw * wf.to(dev)).sum() + (bw * wb.to(dev)).sum()

    grads = {}
    for dev in ("cpu", DEV):
        a, b = f1c.clone().to(dev).requires_grad_(True), f2c.clone().to(dev).requires_grad_(True)
        loss = run(a, b, dev)
        loss.backward()
        grads[dev] = (float(loss), a.grad.cpu(), b.grad.cpu())
    assert abs(grads[DEV][0] - grads["cpu"][0]) <= 2e-5 * abs(grads["cpu"][0]) + 1e-2
    close(grads[DEV][1], grads["cpu"][1], 1e-4, rtol=2e-5, what="d / d feature map 1 through both pyramids")
    close(grads[DEV][2], grads["cpu"][2], 1e-4, rtol=2e-5, what="d / d feature map 2 through both pyramids")
    # odd pooled sizes ('SAME' pooling: TF-only semantics) remain forward-only and say so
    g1 = torch.randn(1, 22, 24, 32, device=DEV, requires_grad=True)
    with pytest.raises(RuntimeError, match="forward-only"):
        raft_tf.calc_all_field(g1, g1, num_pool=3)


@pytest.mark.parametrize("kind", ["basic", "small", "alt", "gma", "l2l"])
def test_amax_words_bound_their_tensors(kind):
    """Round 6: every GEMM-shaped kernel scales its operands from amax words, and a buffer CARRIES a word only if every kernel
    writing it raises the word (ops.tracked).  A writer that forgot would leave the word too low and the fp16 pieces would
    overflow some day; this runs a train step of every model family in audit mode (ops.AMAX_AUDIT: each carried word is
    compared with the tensor's true maximum before the kernel that reads it) and on weights / images scaled far from 1, where
    a missing scale cannot hide."""
    from flow_supervisor_amd import ops
    from flow_supervisor_amd.core.l2l import L2L
    from flow_supervisor_amd.core.raft import RAFT
    old = ops.AMAX_AUDIT
    ops.AMAX_AUDIT = True
    try:
        ops.set_arithmetic(True)
        seed = 3
        if kind == "gma":
            m = _gma_model(seed).train()
        elif kind == "l2l":
            m = L2L(ns(False))
            m.load_state_dict(procedural_state_dict(shapes("l2l_basic"), seed))
            m = m.to(DEV).train()
        else:
            a = ns(kind == "small")
            a.alternate_corr = kind == "alt"
            m = RAFT(a)
            m.load_state_dict(procedural_state_dict(shapes("raft_small" if kind == "small" else "raft_basic"), seed))
            m = m.to(DEV).train()
        m.freeze_bn()
        B, H, W = 2, 128, 192
        im1, im2 = (t.to(DEV) for t in synthetic_pair(B, H, W, seed + 1))
        for scale in (1.0, 3e3):          # the second pass: update-block weights x 3e3 -> activations and gradients far outside fp16's own range
            if scale != 1.0:
                with torch.no_grad():
                    for n_, p in m.named_parameters():
                        if "update_block" in n_ and n_.endswith("weight") and ("convc1" in n_ or "convf1" in n_ or "flow_head.conv2" in n_):
                            p.mul_(scale)
            if kind == "l2l":
                preds = m(im1[:, :, 8:104, 16:144].contiguous(), im2[:, :, 8:104, 16:144].contiguous(), im1, im2,
                          torch.tensor([16] * B), torch.tensor([8] * B), iters=4)
            else:
                preds = m(im1, im2, iters=3)
            loss = O.sequence_loss_zero_gt(preds)
            loss.backward()
            assert torch.isfinite(loss), (kind, scale)
            assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None), (kind, scale)
            m.zero_grad(set_to_none=True)
    finally:
        ops.AMAX_AUDIT = old
        ops.set_arithmetic(True)
