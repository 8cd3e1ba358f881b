"""GPU parity tests, SURVEY.md section 8 rows a1-a5 (+ the TF-shaped twins of the same path): all-pairs volume, pyramid, lookups, alt-corr, their backward, the record GEMMs.
(Split out of the former tests/test_gpu_parity.py in round 6; shared helpers, fixtures and the ONE tolerance table live in
tests/_gpu_common.py.)"""
import pytest

from _gpu_common import *      # noqa: F401,F403  (helpers, fixtures, tolerance table)

pytestmark = pytest.mark.gpu


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
    # (round 6: the reference-shaped result is an NCHW-shaped VIEW of the channels-last lookup, update.NCHW_VIEWS -- same shape and values
    #  as corr.py:50's `.permute(0, 3, 1, 2).contiguous()`, torch.channels_last memory format)
    assert out.shape == (B, 4 * (2 * r + 1) ** 2, H, W) and (out.is_contiguous() or out.is_contiguous(memory_format=torch.channels_last))
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


@pytest.mark.parametrize("sigma,rough", [(0.3, 0), (8.0, 1)])
def test_alternate_corr_block_follows_the_flow_regime(sigma, rough):
    """VERDICT r5 next #5 at the block level, KITTI feature shape (47 x 156, C = 256): three lookups of an AlternateCorrBlock on smooth /
    rough coordinates (N(0, sigma cells) around a constant motion) -- the block's per-launch dispatch sends them to the matrix-pipe
    kernel / the fp32 tile kernel -- against CorrBlock's volume path on the same inputs: lookups and, through one loss over all three,
    the feature-map gradients."""
    from flow_supervisor_amd.core.corr import AlternateCorrBlock, CorrBlock
    from flow_supervisor_amd.core.utils.utils import coords_grid
    B, C, H, W = 1, 256, 47, 156
    g = torch.Generator().manual_seed(int(sigma * 10))
    base = coords_grid(B, H, W, device="cpu") + torch.tensor([3.3, -1.7]).view(1, 2, 1, 1)
    coords = [(base + sigma * torch.randn(B, 2, H, W, generator=g)).to(DEV) for _ in range(3)]
    ups = [rand_tensor((B, 324, H, W), 40 + i).to(DEV) for i in range(3)]
    res = {}
    for kind, cls in (("alt", AlternateCorrBlock), ("vol", CorrBlock)):
        f1 = rand_tensor((B, C, H, W), 21).to(DEV).requires_grad_(True)
        f2 = rand_tensor((B, C, H, W), 22).to(DEV).requires_grad_(True)
        blk = cls(f1, f2, num_levels=4, radius=4)
        outs = [blk(c) for c in coords]
        if kind == "alt":
            assert blk._regime is not None and int(blk._regime[0].item()) == rough, blk._regime.tolist()
        sum((o * u).sum() for o, u in zip(outs, ups)).backward()
        res[kind] = ([o.detach() for o in outs], f1.grad, f2.grad)
    for i in range(3):
        close(res["alt"][0][i], res["vol"][0][i], 1e-4, what=f"lookup {i}, sigma {sigma}")
    close(res["alt"][1], res["vol"][1], 3e-4, what=f"dfmap1, sigma {sigma}")
    close(res["alt"][2], res["vol"][2], 3e-4, what=f"dfmap2, sigma {sigma}")


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
    split-arithmetic GEMM on records) against the oracle's lookup of the dense volume (what AlternateCorrBlock must equal,
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
        # the per-launch dispatch: whichever kernel the flow statistic names, the result is that kernel's, bit for
        # bit; the sums and the ticket are back at zero for the next launch; smooth flow stays on the matrix pipe, rough flow leaves it
        reg = torch.zeros(8, dtype=torch.int32, device=DEV)
        for _ in range(2):
            auto = ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True, recs=recs, regime=reg)
            r = reg.tolist()
            assert r[1:4] == [0, 0, 0] and r[0] in (0, 1) and 0 <= r[4] <= r[5] <= B * H * W, r
            assert torch.equal(auto, fp32 if r[0] else got), f"dispatched alt lookup is not the {'fp32' if r[0] else 'matrix-pipe'} kernel's, {name}"
        if name == "smooth":
            assert r[0] == 0 and r[4] == 0, r
        if name == "rough" and H * W >= 4096:       # (small grids: most rough windows miss the image altogether and cost neither kernel anything)
            assert r[0] == 1, r


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
