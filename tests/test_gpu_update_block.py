"""GPU parity tests, rows a6-a10: update-block convolutions (forward / data gradient / weight gradient), heads, convex upsampler, helpers, weight packs, amax words.
(Split out of the former tests/test_gpu_parity.py in round 6; shared helpers, fixtures and the ONE tolerance table live in
tests/_gpu_common.py.)"""
import pytest

from _gpu_common import *      # noqa: F401,F403  (helpers, fixtures, tolerance table)

pytestmark = pytest.mark.gpu


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
