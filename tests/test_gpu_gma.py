"""GPU parity tests, row a11 / f1 (config 5): GMA attention, aggregate, update block, end to end, bench scale.
(Split out of the former tests/test_gpu_parity.py in round 6; shared helpers, fixtures and the ONE tolerance table live in
tests/_gpu_common.py.)"""
import pytest

from _gpu_common import *      # noqa: F401,F403  (helpers, fixtures, tolerance table)

pytestmark = pytest.mark.gpu


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
