"""GPU parity tests, rows a1-a10 together: end-to-end flow, train steps against reference-generated digests, bench-scale steps, the reference-shaped shell.
(Split out of the former tests/test_gpu_parity.py in round 6; shared helpers, fixtures and the ONE tolerance table live in
tests/_gpu_common.py.)"""
import pytest

from _gpu_common import *      # noqa: F401,F403  (helpers, fixtures, tolerance table)

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("case", ["raft_b3_iters2", "raft_alt_iters3", "l2l_offsets_per_sample", "l2l_sup_grad_samples"])
def test_per_step_batches_against_the_per_iteration_path_on_odd_shapes(case):
    """update.HeadBatch + MotionBatch + the batched heads backward (everything outside the recurrence once per step) against one
    launch per iteration, beyond the shapes of the golden train steps: a batch of 3 on a 9x13 grid with two iterations,
    alt-corr, L2L with per-sample crop offsets, and L2L with sup_grad_samples=1 (uncropped frames of sample 1 encoded without
    a graph, supervisor-phase backward on sample 0 alone) under a loss that keeps that promise, against the plain forward.
    Predictions to 1e-4 px, every parameter gradient outside the feature encoder to 5e-3 relative (tests/_per_iteration_compare.py)."""
    import _per_iteration_compare as D
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
    generation-2 collection happened to run -- ~27 MB per step at the bench shape, a creeping peak. """
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
    from flow_supervisor_amd.core import update as U
    calls = []
    real = U._Engine.context
    U._Engine.context = lambda self, inp, params: (calls.append(1), real(self, inp, params))[1]
    try:
        for m in (a, b):
            preds = m(im1, im2, iters=4)
            raft_sequence_loss(preds).backward()
            outs.append((preds, {k: p.grad for k, p in m.named_parameters() if p.grad is not None}))
    finally:
        U._Engine.context = real
    # (round 6: the reference's loop passes the same `inp` every iteration; its context convolutions run once per step in both shells)
    assert len(calls) == 2, f"context convolutions ran {len(calls)} times for two steps"
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


def test_reference_upsample_flow_takes_the_mask_view():
    """`up_mask` leaves BasicUpdateBlock.forward as an NCHW-shaped view of the channels-last mask (update.as_nchw).  A maintainer who
    keeps the reference's OWN upsample_flow (pytorch/core/raft.py:72-83: `mask.view(N, 1, 9, 8, 8, H, W)`, softmax over the nine taps,
    unfold, sum, permute, reshape -- the oracle's restatement of those lines) must get the same flow from the view as from a contiguous
    copy, and as from this package's kernel."""
    from flow_supervisor_amd.core import update as U
    from flow_supervisor_amd.core.raft import convex_upsample
    ns_ = argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False, corr_levels=4, corr_radius=4)
    blk = U.BasicUpdateBlock(ns_).to(DEV)
    B, H, W = 2, 12, 20
    net = torch.tanh(rand_tensor((B, 128, H, W), 3)).to(DEV)
    inp = torch.relu(rand_tensor((B, 128, H, W), 4)).to(DEV)
    corr = rand_tensor((B, 324, H, W), 5).to(DEV)
    flow = rand_tensor((B, 2, H, W), 6).to(DEV)
    with torch.no_grad():
        n2, mask, delta = blk(net, inp, corr, flow)
    assert mask.shape == (B, 576, H, W) and n2.shape == (B, 128, H, W)
    assert not mask.is_contiguous() and mask.is_contiguous(memory_format=torch.channels_last)
    import torch.nn.functional as F

    def with_view(flow, mask):       # the tensor calls of raft.py:72-83, `.view` included
        N, _, H, W = flow.shape
        w = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
        nb = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
        return (w * nb).sum(2).permute(0, 1, 4, 2, 5, 3).reshape(N, 2, 8 * H, 8 * W)

    up_view = with_view(flow, mask)
    up_copy = with_view(flow, mask.contiguous())
    assert torch.equal(up_view, up_copy)
    close(up_view, O.upsample_flow(flow.cpu(), mask.cpu()), 1e-5, what="view-style upsampling vs the oracle")
    close(convex_upsample(flow, mask), up_copy, 1e-5, what="convex upsampler on the mask view")


def test_reference_shaped_calls_hand_channels_last_twins_from_block_to_block():
    """the NCHW tensors the reference-shaped entry points return (CorrBlock.__call__, BasicUpdateBlock.forward)
    remember their channels-last originals, and the next swapped block continues from those instead of transposing the copy back.
    Same numbers and gradients with the twins on and off; a tensor written in place by the caller loses its twin."""
    from flow_supervisor_amd.core import update as U
    from flow_supervisor_amd.core.raft_dropin import ReferenceShapedRAFT
    seed = 5
    im1, im2 = (t.to(DEV) for t in synthetic_pair(2, 128, 192, seed + 1))
    res = {}
    old = U.TWINS
    try:
        for flag in (True, False):
            U.TWINS = flag
            m = ReferenceShapedRAFT(ns(False))
            m.load_state_dict(procedural_state_dict(shapes("raft_basic"), seed))
            m = m.to(DEV).train()
            m.freeze_bn()
            preds = m(im1, im2, iters=3)
            O.sequence_loss_zero_gt(preds).backward()
            res[flag] = (preds[-1].detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    finally:
        U.TWINS = old
    close(res[True][0], res[False][0], 1e-5, what="last prediction, twins on / off")
    for k in res[False][1]:
        e = _rel_l2(res[True][1][k], res[False][1][k])
        assert e < 2e-3 or res[False][1][k].norm().item() < 1e-4, (k, e)
    # the twin follows the tensor's version
    x_cl = torch.randn(1, 6, 8, 16, device=DEV)
    y = U.from_channels_last(x_cl)
    assert U.to_channels_last(y) is x_cl
    y.mul_(2.0)
    back = U.to_channels_last(y)
    assert back is not x_cl and torch.equal(back, 2.0 * x_cl)
