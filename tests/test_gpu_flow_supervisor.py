"""GPU parity tests, rows f2 / f4: L2L two-phase forward, the flow-supervisor step of the recipes, sequence losses, warm start.
(Split out of the former tests/test_gpu_parity.py in round 6; shared helpers, fixtures and the ONE tolerance table live in
tests/_gpu_common.py.)"""
import pytest

from _gpu_common import *      # noqa: F401,F403  (helpers, fixtures, tolerance table)

pytestmark = pytest.mark.gpu


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
