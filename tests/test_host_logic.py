"""Host-side logic that needs no GPU: weight re-layouts of the encoder path."""
import torch
import torch.nn.functional as F

from flow_supervisor_amd.core.extractor import _s2d_weight, _s2d_weight_grad


def _space_to_depth(x):
    B, C, H, W = x.shape
    return x.view(B, C, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(B, 4 * C, H // 2, W // 2)


def test_stride2_conv_equals_2x2_conv_over_space_to_depth():
    """The identity the stride-2 encoder units rest on (core/extractor.py::_StridedPairFn): conv3x3(stride 2, pad 1)(x) ==
    conv2x2(stride 1, one row / column of padding above / left)(space_to_depth(x)) with the re-laid-out weights, and the
    1x1 stride-2 shortcut == a 1x1 over the first C channels."""
    torch.manual_seed(0)
    w = torch.randn(5, 3, 3, 3, dtype=torch.double)
    x = torch.randn(2, 3, 8, 10, dtype=torch.double)
    xs = _space_to_depth(x)
    out = F.conv2d(F.pad(xs, (1, 0, 1, 0)), _s2d_weight(w))
    assert torch.allclose(out, F.conv2d(x, w, None, 2, 1), atol=1e-12)
    wsc = torch.randn(4, 3, 1, 1, dtype=torch.double)
    assert torch.allclose(F.conv2d(xs[:, :3], wsc), F.conv2d(x, wsc, None, 2, 0), atol=1e-12)


def test_s2d_weight_grad_is_the_adjoint():
    torch.manual_seed(1)
    w = torch.randn(4, 6, 3, 3, dtype=torch.double, requires_grad=True)
    g = torch.randn(4, 24, 2, 2, dtype=torch.double)
    (_s2d_weight(w) * g).sum().backward()
    assert torch.equal(w.grad, _s2d_weight_grad(g, 6))
    # 9 of the 16 (tap, sub-pixel) slots carry weights, 7 are structural zeros
    assert int((_s2d_weight(torch.ones(1, 1, 3, 3)) != 0).sum()) == 9


def test_tf_twins_with_same_pooling_refuse_to_be_trained_through():
    """raft_tf.calc_all_field / build_pyramid / CorrBlock are differentiable on floor-sized pyramids since round 5 (GPU test
    test_tf_twins_train_through_volume_pyramid_and_lookup); a pyramid that needs TensorFlow's 'SAME' pooling (a pooled size is odd:
    TF-only semantics, parity-unpinned, no backward kernels) must still fail loudly with a tensor that requires grad instead of
    returning outputs without grad_fn.  The refusal is decided on the host, before any kernel is asked for."""
    import pytest
    from flow_supervisor_amd import raft_tf
    a = torch.randn(1, 6, 10, 16, requires_grad=True)            # 6 x 10 -> 3 x 5 -> odd: 'SAME' pooling at num_pool = 2
    with pytest.raises(RuntimeError, match="forward-only"):
        raft_tf.calc_all_field(a, a.detach(), num_pool=2)
    with pytest.raises(RuntimeError, match="forward-only"):
        raft_tf.build_pyramid(torch.randn(1, 6, 10, 6, 10, requires_grad=True), 2)
    with pytest.raises(RuntimeError, match="forward-only"):
        raft_tf.CorrBlock(2, 3)([torch.randn(1, 4, 4, 5, 5, requires_grad=True), torch.randn(1, 4, 4, 3, 3)], torch.zeros(1, 4, 4, 2))
    # floor-sized pyramids reach the kernels: on CPU tensors that is the library's "no CPU implementation" error, not a silent result
    b = torch.randn(1, 8, 8, 16, requires_grad=True)
    with pytest.raises(RuntimeError, match="CUDA"):
        raft_tf.calc_all_field(b, b.detach(), num_pool=1)


def test_gradients_handed_out_back_to_back_come_back_as_one_tensor():
    """update._one_tensor (what lets HeadBatch take the loss kernel's T gradients back as one [T*B,...] tensor without a copy):
    pieces of one buffer in order -> a view of it; anything else (a gap, a copy, another order, a missing piece) -> None."""
    from flow_supervisor_amd.core.update import _one_tensor
    big = torch.arange(4 * 2 * 3 * 5, dtype=torch.float32).view(4, 2, 3, 5)
    pieces = list(big.unbind(0))
    one = _one_tensor([p.view(1, 2, 3, 5) for p in pieces])
    assert one is not None and one.shape == (4, 2, 3, 5) and one.data_ptr() == big.data_ptr() and torch.equal(one, big)
    two = _one_tensor([big[0:2], big[2:4]])
    assert two is not None and two.shape == (4, 2, 3, 5) and torch.equal(two, big)
    assert _one_tensor([pieces[0].view(1, 2, 3, 5), pieces[2].view(1, 2, 3, 5)]) is None            # a gap
    assert _one_tensor([pieces[1].view(1, 2, 3, 5), pieces[0].view(1, 2, 3, 5)]) is None            # wrong order
    assert _one_tensor([pieces[0].view(1, 2, 3, 5), pieces[1].clone().view(1, 2, 3, 5)]) is None    # another storage
    assert _one_tensor([pieces[0].view(1, 2, 3, 5), None]) is None
    assert _one_tensor([big[:, :, :, ::2][0:1]]) is None                                          # not contiguous


def test_batched_flow_supervisor_loss_weights():
    """train.semi_sequence_losses: the labelled half is weighted like sequence_loss (gamma, gamma2), the unlabelled half like
    sequence_loss_unsup -- its OWN decay (the reference calls it without gamma: 0.8 whatever args.gamma, train.py:276), times
    unsup_weight, and zero for the supervisor's predictions."""
    from flow_supervisor_amd import train
    seen = {}

    class Fake:
        @staticmethod
        def apply(bs, w_sup, w_unsup, *rest):
            seen["bs"], seen["sup"], seen["unsup"] = bs, w_sup, w_unsup
            return None, None
    real, train._SemiLossFn = train._SemiLossFn, Fake
    try:
        train.semi_sequence_losses([object()] * 6, 1, None, None, gamma=0.85, unsup_weight=0.25)
    finally:
        train._SemiLossFn = real
    assert seen["bs"] == 1
    assert seen["sup"] == [0.85 ** 2, 0.85, 1.0, 1.0, 1.0, 1.0]
    assert seen["unsup"] == [0.25 * 0.8 ** 2, 0.25 * 0.8, 0.25, 0.0, 0.0, 0.0]


def test_per_step_batches_respect_32_bit_offsets():
    """HeadBatch / MotionBatch only exist where the batched buffers stay below 2 GB (the buffer-addressed kernels use 32-bit
    byte offsets); beyond that the loops fall back to one launch per iteration."""
    from flow_supervisor_amd.core.update import HeadBatch
    assert HeadBatch.fits(12, 4, 55, 128)            # the bench shape: 12 x 4 x 7040 x 576 floats = 779 MB
    assert HeadBatch.fits(12, 8, 46, 62)
    assert not HeadBatch.fits(12, 16, 55, 128)       # 3.1 GB
    assert HeadBatch.fits(32, 4, 55, 128)            # 2.08e9 bytes: just below 2^31
    assert not HeadBatch.fits(34, 4, 55, 128)


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without torchrun around it starts N children through
    torch.distributed.run on 127.0.0.1 (the driver's own launch line) and hands back their exit code; with fewer devices
    than ranks the children are told to share them (gloo through the host)."""
    import importlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    a = bench.parse()
    assert bench.self_launch(a) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert "FSRAFT_BENCH_SHARED_GPUS" not in seen["env"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    bench.self_launch(a)
    assert seen["env"]["FSRAFT_BENCH_SHARED_GPUS"] == "1"


def test_flat_gradients_deferred_exchange():
    """begin(exchange=False): the hooks gather, nobody exchanges; exchange_all() is then the step's one collective
    (world size 1 here: the identity) -- the route of a step replayed as two hipGraphs."""
    from flow_supervisor_amd.parallel import FlatGradients
    lin = torch.nn.Linear(3, 2)
    fg = FlatGradients(lin.parameters(), ["update.w", "update.b"])
    fg.begin(exchange=False)
    lin(torch.ones(4, 3)).sum().backward()
    fg.finish()
    assert torch.equal(lin.weight.grad, torch.full((2, 3), 4.0)) and lin.weight.grad.data_ptr() == fg.views[lin.weight].data_ptr()
    fg.exchange_all()
    assert torch.equal(fg.views[lin.bias], torch.full((2,), 4.0))
