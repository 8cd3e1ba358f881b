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


def test_tf_twins_of_the_volume_refuse_to_be_trained_through():
    """raft_tf.calc_all_field / build_pyramid / transpose_volume / CorrBlock have no autograd behind them: with a tensor that
    requires grad they must fail loudly instead of returning outputs without grad_fn."""
    import pytest
    from flow_supervisor_amd import raft_tf
    a = torch.randn(1, 8, 8, 16, requires_grad=True)
    with pytest.raises(RuntimeError, match="forward-only"):
        raft_tf.calc_all_field(a, a.detach(), num_pool=1)
    with pytest.raises(RuntimeError, match="forward-only"):
        raft_tf.build_pyramid(torch.randn(1, 4, 4, 4, 4, requires_grad=True), 1)
    with pytest.raises(RuntimeError, match="forward-only"):
        raft_tf.CorrBlock(2, 3)([torch.randn(1, 4, 4, 4, 4)], torch.zeros(1, 4, 4, 2, requires_grad=True))
