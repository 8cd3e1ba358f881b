import torch

from .. import ops
from ..core import update as _upd
from ..core.raft import convex_upsample


def _forward_only(what, *tensors):
    """The volume / pyramid / lookup twins are forward-only (no autograd.Function behind them, unlike the upsampler and the
    update block): training through them would silently cut the gradient to the feature encoder, so it fails instead.
    (The differentiable path for these three steps is core.corr.CorrBlock.)"""
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise RuntimeError(f"flow_supervisor_amd.raft_tf.{what} is forward-only: call it under torch.no_grad() or on detached "
                           "tensors (use flow_supervisor_amd.core.corr.CorrBlock to train through the volume)")


def calc_all_field(a, b, num_pool=0):
    """a, b: [B,H,W,C] feature maps -> list of num_pool+1 volumes [B,H,W,h_l,w_l] (raft/allfield.py:61-92)."""
    _forward_only("calc_all_field", a, b)
    B, H, W, C = a.shape
    f1 = a.permute(0, 3, 1, 2).contiguous().float()
    f2 = b.permute(0, 3, 1, 2).contiguous().float()
    if _same_needed(H, W, num_pool):
        levels = ops.corr_pool_pyramid(ops.corr_build(f1, f2, 1)[0], num_pool + 1, same=True)
    else:
        levels = ops.corr_build(f1, f2, num_pool + 1)
    return [lv.view(B, H, W, lv.shape[-2], lv.shape[-1]) for lv in levels]


def _same_needed(h, w, num_pool):
    # TF pools level 0 with padding='SAME' (ceil sizes, partial edge windows averaged over their in-range elements); the
    # fused build follows the PyTorch reference (avg_pool2d, floor sizes).  The two agree when every pooled dimension stays
    # even; otherwise the pyramid comes from the SAME pooling kernel (fsraft_corr_pool_pyramid_same) and the lookup is told.
    return bool(num_pool) and bool(h % (1 << num_pool) or w % (1 << num_pool))


def transpose_volume(c_volume):
    """tf.transpose(c_volume, [0, 3, 4, 1, 2]) of a [B,H,W,H2,W2] volume (raft/semi.py:250, 257: the backward-flow volume is
    the forward one read the other way), materialised by the tiled transpose kernel instead of a strided copy."""
    _forward_only("transpose_volume", c_volume)
    B, H, W, H2, W2 = c_volume.shape
    return ops.transpose_batched(c_volume.reshape(B, H * W, H2 * W2).float()).view(B, H2, W2, H, W)


def build_pyramid(c_volume, num_pool=0):
    """[B,H,W,H2,W2] volume -> [c_volume, pooled x2, x4, ...] (raft/allfield.py:94-106), e.g. on transpose_volume(...) for
    the backward flow (raft/semi.py:251, 258) without a second all-pairs GEMM."""
    _forward_only("build_pyramid", c_volume)
    B, H, W, H2, W2 = c_volume.shape
    levels = ops.corr_pool_pyramid(c_volume.reshape(B * H * W, H2, W2).float(), num_pool + 1, same=_same_needed(H2, W2, num_pool))
    return [lv.view(B, H, W, lv.shape[-2], lv.shape[-1]) for lv in levels]


class CorrBlock:
    """Stateless lookup object (raft/corr.py:5-22): obj(corr_pyramid, coords[B,H,W,2]) -> [B,H,W,L*(2r+1)^2]."""

    def __init__(self, num_levels=4, radius=4, is_max_disp=False):
        self.num_levels, self.radius, self.is_max_disp = num_levels, radius, is_max_disp
        self.corr_pyramid = []

    def __call__(self, corr_pyramid, coords, is_coord=True):
        _forward_only("CorrBlock.__call__", coords, *corr_pyramid)
        B, H, W, _ = coords.shape
        levels = [lv.reshape(B * H * W, 1, lv.shape[-2], lv.shape[-1]) for lv in corr_pyramid]
        c = coords.permute(0, 3, 1, 2)                       # NCHW view of the NHWC coords: strides, no copy
        h2, w2 = corr_pyramid[0].shape[-2:]
        same = any(tuple(lv.shape[-2:]) != (h2 >> l, w2 >> l) for l, lv in enumerate(corr_pyramid))    # TF 'SAME' (ceil) sizes
        return ops.corr_lookup_fwd(levels, c.float(), self.radius, nhwc=True, same=same)


class UpsampleConvexWithMask:
    """call([x, mask(, ref)]) with x [B,H,W,C=2], mask [B,H,W,576] -> [B,8H,8W,2] cropped to ref's H,W
    (raft/upsample.py:11-41).  Unlike the PyTorch method the TF layer does not scale the flow by 8."""

    def __init__(self, scale=8, **kwargs):
        if scale != 8:
            raise NotImplementedError("the HIP upsampler is built for the 8x factor RAFT uses")
        self.scale = scale

    def call(self, inputs, training=None, mask=None):
        if not isinstance(inputs, (list, tuple)):
            raise ValueError
        x, m = inputs[0], inputs[1]
        up = convex_upsample(x.permute(0, 3, 1, 2).contiguous() * 0.125, m.contiguous(), channels_last=True)
        up = up.permute(0, 2, 3, 1)
        if len(inputs) == 3:
            up = up[:, : inputs[2].shape[1], : inputs[2].shape[2]]
        return up

    __call__ = call


class BasicUpdateBlock(_upd.BasicUpdateBlock):
    """Keras-style entry point: call([net, inp, corr, flow]) on NHWC tensors -> (net, mask, delta_flow) NHWC
    (raft/smurf_models/raft_update.py:180-212).  Parameters keep the PyTorch names and OIHW shapes."""

    def call(self, inputs, training=None):
        net, inp, corr, flow = inputs
        h, mask, delta = self.forward_cl(net.contiguous(), inp.contiguous(), corr.contiguous(),
                                         flow.permute(0, 3, 1, 2))
        return h, mask, delta.permute(0, 2, 3, 1)
