import torch

from .. import ops
from ..core import update as _upd
from ..core.raft import convex_upsample


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _same_is_forward_only(what, *tensors):
    """The 'SAME'-pooled pyramid (odd pooled sizes: TF-only semantics, parity-unpinned, no backward kernels) stays forward-only:
    training through it would silently cut the gradient to the feature encoder, so it fails instead.  Floor-sized pyramids -- every
    pooled dimension even, e.g. crops that are multiples of 64 -- are differentiable since round 5 (the GradientTape of
    raft/semi.py:198-303 runs over these very calls)."""
    if _wants_grad(*tensors):
        raise RuntimeError(f"flow_supervisor_amd.raft_tf.{what}: the TF 'SAME' pyramid (a pooled size is odd) is forward-only: call it "
                           "under torch.no_grad() or on detached tensors (flow_supervisor_amd.core.corr.CorrBlock trains through the "
                           "volume with the PyTorch reference's floor sizes)")


class _AllFieldFn(torch.autograd.Function):
    """(a, b) NHWC -> the floor-pooled pyramid of their all-pairs volume; backward = pooling chain folded into level 0
    (fsraft_corr_unpool_bwd) + the two volume-backward GEMMs (ops.corr_build_bwd)."""

    @staticmethod
    def forward(ctx, a, b, nlev):
        f1 = a.permute(0, 3, 1, 2).contiguous().float()
        f2 = b.permute(0, 3, 1, 2).contiguous().float()
        ctx.save_for_backward(f1, f2)
        return tuple(ops.corr_build(f1, f2, nlev))

    @staticmethod
    def backward(ctx, *dl):
        f1, f2 = ctx.saved_tensors
        B, C, H, W = f1.shape
        sizes = ops.pyramid_sizes(H, W, len(dl))
        dlev = [(g.contiguous().clone() if l == 0 else g.contiguous()) if g is not None else
                torch.zeros(B * H * W, 1, sizes[l][0], sizes[l][1], device=f1.device) for l, g in enumerate(dl)]
        d1, d2 = ops.corr_build_bwd(f1, f2, dlev)
        return d1.permute(0, 2, 3, 1), d2.permute(0, 2, 3, 1), None


class _TransposeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v):
        B, H, W, H2, W2 = v.shape
        ctx.shape = (B, H, W, H2, W2)
        return ops.transpose_batched(v.reshape(B, H * W, H2 * W2).float()).view(B, H2, W2, H, W)

    @staticmethod
    def backward(ctx, g):
        B, H, W, H2, W2 = ctx.shape
        return ops.transpose_batched(g.reshape(B, H2 * W2, H * W).contiguous()).view(B, H, W, H2, W2)


class _PyramidFn(torch.autograd.Function):
    """level 0 [rows, 1, h, w] -> floor-pooled levels; backward = the pooling chain folded into level 0 in place."""

    @staticmethod
    def forward(ctx, level0, nlev):
        ctx.shape = tuple(level0.shape)
        return tuple(ops.corr_pool_pyramid(level0, nlev))

    @staticmethod
    def backward(ctx, *dl):
        rows, _, h, w = ctx.shape
        sizes = ops.pyramid_sizes(h, w, len(dl))
        dev = next(g for g in dl if g is not None).device
        dlev = [(g.contiguous().clone() if l == 0 else g.contiguous()) if g is not None else
                torch.zeros(rows, 1, sizes[l][0], sizes[l][1], device=dev) for l, g in enumerate(dl)]
        if rows % (h * w):
            raise RuntimeError("raft_tf.build_pyramid backward: the un-pooling kernel takes query counts that are a multiple of the "
                               "level-0 plane (volumes between two feature maps of one size, as in raft/semi.py)")
        ops.corr_unpool_bwd_(dlev, rows // (h * w), h, w)
        return dlev[0], None


class _LookupFn(torch.autograd.Function):
    """Radius-r lookup on row-major levels; backward scatters dOut into zero-initialised level gradients (fsraft_corr_lookup_bwd).
    No gradient for the coordinates: the RAFT loops stop it (raft/__init__.py: tf.stop_gradient(coords1); pytorch raft.py:122)."""

    @staticmethod
    def forward(ctx, coords, radius, *levels):
        B, H, W, _ = coords.shape
        c = coords.permute(0, 3, 1, 2).float()
        ctx.save_for_backward(c)
        ctx.radius, ctx.shapes = radius, [tuple(lv.shape) for lv in levels]
        return ops.corr_lookup_fwd([lv.contiguous() for lv in levels], c, radius, nhwc=True)

    @staticmethod
    def backward(ctx, g):
        (c,) = ctx.saved_tensors
        dl = [torch.zeros(sh, device=g.device, dtype=torch.float32) for sh in ctx.shapes]
        ops.corr_lookup_bwd_(dl, c, g.contiguous(), ctx.radius, nhwc=True)
        return (None, None) + tuple(dl)


def calc_all_field(a, b, num_pool=0):
    """a, b: [B,H,W,C] feature maps -> list of num_pool+1 volumes [B,H,W,h_l,w_l] (raft/allfield.py:61-92)."""
    B, H, W, C = a.shape
    if _same_needed(H, W, num_pool):
        _same_is_forward_only("calc_all_field", a, b)
        f1 = a.permute(0, 3, 1, 2).contiguous().float()
        f2 = b.permute(0, 3, 1, 2).contiguous().float()
        levels = ops.corr_pool_pyramid(ops.corr_build(f1, f2, 1)[0], num_pool + 1, same=True)
    elif _wants_grad(a, b):
        levels = _AllFieldFn.apply(a, b, num_pool + 1)
    else:
        levels = ops.corr_build(a.permute(0, 3, 1, 2).contiguous().float(), b.permute(0, 3, 1, 2).contiguous().float(), num_pool + 1)
    return [lv.view(B, H, W, lv.shape[-2], lv.shape[-1]) for lv in levels]


def _same_needed(h, w, num_pool):
    # TF pools level 0 with padding='SAME' (ceil sizes, partial edge windows averaged over their in-range elements); the
    # fused build follows the PyTorch reference (avg_pool2d, floor sizes).  The two agree when every pooled dimension stays
    # even; otherwise the pyramid comes from the SAME pooling kernel (fsraft_corr_pool_pyramid_same) and the lookup is told.
    return bool(num_pool) and bool(h % (1 << num_pool) or w % (1 << num_pool))


def transpose_volume(c_volume):
    """tf.transpose(c_volume, [0, 3, 4, 1, 2]) of a [B,H,W,H2,W2] volume (raft/semi.py:250, 257: the backward-flow volume is
    the forward one read the other way), materialised by the tiled transpose kernel instead of a strided copy.  Differentiable
    (the gradient is transposed back by the same kernel)."""
    if _wants_grad(c_volume):
        return _TransposeFn.apply(c_volume)
    B, H, W, H2, W2 = c_volume.shape
    return ops.transpose_batched(c_volume.reshape(B, H * W, H2 * W2).float()).view(B, H2, W2, H, W)


def build_pyramid(c_volume, num_pool=0):
    """[B,H,W,H2,W2] volume -> [c_volume, pooled x2, x4, ...] (raft/allfield.py:94-106), e.g. on transpose_volume(...) for
    the backward flow (raft/semi.py:251, 258) without a second all-pairs GEMM."""
    B, H, W, H2, W2 = c_volume.shape
    same = _same_needed(H2, W2, num_pool)
    lv0 = c_volume.reshape(B * H * W, 1, H2, W2).float()
    if same:
        _same_is_forward_only("build_pyramid", c_volume)
        levels = ops.corr_pool_pyramid(lv0, num_pool + 1, same=True)
    elif _wants_grad(c_volume):
        levels = _PyramidFn.apply(lv0.contiguous(), num_pool + 1)
    else:
        levels = ops.corr_pool_pyramid(lv0, num_pool + 1)
    return [lv.view(B, H, W, lv.shape[-2], lv.shape[-1]) for lv in levels]


class CorrBlock:
    """Stateless lookup object (raft/corr.py:5-22): obj(corr_pyramid, coords[B,H,W,2]) -> [B,H,W,L*(2r+1)^2]."""

    def __init__(self, num_levels=4, radius=4, is_max_disp=False):
        self.num_levels, self.radius, self.is_max_disp = num_levels, radius, is_max_disp
        self.corr_pyramid = []

    def __call__(self, corr_pyramid, coords, is_coord=True):
        #  the reference always hands in stop_gradient(coords) (raft/semi.py:59, 99): the lookup has no gradient
        # w.r.t. them.  A caller that passes coords with a live gradient would get it cut silently -- say so once.
        if coords.requires_grad and torch.is_grad_enabled():
            import warnings
            warnings.warn("raft_tf CorrBlock: the lookup does not differentiate through `coords` (the reference passes "
                          "tf.stop_gradient(coords)); their gradient is cut here", RuntimeWarning, stacklevel=2)
        B, H, W, _ = coords.shape
        levels = [lv.reshape(B * H * W, 1, lv.shape[-2], lv.shape[-1]) for lv in corr_pyramid]
        h2, w2 = corr_pyramid[0].shape[-2:]
        same = any(tuple(lv.shape[-2:]) != (h2 >> l, w2 >> l) for l, lv in enumerate(corr_pyramid))    # TF 'SAME' (ceil) sizes
        if same:
            _same_is_forward_only("CorrBlock.__call__", *corr_pyramid)
        elif _wants_grad(*corr_pyramid):
            return _LookupFn.apply(coords.detach(), self.radius, *levels)
        c = coords.permute(0, 3, 1, 2)                       # NCHW view of the NHWC coords: strides, no copy
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
