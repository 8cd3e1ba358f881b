"""Feature / context encoders.  These are callers of the hot path, kept as PyTorch-ROCm
convolutions on purpose (BASELINE.json north_star); only the module/parameter names of
pytorch/core/extractor.py are reproduced so reference checkpoints load (state_dict keys
``conv1``, ``norm1``, ``layer{1,2,3}.{0,1}.conv{1,2,3}``, ``...downsample.{0,1}``, ``conv2``).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .._lib import on_tensor_device


class _InstNormRelu(torch.autograd.Function):
    """relu?(instance_norm(x)) for NCHW fp32 on the fsraft kernels (two passes over the data each way)."""

    @staticmethod
    def forward(ctx, x, eps, relu):
        from .. import _lib as L
        x = x.contiguous()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(N * C, 2, device=x.device, dtype=torch.float32)
        L.check(L.load().fsraft_inorm_relu_fwd(L.ptr(x), L.ptr(y), L.ptr(stats), N * C, H * W, float(eps), int(relu),
                                               L.stream()), "inorm_relu_fwd")
        ctx.save_for_backward(x, stats)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, g):
        from .. import _lib as L
        x, stats = ctx.saved_tensors
        N, C, H, W = x.shape
        g = g.contiguous()
        dx = torch.empty_like(x)
        L.check(L.load().fsraft_inorm_relu_bwd(L.ptr(g), L.ptr(x), L.ptr(stats), L.ptr(dx), N * C, H * W, int(ctx.relu),
                                               L.stream()), "inorm_relu_bwd")
        return dx, None, None


class _FrozenBNRelu(torch.autograd.Function):
    """relu?(batch_norm(x + cbias[c])) with running statistics (eval mode / freeze_bn): y = x * scale[c] + shift[c].
    `cbias` is the bias of the convolution that produced x (or None): the convolution runs without it, the constant is
    folded into the shift, and its gradient -- the per-channel sum of the convolution's output gradient -- falls out of
    the sums this backward computes anyway, instead of a separate [N,H,W] reduction inside the convolution backward."""

    @staticmethod
    def forward(ctx, x, cbias, weight, bias, rm, rv, eps, relu):
        from .. import _lib as L
        x = x.contiguous()
        N, C, H, W = x.shape
        rs = torch.rsqrt(rv.float() + eps)
        scale = (weight.float() * rs).contiguous()
        rmc = rm.float() - cbias.float() if cbias is not None else rm.float()       # effective mean seen by x
        shift = (bias.float() - rmc * scale).contiguous()
        y = torch.empty_like(x)
        L.check(L.load().fsraft_affine_relu_fwd(L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(y), N * C, C, H * W, int(relu),
                                                L.stream()), "affine_relu_fwd")
        ctx.save_for_backward(x, scale, shift, rs, rmc)
        ctx.relu = relu
        ctx.has_cbias = cbias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        from .. import _lib as L
        x, scale, shift, rs, rm = ctx.saved_tensors
        N, C, H, W = x.shape
        g = g.contiguous()
        dx = torch.empty_like(x)
        sums = ops.zeros(2, C, device=x.device)
        L.check(L.load().fsraft_affine_relu_bwd(L.ptr(g), L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(dx), L.ptr(sums[0]),
                                                L.ptr(sums[1]), N * C, C, H * W, int(ctx.relu), L.stream()), "affine_relu_bwd")
        dweight = rs * (sums[1] - rm * sums[0])               # sum g' * (x + cbias - rm) * rs
        dcbias = scale * sums[0] if ctx.has_cbias else None   # = sum over pixels of dx
        return dx, dcbias, dweight, sums[0], None, None, None, None


def _is_cl(x):
    return x.dim() == 4 and x.shape[1] > 1 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()


def _cl_norm_ok(x):
    return x.shape[1] % 4 == 0 and 4 <= x.shape[1] <= 256


def _fast(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4


def _as_cl(x):
    """x as a channels_last tensor (no autograd: for use inside Functions).  The tiled fsraft transposes move an NCHW
    activation at 4-6 TB/s; the framework's strided copy reaches 1.4-2 TB/s on the same tensors (round 2, docs/history)."""
    if _is_cl(x):
        return x
    if _fast(x) and x.is_contiguous():
        from .. import ops
        B, C, H, W = x.shape
        return ops.nchw_to_nhwc(x, torch.empty(B, H, W, C, device=x.device, dtype=torch.float32)).permute(0, 3, 1, 2)
    return x.contiguous(memory_format=torch.channels_last)


def _as_nchw(x):
    if x.is_contiguous():
        return x
    if _fast(x) and _is_cl(x):
        from .. import ops
        return ops.nhwc_to_nchw(x.permute(0, 2, 3, 1))
    return x.contiguous()


class _ToNCHW(torch.autograd.Function):
    """x.contiguous() for a channels_last activation whose consumer is an NCHW MIOpen call; the gradient goes back
    channels_last, the layout its producer works in."""

    @staticmethod
    def forward(ctx, x):
        return _as_nchw(x)

    @staticmethod
    def backward(ctx, g):
        return _as_cl(g)


class _ResLink:
    """Ties the two consumers of a residual unit's input x together in backward: the unit's last norm kernel writes the
    shortcut gradient dres and parks it HERE (autograd gets None from it for x); the first convolution's backward -- which always
    runs later: the norm sits downstream of it -- accumulates its data gradient into that tensor through its epilogue and
    returns the SUM as x's gradient.  Autograd would otherwise add the two x-sized tensors in a separate pass (12 of them per
    step, up to 230 MB each).  Autograd sees one ordinary gradient for x from the unit, so further consumers of x, tensor
    hooks and retain_grad behave as usual (round 3 handed dres to autograd and then added into it behind the engine's back,
    which was only right while x had exactly these two consumers: ADVICE r2).
    `armed` is set by the first convolution's forward (_ConvCL, the only consumer that looks into the link): a norm whose unit's
    first convolution took another route (F.conv2d / MIOpen never sees the link) hands dres to autograd as usual instead of
    parking it where nobody collects it."""
    __slots__ = ("dres", "armed")

    def __init__(self):
        self.dres = None
        self.armed = False


class _InstNormReluCL(torch.autograd.Function):
    """_InstNormRelu for channels_last tensors (storage [N][H*W][C]): csrc/norm_cl.hip.  With `res` (the shortcut of a
    residual unit, channels_last) the result is relu(res + relu?(norm(x))) in the same pass -- the unit's add and final ReLU
    (and their backward) cost no extra trip over the tensor."""

    @staticmethod
    def forward(ctx, x, eps, relu, res=None, link=None, sums=None, s2d=False):
        from .. import _lib as L
        ctx.in_cl = _is_cl(x)
        ctx.link = link
        x = _as_cl(x)
        N, C, H, W = x.shape
        if res is not None:
            res = _as_cl(res)
        # s2d: the result leaves in the space-to-depth layout of the stride-2 unit that consumes it -- a [N, 4C, H/2, W/2]
        # channels_last tensor -- and the gradient comes back in it (see _emit_s2d)
        ctx.s2w = W if s2d else 0
        y = (torch.empty(N, H // 2, W // 2, 4 * C, device=x.device, dtype=torch.float32).permute(0, 3, 1, 2) if s2d
             else torch.empty_like(x))                            # preserves channels_last
        # partial rows (norm_cl.hip) -- already filled when the convolution that produced x carried the sums in its epilogue
        have = sums is not None and sums.acc is not None and tuple(sums.acc.shape) == (2, N * 8, C)
        acc = sums.acc if have else ops.zeros(2, N * 8, C, device=x.device)
        stats = torch.empty(N, C, 2, device=x.device, dtype=torch.float32)
        ops.tracked(y)          # (the kernel raises y's amax word: the convolution behind the norm finds its scale there)
        L.check(L.load().fsraft_inorm_relu_cl_fwd(L.ptr(x), L.ptr(res), L.ptr(y), L.ptr(acc[0]), L.ptr(acc[1]), L.ptr(stats), N, H * W,
                                                  C, float(eps), int(relu), int(have), ctx.s2w, L.ptr(ops.amax_of(y)), L.stream()), "inorm_relu_cl_fwd")
        ctx.fused = res is not None
        ctx.save_for_backward(x, stats, y if ctx.fused else None)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, g):
        from .. import _lib as L
        x, stats, out = ctx.saved_tensors
        N, C, H, W = x.shape
        g = _as_cl(g)
        dx = ops.tracked(torch.empty_like(x))              # (its word is raised by the kernel: the producing convolution's backward reads dx)
        dres = torch.empty_like(x) if ctx.fused else None
        acc = ops.zeros(2, N * 8, C, device=x.device)      # partial rows (norm_cl.hip)
        L.check(L.load().fsraft_inorm_relu_cl_bwd(L.ptr(g), L.ptr(x), L.ptr(stats), L.ptr(out), L.ptr(acc[0]), L.ptr(acc[1]), L.ptr(dx),
                                                  L.ptr(dres), N, H * W, C, int(ctx.relu), ctx.s2w, L.ptr(ops.amax_of(dx)), L.stream()),
                "inorm_relu_cl_bwd")
        if ctx.link is not None and ctx.link.armed:
            ctx.link.dres, dres = dres, None      # the shortcut's gradient travels through the link: conv1's backward returns the sum
        return dx if ctx.in_cl else _as_nchw(dx), None, None, dres, None, None, None     # an NCHW producer (MIOpen) gets an NCHW gradient


class _FrozenBNReluCL(torch.autograd.Function):
    """_FrozenBNRelu for channels_last tensors; `res` as in _InstNormReluCL."""

    @staticmethod
    def forward(ctx, x, cbias, weight, bias, rm, rv, eps, relu, res=None, link=None, s2d=False):
        from .. import _lib as L
        ctx.in_cl = _is_cl(x)
        ctx.link = link
        x = _as_cl(x)
        N, C, H, W = x.shape
        ctx.s2w = W if s2d else 0                     # (as in _InstNormReluCL)
        if res is not None:
            res = _as_cl(res)
        fold = torch.empty(4, C, device=x.device, dtype=torch.float32)       # scale, shift, rs, rmc: one launch (csrc/norm_cl.hip)
        scale, shift, rs, rmc = fold[0], fold[1], fold[2], fold[3]
        L.check(L.load().fsraft_bn_fold(L.ptr(weight.detach().float().contiguous()), L.ptr(bias.detach().float().contiguous()),
                                        L.ptr(rm.float().contiguous()), L.ptr(rv.float().contiguous()),
                                        L.ptr(cbias.detach().float().contiguous()) if cbias is not None else None, float(eps), C,
                                        L.ptr(scale), L.ptr(shift), L.ptr(rs), L.ptr(rmc), L.stream()), "bn_fold")
        y = torch.empty(N, H // 2, W // 2, 4 * C, device=x.device, dtype=torch.float32).permute(0, 3, 1, 2) if s2d else torch.empty_like(x)
        ops.tracked(y)
        L.check(L.load().fsraft_affine_relu_cl_fwd(L.ptr(x), L.ptr(res), L.ptr(scale), L.ptr(shift), L.ptr(y), N * H * W, C, int(relu),
                                                   H * W, ctx.s2w, L.ptr(ops.amax_of(y)), L.stream()), "affine_relu_cl_fwd")
        ctx.fused = res is not None
        ctx.save_for_backward(x, scale, shift, rs, rmc, y if ctx.fused else None)
        ctx.relu = relu
        ctx.has_cbias = cbias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        from .. import _lib as L
        x, scale, shift, rs, rm, out = ctx.saved_tensors
        N, C, H, W = x.shape
        g = _as_cl(g)
        dx = ops.tracked(torch.empty_like(x))
        dres = torch.empty_like(x) if ctx.fused else None
        part = ops.zeros(2, N * 8, C, device=x.device)      # partial rows (see norm_cl.hip)
        L.check(L.load().fsraft_affine_relu_cl_bwd(L.ptr(g), L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(out), L.ptr(dx), L.ptr(dres),
                                                   L.ptr(part[0]), L.ptr(part[1]), N, H * W, C, int(ctx.relu), ctx.s2w,
                                                   L.ptr(ops.amax_of(dx)), L.stream()), "affine_relu_cl_bwd")
        dpar = torch.empty(3, C, device=x.device, dtype=torch.float32)       # dweight, dbias, dcbias: one launch
        L.check(L.load().fsraft_bn_fold_bwd(L.ptr(part), N * 8, C, L.ptr(rs), L.ptr(rm), L.ptr(scale), L.ptr(dpar[0]), L.ptr(dpar[1]),
                                            L.ptr(dpar[2]) if ctx.has_cbias else None, L.stream()), "bn_fold_bwd")
        if ctx.link is not None and ctx.link.armed:
            ctx.link.dres, dres = dres, None      # (see _ResLink)
        return (dx if ctx.in_cl else _as_nchw(dx), dpar[2] if ctx.has_cbias else None, dpar[0], dpar[1], None, None, None, None, dres,
                None, None)


def _conv_key(conv):
    from .. import ops
    w = conv.weight
    return (w.data_ptr(), w._version, ops.exact_mode())


def _queue_conv_packs(plan, conv):
    """Queue the packed forward / data-gradient matrices of a stride-1 1x1 / 3x3 convolution; returns the handle tuple that
    _finish_conv_packs turns into the cached `(key, fwd fp32, fwd split, dgrad fp32, dgrad split, fwd fragment order, dgrad
    fragment order)`.  The fragment-order packs are the resident-patch kernel's (33..64 input channels, 3x3)."""
    from .. import ops
    wd = conv.weight.detach()
    N, C, KH, KW = wd.shape
    exact = ops.exact_mode()
    fs = plan.pack([wd], [C], 10)
    f = plan.pack([wd], [C], 0) if (exact or N <= 32) else fs
    ds = plan.pack([wd], [C], 11)
    d = plan.pack([wd], [C], 1) if (exact or C <= 32) else ds
    ff = plan.pack([wd], [C], 10, frag=True) if (KH, KW) == (3, 3) and 32 < C <= 64 else None
    df = plan.pack([wd], [C], 11, frag=True) if (KH, KW) == (3, 3) and 32 < N <= 64 else None
    return (f, fs, d, ds, ff, df)


def _prepare_packs(root):
    """Every packed weight image of an encoder (or of one convolution) whose parameter changed since it was packed, built by
    ONE batched launch per 16 matrices (fsraft_pack_conv_weights) instead of two to four launches per layer plus the torch
    ops of the space-to-depth rewrite and of the fragment-order permutation."""
    from .. import ops
    todo = []
    for m in root.modules():
        if isinstance(m, _Block) and _pair_shape_ok(m):
            w3, wsc = m.conv1.weight, m.downsample[0].weight
            key = (w3.data_ptr(), w3._version, wsc.data_ptr(), wsc._version, ops.exact_mode())
            c = m.__dict__.get("_fs_pair_packs")
            if (c is None or c[0] != key) and w3.is_cuda and w3.dtype == torch.float32:
                todo.append(("pair", m, key))
        if (isinstance(m, nn.Conv2d) and m.stride == (1, 1) and m.kernel_size in ((1, 1), (3, 3)) and m.groups == 1
                and m.weight.is_cuda and m.weight.dtype == torch.float32):
            c = m.__dict__.get("_fs_packs")
            key = _conv_key(m)
            if c is None or c[0] != key:
                todo.append(("conv", m, key))
    if not todo:
        return
    with torch.no_grad():
        plan = ops.PackPlan(todo[0][1].conv1.weight.device if todo[0][0] == "pair" else todo[0][1].weight.device)
        hs = []
        exact = ops.exact_mode()
        for kind, m, key in todo:
            if kind == "conv":
                hs.append(_queue_conv_packs(plan, m))
                continue
            w3, ws = m.conv1.weight.detach(), m.downsample[0].weight.detach()
            N, C = w3.shape[:2]
            # the 3x3 stride-2 weight seen as its 2x2 stride-1 equivalent over the space-to-depth input (flag s2d): [N][4C][2][2]
            a = plan.pack([w3], [4 * C], 10, cin_full=C, s2d=True)
            b = plan.pack([w3], [4 * C], 11, cin_full=C, s2d=True)
            a0 = plan.pack([w3], [4 * C], 0, cin_full=C, s2d=True) if (exact or N <= 32) else a
            b0 = plan.pack([w3], [4 * C], 1, cin_full=C, s2d=True) if (exact or 4 * C <= 32) else b
            Ns, Cs = ws.shape[:2]
            c_ = plan.pack([ws], [Cs], 10)
            d_ = plan.pack([ws], [Cs], 11)
            c0 = plan.pack([ws], [Cs], 0) if (exact or Ns <= 32) else c_
            d0 = plan.pack([ws], [Cs], 1) if (exact or Cs <= 32) else d_
            hs.append(((a0, a, b0, b), (c0, c_, d0, d_)))
        out = plan.run()
        for (kind, m, key), h in zip(todo, hs):
            if kind == "conv":
                m.__dict__["_fs_packs"] = (key,) + tuple(None if i is None else out[i] for i in h)
            else:
                m.__dict__["_fs_pair_packs"] = (key, tuple(out[i] for i in h[0]), tuple(out[i] for i in h[1]))


def _weight_packs(conv):
    """Packed forward / data-gradient matrices of a convolution for the split (fp16x3) implicit GEMM, cached on the module and
    rebuilt whenever the weight tensor changes (optimizer step, load_state_dict).  The encoders build all of theirs at once
    (_prepare_packs at the top of forward); a convolution used on its own gets here with a stale cache and packs itself."""
    c = conv.__dict__.get("_fs_packs")
    if c is None or c[0] != _conv_key(conv):
        _prepare_packs(conv)
        c = conv.__dict__["_fs_packs"]
    return c


class _ConvCL(torch.autograd.Function):
    """Stride-1 'same' convolution (1x1 or 3x3) of a channels_last tensor on the update block's implicit-GEMM kernels
    (csrc/conv_igemm.hip): forward and data gradient on fsraft_conv_forward, weight (+ bias) gradient on
    fsraft_conv_wgrad -- whose few-channel variant packs several taps into one tile for these 64/96-channel layers.
    The storage of a channels_last [B,C,H,W] tensor IS the kernels' [B,H,W,C] layout, so nothing is transposed."""

    @staticmethod
    def forward(ctx, x, weight, bias, packs, link=None, sums=None):
        from .. import ops
        ctx.link = link if _is_cl(x) else None       # (a converted copy of x is not the tensor the shortcut gradient belongs to)
        if ctx.link is not None:
            ctx.link.armed = True                    # this node's backward will collect the parked shortcut gradient
        x = _as_cl(x)
        B, C, H, W = x.shape
        N, _, KH, KW = weight.shape
        out = torch.empty(B, H, W, N, device=x.device, dtype=torch.float32)
        # sums (a _NormSums holder): the InstanceNorm behind this convolution wants the per-image column sums of the result; the
        # kernel adds them up in its epilogue where it can, and the norm then skips its own statistics pass
        acc = ops.zeros(2, B * 8, N, device=x.device) if (sums is not None and bias is None and STATS_IN_EPILOGUE) else None
        carried = ops.conv_forward([ops.V(x.permute(0, 2, 3, 1), C)], packs[1], bias, B, H, W, KH, KW, N, [ops.Dst.nhwc(out)],
                                   wpk_split=packs[2], wpk_frag=packs[5], stats=acc)
        if sums is not None:
            sums.acc = acc if carried else None
        ctx.save_for_backward(x, weight)
        ctx.packs = packs
        ctx.has_bias = bias is not None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        x, weight = ctx.saved_tensors
        B, C, H, W = x.shape
        N, _, KH, KW = weight.shape
        g = _as_cl(g)
        gv = ops.V(g.permute(0, 2, 3, 1), N)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dres = ctx.link.dres if ctx.link is not None else None
            if dres is not None and dres.shape == x.shape and _is_cl(dres) and dres.dtype == torch.float32:
                # residual unit: this data gradient is added into the parked shortcut gradient, the sum is x's gradient (see _ResLink)
                ctx.link.dres = None
                ops.conv_forward([gv], ctx.packs[3], None, B, H, W, KH, KW, C, [ops.Dst.nhwc(dres.permute(0, 2, 3, 1), acc=True)],
                                 wpk_split=ctx.packs[4], wpk_frag=ctx.packs[6])
                dx = dres
            else:
                if dres is not None:          # (a parked gradient this route cannot add into: hand both over, the sum is made here)
                    ctx.link.dres = None
                dxb = torch.empty(B, H, W, C, device=x.device, dtype=torch.float32)
                ops.conv_forward([gv], ctx.packs[3], None, B, H, W, KH, KW, C, [ops.Dst.nhwc(dxb)], wpk_split=ctx.packs[4],
                                 wpk_frag=ctx.packs[6])
                dx = dxb.permute(0, 3, 1, 2)
                if dres is not None:
                    dx = dx + dres
        elif ctx.link is not None:
            ctx.link.dres = None              # (nobody asked for x's gradient: autograd.grad w.r.t. other inputs)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            dwpk = ops.zeros(N, ops.conv_ktot([C], KH, KW), device=x.device)
            db = ops.zeros(N, device=x.device) if want_b else None
            ops.conv_wgrad(gv, [ops.V(x.permute(0, 2, 3, 1), C)], dwpk, B, H, W, KH, KW, dbias=db)
            dw = ops.unpack_weight_grad(dwpk, tuple(weight.shape), [C])
        return dx, dw, db, None, None, None


class _StemFn(torch.autograd.Function):
    """conv1 of the encoders (7x7, stride 2, 3 input channels; pytorch/core/extractor.py:135, :212) on csrc/stem.hip: NCHW image
    in, channels_last activation out; backward is the weight gradient only (the image is the network input)."""

    @staticmethod
    def forward(ctx, x, w):
        from .. import ops
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.stem_fwd(x, w.detach()).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        (x,) = ctx.saved_tensors
        return None, ops.stem_wgrad(x, _as_cl(g).permute(0, 2, 3, 1))


def _stem_ok(conv, x):
    import os
    return (STEM_KERNEL and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros" and conv.in_channels == 3
            and conv.out_channels in (32, 64) and _fast(x) and not x.requires_grad and not torch.is_autocast_enabled()
            and x.shape[2] >= 7 and x.shape[3] >= 7)


STATS_IN_EPILOGUE = True   # False: every InstanceNorm runs its own statistics pass
STEM_KERNEL = True         # False: the 7x7 stride-2 stem as a MIOpen call
S2D_UNITS = True           # False: the stride-2 units fall back to MIOpen behind layout hops (the path odd-sized inputs take)


class _NormSums:
    """Hand-over between a convolution and the InstanceNorm behind it: `acc` = the [2, B * 8, C] partial rows of the
    result's column sums / sums of squares when the convolution's kernel accumulated them in its epilogue, else None."""
    __slots__ = ("acc",)

    def __init__(self):
        self.acc = None


def _conv(conv, x, bias, link=None, stem_cl=False, sums=None):
    """conv(x) with the given bias (None: without).  A channels_last fp32 input of a stride-1 1x1 / 3x3 convolution takes
    the fsraft kernels; anything else is MIOpen on an NCHW tensor (its NHWC fp32 kernels are far slower than its NCHW
    ones on gfx950 -- the backward-weights one by two orders of magnitude -- so a channels_last input is converted)."""
    k = conv.kernel_size
    if (_is_cl(x) and x.is_cuda and x.dtype == torch.float32 and k in ((3, 3), (1, 1)) and conv.stride == (1, 1)
            and conv.padding == (k[0] // 2, k[1] // 2) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding_mode == "zeros" and x.shape[1] % 4 == 0 and not torch.is_autocast_enabled()):
        return _ConvCL.apply(x, conv.weight, bias, _weight_packs(conv), link, sums)
    if stem_cl and bias is None and _stem_ok(conv, x):
        return _StemFn.apply(x, conv.weight)
    return F.conv2d(_ToNCHW.apply(x) if _is_cl(x) else x, conv.weight, bias, conv.stride, conv.padding, conv.dilation, conv.groups)


_S2D_TAP = ((0, 1), (1, 0), (1, 1))          # 3x3 tap index k -> (2x2 tap t, sub-pixel s): input row 2y - 1 + k = 2(y + t - 1) + s


def _s2d_weight(w3):
    """[N,C,3,3] stride-2 weights -> the equivalent stride-1 [N,4C,2,2] weights over the space-to-depth input (channel
    (sy*2+sx)*C + c = pixel (2y+sy, 2x+sx)); 7 of the 16 (tap, sub-pixel) slots stay zero.  Host statement of the rewrite that
    fsraft_pack_conv_weights performs in its address arithmetic (flag bit 1); the tests hold the kernel to it."""
    N, C = w3.shape[:2]
    w = w3.new_zeros(N, 2, 2, C, 2, 2)           # n, sy, sx, c, ty, tx
    for ky, (ty, sy) in enumerate(_S2D_TAP):
        for kx, (tx, sx) in enumerate(_S2D_TAP):
            w[:, sy, sx, :, ty, tx] = w3[:, :, ky, kx]
    return w.reshape(N, 4 * C, 2, 2)


def _s2d_weight_grad(dw, C):
    """Gradient of _s2d_weight: [N,4C,2,2] -> [N,C,3,3]."""
    N = dw.shape[0]
    d = dw.reshape(N, 2, 2, C, 2, 2)
    out = dw.new_empty(N, C, 3, 3)
    for ky, (ty, sy) in enumerate(_S2D_TAP):
        for kx, (tx, sx) in enumerate(_S2D_TAP):
            out[:, :, ky, kx] = d[:, sy, sx, :, ty, tx]
    return out


def _pair_shape_ok(block):
    """A unit whose first convolution is 3x3 / stride 2 / pad 1 with a 1x1 / stride 2 shortcut: the pair that runs on the
    stride-1 kernels over a space-to-depth copy of the input."""
    c1 = getattr(block, "conv1", None)
    ds = block.downsample[0] if getattr(block, "downsample", None) is not None else None
    return (block.n == 2 and c1 is not None and ds is not None and c1.kernel_size == (3, 3) and c1.stride == (2, 2)
            and c1.padding == (1, 1) and c1.dilation == (1, 1) and c1.groups == 1 and ds.kernel_size == (1, 1)
            and ds.stride == (2, 2) and ds.padding == (0, 0) and ds.groups == 1)


def _pair_packs(block):
    """Packed matrices of a stride-2 residual unit's first convolution (as 2x2 over space-to-depth: the rewrite
    `input row 2y - 1 + k = 2(y + t - 1) + s` maps 3x3 tap k to (2x2 tap t, sub-pixel s), 7 of the 16 (tap, sub-pixel) slots
    are structural zeros) and of its 1x1 shortcut: `(key, (fwd fp32, fwd split, dgrad fp32, dgrad split) x 2)`."""
    from .. import ops
    w3, wsc = block.conv1.weight, block.downsample[0].weight
    key = (w3.data_ptr(), w3._version, wsc.data_ptr(), wsc._version, ops.exact_mode())
    c = block.__dict__.get("_fs_pair_packs")
    if c is None or c[0] != key:
        _prepare_packs(block)
        c = block.__dict__["_fs_pair_packs"]
    return c


class _StridedPairFn(torch.autograd.Function):
    """First convolution (3x3, stride 2, pad 1) and shortcut (1x1, stride 2) of a stride-2 ResidualBlock
    (pytorch/core/extractor.py:13, 39) on the stride-1 kernels: the channels_last input is permuted once to
    space-to-depth order ([B,H/2,W/2,4C], fsraft_space_to_depth2); over that tensor the 3x3 becomes a 2x2 convolution
    with 4C input channels (pad 1 forward, pad 0 in the data gradient) and the shortcut a 1x1 over its first C channels.
    Both data gradients land in one [B,H/2,W/2,4C] buffer (the shortcut's accumulates into channels [0, C)) that is
    permuted back once.  No MIOpen call, no NCHW hop.  Biases are the caller's business (dropped before InstanceNorm,
    folded into a frozen BatchNorm)."""

    @staticmethod
    def forward(ctx, x, w3, wsc, packs, pre=False):
        from .. import ops
        x = _as_cl(x)
        ctx.pre = pre
        if pre:         # x already IS the space-to-depth tensor [B, 4C, H/2, W/2] (written by the norm before it: _S2D)
            B, C4, h, w = x.shape
            C, H, W = C4 // 4, 2 * h, 2 * w
            xs = x.permute(0, 2, 3, 1)
        else:
            B, C, H, W = x.shape
            h, w = H // 2, W // 2
            xs = ops.space_to_depth2(x.permute(0, 2, 3, 1))
        N, Ns = w3.shape[0], wsc.shape[0]
        y1 = torch.empty(B, h, w, N, device=x.device, dtype=torch.float32)
        ys = torch.empty(B, h, w, Ns, device=x.device, dtype=torch.float32)
        p, q = packs[1], packs[2]
        ops.conv_forward([ops.V(xs, 4 * C)], p[0], None, B, h, w, 2, 2, N, [ops.Dst.nhwc(y1)], wpk_split=p[1])
        ops.conv_forward([ops.V(xs, C, 0)], q[0], None, B, h, w, 1, 1, Ns, [ops.Dst.nhwc(ys)], wpk_split=q[1])
        ctx.save_for_backward(xs, w3, wsc)
        ctx.packs = packs
        return y1.permute(0, 3, 1, 2), ys.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g1, gs):
        from .. import ops
        xs, w3, wsc = ctx.saved_tensors
        B, h, w, C4 = xs.shape
        C, N, Ns = C4 // 4, w3.shape[0], wsc.shape[0]
        p, q = ctx.packs[1], ctx.packs[2]
        g1v = ops.V(_as_cl(g1).permute(0, 2, 3, 1), N)
        gsv = ops.V(_as_cl(gs).permute(0, 2, 3, 1), Ns)
        dx = dw3 = dwsc = None
        if ctx.needs_input_grad[0]:
            dxs = torch.empty(B, h, w, C4, device=xs.device, dtype=torch.float32)
            ops.conv_forward([g1v], p[2], None, B, h, w, 2, 2, C4, [ops.Dst.nhwc(dxs)], wpk_split=p[3], pad=(0, 0))
            ops.conv_forward([gsv], q[2], None, B, h, w, 1, 1, C, [ops.Dst.nhwc(dxs, 0, 0, True)], wpk_split=q[3])
            dx = (dxs if ctx.pre else ops.space_to_depth2(dxs, inverse=True)).permute(0, 3, 1, 2)
        items = []
        if ctx.needs_input_grad[1]:
            dwpk = ops.zeros(N, ops.conv_ktot([C4], 2, 2), device=xs.device)
            ops.conv_wgrad(g1v, [ops.V(xs, C4)], dwpk, B, h, w, 2, 2)
            dw3 = torch.empty(N, C, 3, 3, device=xs.device, dtype=torch.float32)
            items.append((dwpk, [dw3], [C4], [0], C, 2, 2, 1.0, True))       # gradient of the space-to-depth rewrite: a gather
        if ctx.needs_input_grad[2]:
            dwpk = ops.zeros(Ns, ops.conv_ktot([C], 1, 1), device=xs.device)
            ops.conv_wgrad(gsv, [ops.V(xs, C, 0)], dwpk, B, h, w, 1, 1)
            dwsc = torch.empty(Ns, C, 1, 1, device=xs.device, dtype=torch.float32)
            items.append((dwpk, [dwsc], [C], [0], C, 1, 1, 1.0, False))
        ops.unpack_weight_grads(items, xs.device)
        return dx, dw3, dwsc, None, None


def _norm_act(norm, y, cbias, relu, res=None):
    """relu?(norm(y + cbias)) (+ fused residual) for a channels_last convolution output and one of the two norm kinds the
    channels_last path covers; cbias is the bias the convolution ran without."""
    if isinstance(norm, nn.InstanceNorm2d):
        return _InstNormReluCL.apply(y, norm.eps, relu, res)
    return _FrozenBNReluCL.apply(y, cbias, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.eps, relu, res)


def _pair_ok_shape(block, C, H, W):
    import os
    return (S2D_UNITS and _pair_shape_ok(block) and H % 2 == 0
            and W % 2 == 0 and C % 4 == 0 and _cl_norm_ok(torch.empty(0, block.conv1.out_channels)))


def _pair_ok(block, x):
    return _pair_ok_shape(block, x.shape[1], x.shape[2], x.shape[3])


S2D_EMIT = True   # False: the stride-2 units copy their input into the space-to-depth layout themselves


class _S2D:
    """A residual unit's output handed to the stride-2 unit behind it AS the space-to-depth tensor that unit's two convolutions
    read ([B, 4C, H/2, W/2] channels_last; sub-pixel (sy, sx) of pixel (2y + sy, 2x + sx) in channels (2 sy + sx) C ..): the
    unit's last norm kernel writes it in that layout and reads the gradient from it, so neither direction needs a copy."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


def _conv_norm(conv, norm, x, relu, to_cl=False, res=None, link=None, res_link=None, s2d=False):
    """relu?(norm(conv(x))) of pytorch/core/extractor.py.  The convolution stays a PyTorch-ROCm (MIOpen) call; for fp32
    CUDA tensors the normalisation + ReLU around it runs on the fused fsraft kernels:
      * non-affine InstanceNorm2d (feature encoder).  A per-channel constant added before it is removed again by its
        mean subtraction, so the convolution runs without its bias: same output (to rounding), one bias-add kernel less
        forward and one [N,H,W] reduction less backward.  The bias then receives no gradient (mathematically it is
        exactly zero; parallel.FlatGradients keeps a zero for it, so the optimizer treats it as the reference does);
      * BatchNorm2d using running statistics (context encoder after freeze_bn, or eval mode).
    Everything else (training-mode BatchNorm, GroupNorm, autocast dtypes, CPU) is the framework's own modules.
    A channels_last input (or to_cl) keeps the chain channels_last: the convolution output is converted if the library
    handed back NCHW, and the [N][HW][C] twins of the norm kernels are used."""
    fused = x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
    if isinstance(norm, nn.InstanceNorm2d) and not norm.affine and not norm.track_running_stats:
        if fused:
            sums = _NormSums() if (to_cl or _is_cl(x)) else None
            y = _conv(conv, x, None, link, stem_cl=to_cl, sums=sums)
            if (to_cl or _is_cl(x)) and _cl_norm_ok(y):
                return _InstNormReluCL.apply(y, norm.eps, relu, res, res_link, sums, s2d)
            assert not s2d
            y = _InstNormRelu.apply(y, norm.eps, relu)
            return y if res is None else F.relu(res + y)
        y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups) if conv.bias is not None else conv(x)
        y = norm(y)
    elif isinstance(norm, nn.BatchNorm2d) and not norm.training and norm.track_running_stats and norm.affine and fused:
        y = _conv(conv, x, None, link, stem_cl=to_cl)
        if (to_cl or _is_cl(x)) and _cl_norm_ok(y):
            return _FrozenBNReluCL.apply(y, conv.bias, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.eps, relu, res,
                                         res_link, s2d)
        assert not s2d
        y = _FrozenBNRelu.apply(y, conv.bias, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.eps, relu)
        return y if res is None else F.relu(res + y)
    else:
        y = norm(conv(x))
    assert not s2d, "space-to-depth output is a feature of the fused channels_last norm kernels"
    y = F.relu(y, inplace=True) if relu else y
    return y if res is None else F.relu(res + y)


def _make_norm(kind, ch, groups):
    if kind == "group":
        return nn.GroupNorm(num_groups=groups, num_channels=ch)
    if kind == "batch":
        return nn.BatchNorm2d(ch)
    if kind == "instance":
        return nn.InstanceNorm2d(ch)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(kind)


class _Block(nn.Module):
    """Residual unit.  `widths` lists (out_channels, kernel) of its convolutions; the first 3x3
    carries the stride.  The shortcut norm is the same module object as downsample[1], which is
    why it also appears under its own name (norm3 / norm4) in reference checkpoints."""

    def __init__(self, cin, cout, norm_fn, stride, bottleneck):
        super().__init__()
        mid = cout // 4
        if bottleneck:
            spec = [(cin, mid, 1, 1), (mid, mid, 3, stride), (mid, cout, 1, 1)]
        else:
            spec = [(cin, cout, 3, stride), (cout, cout, 3, 1)]
        groups = cout // 8
        self.n = len(spec)
        for i, (a, b, k, s) in enumerate(spec, 1):
            setattr(self, f"conv{i}", nn.Conv2d(a, b, kernel_size=k, padding=k // 2, stride=s))
        for i, (a, b, k, s) in enumerate(spec, 1):
            setattr(self, f"norm{i}", _make_norm(norm_fn, b, groups))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1:
            short = _make_norm(norm_fn, cout, groups)
            setattr(self, f"norm{self.n + 1}", short)
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, stride=stride), short)

    def forward(self, x, emit_for=None):
        """emit_for: the stride-2 unit that consumes this unit's output (the encoder's channels_last walk passes it): the output then
        leaves as an _S2D holder where that unit can take it."""
        if isinstance(x, _S2D):
            y1, ys = _StridedPairFn.apply(x.t, self.conv1.weight, self.downsample[0].weight, _pair_packs(self), True)
            y = _norm_act(self.norm1, y1, self.conv1.bias, True)
            xs = _norm_act(self.downsample[1], ys, self.downsample[0].bias, False)
            return _conv_norm(self.conv2, self.norm2, y, True, to_cl=True, res=xs)
        cl = _is_cl(x)
        if cl and self.downsample is not None and _pair_ok(self, x):
            # stride-2 unit, channels_last: both strided convolutions on the stride-1 kernels over the space-to-depth input
            y1, ys = _StridedPairFn.apply(x, self.conv1.weight, self.downsample[0].weight, _pair_packs(self))
            y = _norm_act(self.norm1, y1, self.conv1.bias, True)
            xs = _norm_act(self.downsample[1], ys, self.downsample[0].bias, False)
            return _conv_norm(self.conv2, self.norm2, y, True, to_cl=True, res=xs)
        if cl and self.downsample is not None:
            x = _ToNCHW.apply(x)        # the strided convolutions (first 3x3 / shortcut 1x1) are MIOpen NCHW calls: one copy for both
        y = x
        # stride-1 unit on the channels_last path: x feeds the first convolution AND the shortcut -- their two gradients are
        # merged inside the convolution's data-gradient epilogue instead of by an autograd add (_ResLink)
        link = _ResLink() if (cl and self.downsample is None and self.n >= 2 and torch.is_grad_enabled() and x.requires_grad) else None
        if self.downsample is not None:
            x = _conv_norm(self.downsample[0], self.downsample[1], x, False, to_cl=cl)
        for i in range(1, self.n):
            y = _conv_norm(getattr(self, f"conv{i}"), getattr(self, f"norm{i}"), y, True, to_cl=cl, link=link if i == 1 else None)
        # last convolution of the unit: relu(x + relu(norm(conv(y)))), the add and outer ReLU fused into the norm kernel
        last = getattr(self, f"conv{self.n}")
        emit = bool(cl and S2D_EMIT and emit_for is not None and self.downsample is None and _is_cl(y) and _cl_norm_ok(torch.empty(0, last.out_channels))
                    and _pair_ok_shape(emit_for, last.out_channels, y.shape[2], y.shape[3])
                    and isinstance(getattr(self, f"norm{self.n}"), (nn.InstanceNorm2d, nn.BatchNorm2d)))
        out = _conv_norm(last, getattr(self, f"norm{self.n}"), y, True, to_cl=cl, res=x, res_link=link, s2d=emit)
        return _S2D(out) if emit else out


class ResidualBlock(_Block):
    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__(in_planes, planes, norm_fn, stride, bottleneck=False)


class BottleneckBlock(_Block):
    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__(in_planes, planes, norm_fn, stride, bottleneck=True)


def _channels_last_ok(enc, x):
    """Channels_last encoder path: fp32 CUDA, no autocast, norms that the fused channels_last kernels cover (non-affine
    InstanceNorm, BatchNorm on running statistics).  FSRAFT_ENCODER_CL=0 keeps the encoder NCHW on MIOpen throughout."""
    import os
    mode = int(os.environ.get("FSRAFT_ENCODER_CL", "1"))
    if mode == 0:
        return 0
    if not (x.is_cuda and x.dtype == torch.float32) or torch.is_autocast_enabled():
        return 0
    n = enc.norm1
    if isinstance(n, nn.InstanceNorm2d):
        return mode if not n.affine and not n.track_running_stats else 0
    if isinstance(n, nn.BatchNorm2d):
        return mode if not n.training and n.track_running_stats and n.affine else 0
    return 0


class _Encoder(nn.Module):
    def __init__(self, widths, block, stem_groups, output_dim, norm_fn, dropout):
        super().__init__()
        self.norm_fn = norm_fn
        c0, c1, c2 = widths
        self.norm1 = _make_norm(norm_fn, c0, stem_groups)
        self.conv1 = nn.Conv2d(3, c0, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(block(c0, c0, norm_fn, 1), block(c0, c0, norm_fn, 1))
        self.layer2 = nn.Sequential(block(c0, c1, norm_fn, 2), block(c1, c1, norm_fn, 1))
        self.layer3 = nn.Sequential(block(c1, c2, norm_fn, 2), block(c2, c2, norm_fn, 1))
        self.conv2 = nn.Conv2d(c2, output_dim, kernel_size=1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    @on_tensor_device
    def forward(self, x):
        pair = isinstance(x, (tuple, list))
        if pair:
            n = x[0].shape[0]
            x = torch.cat(x, dim=0)
        # channels_last after the stem (an NCHW MIOpen call: 3 input channels): the stride-1 convolutions then run on the
        # fsraft implicit-GEMM kernels and the norm + ReLU kernels read and write [N][HW][C] directly
        cl = _channels_last_ok(self, x)
        if cl:
            _prepare_packs(self)         # all packed weight images of this encoder that are stale: one batched launch
        x = _conv_norm(self.conv1, self.norm1, x, True, to_cl=bool(cl))
        if cl:
            # (the Sequential containers walked by hand: a unit in front of a stride-2 unit is told so, see _S2D)
            stages = (self.layer1, self.layer2, self.layer3)
            for si, seq in enumerate(stages):
                units = list(seq)
                for ui, unit in enumerate(units):
                    nxt = stages[si + 1][0] if (ui == len(units) - 1 and si + 1 < len(stages)) else None
                    x = unit(x, nxt) if isinstance(unit, _Block) else unit(x)
        else:
            x = self.layer3(self.layer2(self.layer1(x)))
        if cl:
            x = _conv(self.conv2, x, self.conv2.bias)
            # out_channels_last (set by the models on their context encoder): the result stays a channels_last tensor -- its
            # consumers (split, tanh / ReLU, update.to_channels_last) work on it in place of three layout copies per direction
            if pair or not getattr(self, "out_channels_last", False) or not _is_cl(x):
                x = _ToNCHW.apply(x)
        else:
            x = self.conv2(x)
        if self.training and self.dropout is not None:
            x = self.dropout(x)
        if pair:
            x = torch.split(x, [n, n], dim=0)
        return x


class BasicEncoder(_Encoder):
    """extractor.py:118-192: 64/96/128 residual stages to 1/8 resolution."""

    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0):
        super().__init__((64, 96, 128), ResidualBlock, 8, output_dim, norm_fn, dropout)


class SmallEncoder(_Encoder):
    """extractor.py:195-267: 32/64/96 bottleneck stages."""

    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0):
        super().__init__((32, 64, 96), BottleneckBlock, 8, output_dim, norm_fn, dropout)
