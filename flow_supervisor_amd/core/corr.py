"""Correlation blocks on HIP kernels (rows a1-a5 of SURVEY.md section 8).

Drop-in for pytorch/core/corr.py: ``CorrBlock(fmap1, fmap2, num_levels=4, radius=4)`` does the
one-time all-pairs volume + pyramid build in its constructor and ``obj(coords)`` returns the
``[B, L*(2r+1)^2, H, W]`` lookup (corr.py:13-50); ``AlternateCorrBlock`` computes the same
numbers without the N x N volume (corr.py:63-91 + alt_cuda_corr).

Autograd design (differs from the reference on purpose): the reference lets autograd
allocate a dense zero gradient for the whole volume on every one of the 4x12 grid_sample
backward calls.  Here each CorrBlock owns ONE gradient pyramid that all lookups of a step
accumulate into in place (each query owns its slice, so no atomics), and the volume
backward (un-pool + two fp32-MFMA GEMMs) runs once, when autograd reaches the build node.
A 1-element "anchor" tensor threads that dependency through the autograd graph.
Coordinates get no gradient: every caller detaches them first (raft.py:123).
"""
import math

import torch
import torch.nn.functional as F

from .. import ops


class _GradState:
    """Accumulated dL/dV pyramid of one CorrBlock (allocated by the first lookup backward)."""
    __slots__ = ("dlevels", "shapes", "zero")

    def __init__(self, levels):
        self.dlevels = None
        self.shapes = [tuple(l.shape) for l in levels]
        self.zero = torch.zeros(1, device=levels[0].device)


class _BuildFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap1, fmap2, num_levels, holder):
        levels = ops.corr_build(fmap1, fmap2, num_levels)
        state = _GradState(levels)
        holder.append(state)
        ctx.state = state
        ctx.save_for_backward(fmap1, fmap2)
        anchor = torch.zeros(1, device=fmap1.device)
        ctx.mark_non_differentiable(*levels)
        return (anchor, *levels)

    @staticmethod
    def backward(ctx, ganchor, *glevels):
        fmap1, fmap2 = ctx.saved_tensors
        st = ctx.state
        if st.dlevels is None:                      # no lookup contributed a gradient
            return torch.zeros_like(fmap1), torch.zeros_like(fmap2), None, None
        dl, st.dlevels = st.dlevels, None
        d1, d2 = ops.corr_build_bwd(fmap1, fmap2, dl)
        return d1, d2, None, None


class _LookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, coords, block, channels_last):
        out = ops.corr_lookup_fwd(block.corr_pyramid, coords, block.radius, nhwc=channels_last)
        ctx.state = block._state
        ctx.radius = block.radius
        ctx.cl = channels_last
        ctx.save_for_backward(coords)
        return out

    @staticmethod
    def backward(ctx, dout):
        (coords,) = ctx.saved_tensors
        st = ctx.state
        if st.dlevels is None:
            st.dlevels = [torch.zeros(s, device=dout.device, dtype=torch.float32) for s in st.shapes]
        ops.corr_lookup_bwd_(st.dlevels, coords, dout, ctx.radius, nhwc=ctx.cl)
        return st.zero, None, None, None


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        if num_levels != 4:
            raise NotImplementedError("the HIP lookup kernels are built for 4 pyramid levels (all RAFT variants use 4)")
        if radius not in (3, 4):
            raise NotImplementedError("the HIP lookup kernels are built for radius 3 (raft-small) and 4 (RAFT)")
        self.num_levels = num_levels
        self.radius = radius
        fmap1 = fmap1.float()
        fmap2 = fmap2.float()
        self._tracks_grad = torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad)
        if self._tracks_grad:
            holder = []
            self._anchor, *levels = _BuildFn.apply(fmap1, fmap2, num_levels, holder)
            self._state = holder[0]
        else:
            levels = ops.corr_build(fmap1, fmap2, num_levels)
            self._anchor, self._state = None, None
        self.corr_pyramid = list(levels)            # [B*H*W, 1, h_l, w_l], as in corr.py:19-27

    def __call__(self, coords, channels_last=False):
        """coords [B,2,H,W] (x,y).  Returns [B, 4*(2r+1)^2, H, W] contiguous (or [B,H,W,C] when
        channels_last=True, the layout our update block consumes directly)."""
        coords = coords.float()
        if self._tracks_grad and torch.is_grad_enabled():
            return _LookupFn.apply(self._anchor, coords.detach(), self, channels_last)
        return ops.corr_lookup_fwd(self.corr_pyramid, coords, self.radius, nhwc=channels_last)

    @staticmethod
    def corr(fmap1, fmap2):
        """[B,H,W,1,H,W] all-pairs volume / sqrt(C) (corr.py:52-60); level 0 of the HIP build."""
        B, C, H, W = fmap1.shape
        lvl0 = ops.corr_build(fmap1.float(), fmap2.float(), 1)[0]
        return lvl0.view(B, H, W, 1, H, W)


class _AltCorrFn(torch.autograd.Function):
    """alt_cuda_corr.forward with the backward the reference compiled but never wired (corr.py:74-91)."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, coords, radius):
        ctx.save_for_backward(fmap1, fmap2, coords)
        ctx.radius = radius
        return ops.altcorr_fwd(fmap1, fmap2, coords, radius)

    @staticmethod
    def backward(ctx, g):
        fmap1, fmap2, coords = ctx.saved_tensors
        g1, g2, _ = ops.altcorr_bwd(fmap1, fmap2, coords, g.contiguous(), ctx.radius)
        return g1, g2, None, None


class AlternateCorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        fmap1 = fmap1.float()
        fmap2 = fmap2.float()
        self.pyramid = [(fmap1, fmap2)]
        for _ in range(self.num_levels):
            fmap1 = F.avg_pool2d(fmap1, 2, stride=2)
            fmap2 = F.avg_pool2d(fmap2, 2, stride=2)
            self.pyramid.append((fmap1, fmap2))
        # channels-last copies once per pair instead of once per level per iteration (corr.py:82-83)
        self._f1 = self.pyramid[0][0].permute(0, 2, 3, 1).contiguous()
        self._f2 = [self.pyramid[i][1].permute(0, 2, 3, 1).contiguous() for i in range(self.num_levels)]

    def __call__(self, coords):
        coords = coords.permute(0, 2, 3, 1)
        B, H, W, _ = coords.shape
        dim = self._f1.shape[-1]
        outs = []
        for i in range(self.num_levels):
            ci = (coords / 2 ** i).reshape(B, 1, H, W, 2).contiguous()
            outs.append(_AltCorrFn.apply(self._f1, self._f2[i], ci, self.radius).squeeze(1))
        corr = torch.stack(outs, dim=1).reshape(B, -1, H, W)
        return corr / math.sqrt(float(dim))
