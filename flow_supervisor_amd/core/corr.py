"""Correlation blocks on HIP kernels (rows a1-a5 of SURVEY.md section 8).

Drop-in for pytorch/core/corr.py: ``CorrBlock(fmap1, fmap2, num_levels=4, radius=4)`` does the
one-time all-pairs volume + pyramid build in its constructor and ``obj(coords)`` returns the
``[B, L*(2r+1)^2, H, W]`` lookup (corr.py:13-50); ``AlternateCorrBlock`` computes the same
numbers without the N x N volume (corr.py:63-91 + alt_cuda_corr).

Autograd design (differs from the reference on purpose): the reference lets autograd
allocate a dense zero gradient for the whole volume on every one of the 4x12 grid_sample
backward calls.  Here a lookup's backward only keeps (coords, dOut) -- the window gradients
are a pure function of them -- and when autograd reaches the build node the gradient volume
is written ONCE (fsraft_corr_dvol_build: every query row accumulated in LDS over all
lookups of the step, no zero fill, no read-modify-write) and two GEMMs that contract over
whole rows produce dF1 / dF2; the pooling chain's backward runs on the 7 MB feature
gradient, not on the 1 GB volume gradient.  A 1-element "anchor" tensor threads the
dependency through the autograd graph.  Coordinates get no gradient: every caller
detaches them first (raft.py:123).
"""
import math

import torch
import torch.nn.functional as F

from .. import ops
from .utils.utils import coords_grid


class _GradState:
    """Backward state of one CorrBlock: the (coords, dOut) pairs of every lookup whose gradient has arrived.  The window
    gradients are a pure function of those, so nothing else is done until autograd reaches the volume build."""
    __slots__ = ("stash", "is_flow")

    def __init__(self, device):
        self.stash = []
        self.is_flow = False                    # stash entries hold flows (pixel grid added by the kernel), not coordinates


def _build_records(fmap1, fmap2):
    """(f1r, f2r) pixel-major records of the feature maps for the record-core build, or None when that build does not apply
    (exact-fp32 test mode, C not a multiple of 32)."""
    if not (ops.BUILD_REC and ops.SPLIT_VOLUME_BWD and fmap1.shape[1] % 32 == 0 and fmap1.is_cuda):
        return None
    return ops.fmap_records(fmap1), ops.fmap_records(fmap2)


class _BuildFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap1, fmap2, num_levels, radius, holder):
        recs = _build_records(fmap1, fmap2)
        vol, lay = ops.corr_build_tiled(fmap1, fmap2, num_levels, recs=recs)
        ctx.f1r = recs[0] if recs is not None else None
        state = _GradState(fmap1.device)
        holder.append((state, lay))
        ctx.state, ctx.lay, ctx.radius = state, lay, radius
        ctx.save_for_backward(fmap1, fmap2)
        anchor = ops.zeros(1, device=fmap1.device)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(vol)
        return anchor, vol

    @staticmethod
    def backward(ctx, ganchor, gvol):
        fmap1, fmap2 = ctx.saved_tensors
        st = ctx.state
        if not st.stash:                            # no lookup contributed a gradient
            return torch.zeros_like(fmap1), torch.zeros_like(fmap2), None, None, None
        stash, st.stash = st.stash, []
        rec = ops.SPLIT_VOLUME_BWD             # (off = the exact-fp32 test mode: fp32 gradient volume, exact GEMMs)
        dvol = ops.corr_dvol_build([d for _, d in stash], [c for c, _ in stash], ctx.lay, fmap1.shape[0], ctx.radius, records=rec,
                                   is_flow=st.is_flow)
        del stash
        d1, d2 = ops.corr_build_bwd_tiled(fmap1, fmap2, dvol, ctx.lay, records=rec, f1r=ctx.f1r if rec else None)
        ctx.f1r = None
        return d1, d2, None, None, None


class _LookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, coords, block, channels_last, is_flow):
        out = ops.corr_lookup_tiled_fwd(block._vol, block._lay, coords, block.radius, is_flow)
        ctx.state = block._state
        ctx.cl, ctx.is_flow = channels_last, is_flow
        ctx.save_for_backward(coords)
        return out if channels_last else ops.nhwc_to_nchw(out)

    @staticmethod
    def backward(ctx, dout):
        (coords,) = ctx.saved_tensors
        dout = dout.contiguous() if ctx.cl else ops.nchw_to_nhwc(dout)
        if ctx.is_flow != ctx.state.is_flow:       # (a block looked up both ways: keep one convention in the stash)
            B, _, H, W = coords.shape
            g = coords_grid(B, H, W, device=coords.device)
            coords = coords - g if ctx.state.is_flow else coords + g
        ctx.state.stash.append((coords, dout))
        return None, None, None, None, None   # (no gradient tensor for the anchor: the build node still runs after every lookup)


class CorrBlock:
    """All-pairs volume + pyramid + lookup (pytorch/core/corr.py:13-60).  The volume lives in the tiled-row layout of
    csrc/corr_layout.hpp (one row per query, all levels, 4x4-cell tiles); `corr_pyramid` -- the reference's list of
    [B*H*W, 1, h_l, w_l] tensors -- is materialised from it on first access (API edge; nothing on the path reads it)."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        if not 1 <= num_levels <= 4:
            raise NotImplementedError("the HIP correlation kernels are built for 1..4 pyramid levels (all RAFT variants use 4)")
        if radius not in (3, 4):
            raise NotImplementedError("the HIP lookup kernels are built for radius 3 (raft-small) and 4 (RAFT)")
        self.num_levels = num_levels
        self.radius = radius
        fmap1 = fmap1.float()
        fmap2 = fmap2.float()
        self._pyr = None
        self._tracks_grad = torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad)
        if self._tracks_grad:
            holder = []
            self._anchor, self._vol = _BuildFn.apply(fmap1, fmap2, num_levels, radius, holder)
            self._state, self._lay = holder[0]
        else:
            self._vol, self._lay = ops.corr_build_tiled(fmap1, fmap2, num_levels, recs=_build_records(fmap1, fmap2))
            self._anchor, self._state = None, None

    @property
    def corr_pyramid(self):
        """[B*H*W, 1, h_l, w_l] per level, as in corr.py:19-27 (copies out of the tiled rows)."""
        if self._pyr is None:
            self._pyr = [self._lay.level_view(self._vol, l) for l in range(self.num_levels)]
        return self._pyr

    def __call__(self, coords, channels_last=False, is_flow=False):
        """coords [B,2,H,W] (x,y).  Returns [B, L*(2r+1)^2, H, W] contiguous (or [B,H,W,C] when
        channels_last=True, the layout our update block consumes directly).  is_flow=True: the tensor holds the flow and
        the lookup is centred on pixel grid + flow (what the RAFT loop passes: it never forms coords1)."""
        coords = coords.float()
        if self._tracks_grad and torch.is_grad_enabled():
            if not self._state.stash:
                self._state.is_flow = is_flow
            return _LookupFn.apply(self._anchor, coords.detach(), self, channels_last, is_flow)
        out = ops.corr_lookup_tiled_fwd(self._vol, self._lay, coords, self.radius, is_flow)
        return out if channels_last else ops.nhwc_to_nchw(out)

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
