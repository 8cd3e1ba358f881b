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
from .._lib import on_tensor_device
from .utils.utils import coords_grid


class _GradState:
    """Backward state of one CorrBlock: the (coords, dOut) pairs of every lookup whose gradient has arrived.  The window
    gradients are a pure function of those, so nothing else is done until autograd reaches the volume build."""
    __slots__ = ("stash", "is_flow", "gs")

    def __init__(self, device):
        self.stash = []
        self.is_flow = False                    # stash entries hold flows (pixel grid added by the kernel), not coordinates
        self.gs = None                          # CorrBlock(grad_samples=k): only the first k samples' lookups receive gradient


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
        Bf, f1r = fmap1.shape[0], ctx.f1r
        sliced = st.gs is not None and 0 < st.gs < Bf
        if sliced:       # the other samples' window gradients are zeros by the caller's word: build and contract the first k only
            k = st.gs
            full1, full2 = fmap1, fmap2
            fmap1, fmap2 = fmap1[:k], fmap2[:k]
            stash = [(c[:k], d[:k]) for c, d in stash]
            f1r = f1r[:k] if f1r is not None else None
        rec = ops.SPLIT_VOLUME_BWD             # (off = the exact-fp32 test mode: fp32 gradient volume, exact GEMMs)
        # which k-tiles of the two volume-backward GEMMs the step's lookups can reach: the GEMMs walk those only, and the gradient
        # volume is only written where they will read (the rest would be zero records)
        kt = ops.corr_bwd_ktiles([c for c, _ in stash], ctx.lay, fmap1.shape[0], ctx.radius, st.is_flow) if rec and ops.BWD_KSKIP else None
        dvol = ops.corr_dvol_build([d for _, d in stash], [c for c, _ in stash], ctx.lay, fmap1.shape[0], ctx.radius, records=rec,
                                   is_flow=st.is_flow, wmask=kt.wmask if kt is not None and ops.DVOL_WMASK else None)
        del stash
        d1, d2 = ops.corr_build_bwd_tiled(fmap1, fmap2, dvol, ctx.lay, records=rec, f1r=f1r if rec else None, ktiles=kt)
        ctx.f1r = None
        if sliced:
            z1, z2 = torch.zeros_like(full1), torch.zeros_like(full2)
            z1[:k].copy_(d1)
            z2[:k].copy_(d2)
            d1, d2 = z1, z2
        return d1, d2, None, None, None


class _LookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, coords, block, channels_last, is_flow, out_buf=None):
        out = ops.corr_lookup_tiled_fwd(block._vol, block._lay, coords, block.radius, is_flow, out=out_buf)
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
        return None, None, None, None, None, None   # (no gradient tensor for the anchor: the build node still runs after every lookup)


class CorrBlock:
    """All-pairs volume + pyramid + lookup (pytorch/core/corr.py:13-60).  The volume lives in the tiled-row layout of
    csrc/corr_layout.hpp (one row per query, all levels, 4x4-cell tiles); `corr_pyramid` -- the reference's list of
    [B*H*W, 1, h_l, w_l] tensors -- is materialised from it on first access (API edge; nothing on the path reads it).
    grad_samples (extension, default None): the caller vouches that only the lookups of the first k samples receive gradient
    (core/l2l.py: the supervisor phase of a batched flow-supervisor step); the backward then builds and contracts the gradient
    volume of those samples only."""

    @on_tensor_device
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, grad_samples=None):
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
            self._state.gs = grad_samples
        else:
            self._vol, self._lay = ops.corr_build_tiled(fmap1, fmap2, num_levels, recs=_build_records(fmap1, fmap2))
            self._anchor, self._state = None, None

    @property
    def corr_pyramid(self):
        """[B*H*W, 1, h_l, w_l] per level, as in corr.py:19-27 (copies out of the tiled rows)."""
        if self._pyr is None:
            self._pyr = [self._lay.level_view(self._vol, l) for l in range(self.num_levels)]
        return self._pyr

    @on_tensor_device
    def __call__(self, coords, channels_last=False, is_flow=False, out=None):
        """coords [B,2,H,W] (x,y).  Returns [B, L*(2r+1)^2, H, W] contiguous (or [B,H,W,C] when
        channels_last=True, the layout our update block consumes directly).  is_flow=True: the tensor holds the flow and
        the lookup is centred on pixel grid + flow (what the RAFT loop passes: it never forms coords1).  out (channels_last
        only): a preallocated [B,H,W,C] buffer the lookup writes into (update.MotionBatch's slots)."""
        coords = coords.float()
        if out is not None and not channels_last:
            raise ValueError("out= is for channels_last lookups")
        if self._tracks_grad and torch.is_grad_enabled():
            if not self._state.stash:
                self._state.is_flow = is_flow
            if channels_last:
                return _LookupFn.apply(self._anchor, coords.detach(), self, True, is_flow, out)
            # the reference-shaped call: the lookup itself stays channels-last (an autograd node of its own); the NCHW-shaped tensor the
            # caller receives is a view of it (update.as_nchw) -- BasicUpdateBlock.forward continues from the original (update.to_channels_last)
            from .update import as_nchw
            return as_nchw(_LookupFn.apply(self._anchor, coords.detach(), self, True, is_flow, None))
        res = ops.corr_lookup_tiled_fwd(self._vol, self._lay, coords, self.radius, is_flow, out=out)
        if channels_last:
            return res
        from . import update as _u
        if _u.NCHW_VIEWS and _u.TWINS:
            return res.permute(0, 3, 1, 2)
        y = ops.nhwc_to_nchw(res)
        if _u.TWINS:
            y._fs_cl = (res, y._version)
        return y

    @staticmethod
    @on_tensor_device
    def corr(fmap1, fmap2):
        """[B,H,W,1,H,W] all-pairs volume / sqrt(C) (corr.py:52-60); level 0 of the HIP build."""
        B, C, H, W = fmap1.shape
        lvl0 = ops.corr_build(fmap1.float(), fmap2.float(), 1)[0]
        return lvl0.view(B, H, W, 1, H, W)


class _AltBuildFn(torch.autograd.Function):
    """(fmap1, fmap2) -> anchor.  Forward does nothing; backward -- reached once every lookup of the step has handed in its
    (coords, dOut) -- runs the volume backward a chunk of queries at a time (ops.corr_bwd_chunked): the gradient volume of
    `chunk` queries, two record GEMMs, no O(N^2) buffer and no atomics on the feature gradients."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, state, num_levels, radius):
        # (the node holds the block's STATE, not the block: block -> anchor -> grad_fn -> ctx -> block would be a reference cycle
        #  that keeps the feature records and the pooled pyramid waiting for the cyclic collector, 32 MB per step at 2 x 32x48)
        ctx.state, ctx.num_levels, ctx.radius = state, num_levels, radius
        ctx.save_for_backward(fmap1, fmap2)
        ctx.set_materialize_grads(False)
        return ops.zeros(1, device=fmap1.device)

    @staticmethod
    def backward(ctx, ganchor):
        fmap1, fmap2 = ctx.saved_tensors
        st = ctx.state
        stash, st.stash = st.stash, []
        if not stash:
            return torch.zeros_like(fmap1), torch.zeros_like(fmap2), None, None, None
        lay = ops.VolLayout.get(fmap1.shape[2], fmap1.shape[3], ctx.num_levels)
        d1, d2 = ops.corr_bwd_chunked(fmap1, fmap2, [d for _, d in stash], [c for c, _ in stash], lay, ctx.radius,
                                      is_flow=st.is_flow)
        return d1, d2, None, None, None


class _AltLookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, coords, block, channels_last, is_flow, out_buf=None):
        out = ops.altcorr_fused_fwd(block._f1, block._f2, coords, block.radius, is_flow, recs=block._recs, out=out_buf, regime=block._regime)
        ctx.state, ctx.cl, ctx.is_flow = block._state, channels_last, is_flow
        ctx.save_for_backward(coords)
        return out if channels_last else ops.nhwc_to_nchw(out)

    @staticmethod
    def backward(ctx, dout):
        (coords,) = ctx.saved_tensors
        st = ctx.state
        dout = dout.contiguous() if ctx.cl else ops.nchw_to_nhwc(dout)
        if ctx.is_flow != st.is_flow:
            B, _, H, W = coords.shape
            g = coords_grid(B, H, W, device=coords.device)
            coords = coords - g if st.is_flow else coords + g
        st.stash.append((coords, dout))
        return None, None, None, None, None, None


class AlternateCorrBlock:
    """Memory-efficient correlation (pytorch/core/corr.py:63-91 + alt_cuda_corr): the same numbers as CorrBlock without the
    N x N volume.  One fused launch per lookup (all levels, channels-last, scaled); the backward the reference compiled but
    never wired is live here: lookups only stash (coords, dOut), the feature gradients are formed once per step from
    chunks of the gradient volume (no O(N^2) buffer, no atomics).  alt_cuda_corr.forward / .backward themselves (one
    level per call, the extension's signature) are in flow_supervisor_amd/alt_cuda_corr.py."""

    @on_tensor_device
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        if not 1 <= num_levels <= 4:
            raise NotImplementedError("the HIP correlation kernels are built for 1..4 pyramid levels")
        if fmap1.shape[1] % 4 != 0:
            raise NotImplementedError("the HIP alt-corr kernels need a channel count that is a multiple of 4")
        self.num_levels = num_levels
        self.radius = radius
        fmap1 = fmap1.float()
        fmap2 = fmap2.float()
        self._state = _GradState(fmap1.device)
        self._tracks_grad = torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad)
        self._anchor = _AltBuildFn.apply(fmap1, fmap2, self._state, num_levels, radius) if self._tracks_grad else None
        with torch.no_grad():
            self.pyramid = [(fmap1, fmap2)]
            f1, f2 = fmap1, fmap2
            for _ in range(self.num_levels):
                f1 = F.avg_pool2d(f1, 2, stride=2)
                f2 = F.avg_pool2d(f2, 2, stride=2)
                self.pyramid.append((f1, f2))
            # channels-last copies once per pair instead of once per level per iteration (corr.py:82-83)
            self._f1 = ops.nchw_to_nhwc(fmap1.detach())
            self._f2 = [ops.nchw_to_nhwc(self.pyramid[i][1].detach()) for i in range(self.num_levels)]
            # ... and the same maps pre-split to records for the tile GEMM of the lookup (split arithmetic, as the volume build)
            self._recs = None
            self._regime = None
            C = fmap1.shape[1]
            if C % 32 == 0 and C <= 256 and fmap1.is_cuda:
                B = fmap1.shape[0]
                # (one amax word for every level of the target maps: the pooled levels are means of level 0)
                w2 = ops.amax_tensor(self._f2[0])
                self._recs = (ops.to_records(self._f1.view(B, -1, C)), [ops.to_records(f.view(B, -1, C), amax=w2) for f in self._f2])
                # every lookup picks its kernel by the spread of the flow it is handed (fsraft_altcorr_mfma_fwd's `regime`)
                self._regime = torch.zeros(8, dtype=torch.int32, device=fmap1.device) if ops.ALT_DISPATCH else None

    @on_tensor_device
    def __call__(self, coords, channels_last=False, is_flow=False, out=None):
        coords = coords.float()
        if out is not None and not channels_last:
            raise ValueError("out= is for channels_last lookups")
        if self._tracks_grad and torch.is_grad_enabled():
            if not self._state.stash:
                self._state.is_flow = is_flow
            if channels_last:
                return _AltLookupFn.apply(self._anchor, coords.detach(), self, True, is_flow, out)
            from .update import as_nchw          # (as CorrBlock.__call__)
            return as_nchw(_AltLookupFn.apply(self._anchor, coords.detach(), self, True, is_flow, None))
        res = ops.altcorr_fused_fwd(self._f1, self._f2, coords, self.radius, is_flow, recs=self._recs, out=out, regime=self._regime)
        if channels_last:
            return res
        from . import update as _u
        if _u.NCHW_VIEWS and _u.TWINS:
            return res.permute(0, 3, 1, 2)
        y = ops.nhwc_to_nchw(res)
        if _u.TWINS:
            y._fs_cl = (res, y._version)
        return y
