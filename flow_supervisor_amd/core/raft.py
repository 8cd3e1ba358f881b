"""RAFT model shell around the HIP hot path.  Same constructor, ``forward`` signature, return
values and state_dict keys as pytorch/core/raft.py:24-144; the loop body differs only in that
tensors between the lookup, the update block and the upsampler stay channels-last so no
layout conversion (and no torch.cat / softmax / unfold) runs per iteration.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from . import streams
from .corr import AlternateCorrBlock, CorrBlock
from .extractor import BasicEncoder, SmallEncoder
from .update import BasicUpdateBlock, SmallUpdateBlock, to_channels_last
from .utils.utils import coords_grid, upflow8
from .._lib import on_tensor_device

autocast = torch.autocast


class _ConvexUpsample(torch.autograd.Function):
    """upsample_flow on the HIP kernel; mask is channels-last [N,H,W,576]."""

    @staticmethod
    def forward(ctx, flow, mask_cl):
        ctx.save_for_backward(flow, mask_cl)
        return ops.upsample_fwd(flow, mask_cl)

    @staticmethod
    def backward(ctx, g):
        flow, mask_cl = ctx.saved_tensors
        dflow, dmask = ops.upsample_bwd(flow, mask_cl, g)
        return dflow, dmask


def convex_upsample(flow, mask, channels_last=False):
    """[N,2,H,W] x mask -> [N,2,8H,8W] (raft.py:72-83).  mask is [N,576,H,W] unless channels_last."""
    flow = flow.float()
    if not channels_last:
        mask = to_channels_last(mask.float())
    if flow.stride(2) != flow.shape[3] * flow.stride(3):
        flow = flow.contiguous()
    return _ConvexUpsample.apply(flow, mask)


class RAFT(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        if args.small:
            self.hidden_dim, self.context_dim = 96, 64
            args.corr_levels, args.corr_radius = 4, 3
        else:
            self.hidden_dim, self.context_dim = 128, 128
            args.corr_levels, args.corr_radius = 4, 4
        if "dropout" not in self.args:
            self.args.dropout = 0
        if "alternate_corr" not in self.args:
            self.args.alternate_corr = False
        if "mixed_precision" not in self.args:
            self.args.mixed_precision = False
        from .utils.utils import warn_mixed_precision
        warn_mixed_precision(self.args)
        hdim, cdim = self.hidden_dim, self.context_dim
        if args.small:
            self.fnet = SmallEncoder(output_dim=128, norm_fn="instance", dropout=args.dropout)
            self.cnet = SmallEncoder(output_dim=hdim + cdim, norm_fn="none", dropout=args.dropout)
            self.update_block = SmallUpdateBlock(self.args, hidden_dim=hdim)
        else:
            self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=args.dropout)
            self.cnet = BasicEncoder(output_dim=hdim + cdim, norm_fn="batch", dropout=args.dropout)
            self.cnet.out_channels_last = True   # the context features reach the update block channels_last (extractor._Encoder.forward)
            self.update_block = BasicUpdateBlock(self.args, hidden_dim=hdim)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def initialize_flow(self, img):
        """coords0 == coords1 == pixel grid at 1/8 resolution; flow = coords1 - coords0 (raft.py:63-70)."""
        N, C, H, W = img.shape
        c = coords_grid(N, H // 8, W // 8, device=img.device)
        return c, c.clone()

    @on_tensor_device
    def upsample_flow(self, flow, mask):
        """[N,2,H,W], [N,576,H,W] -> [N,2,8H,8W] convex combination (raft.py:72-83)."""
        return convex_upsample(flow, mask)

    @on_tensor_device
    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False):
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        hdim, cdim = self.hidden_dim, self.context_dim
        amp = False          # args.mixed_precision: no autocast here, see utils.warn_mixed_precision (fp32 storage everywhere, at full speed)

        def context():
            with autocast("cuda", enabled=amp):
                cnet = self.cnet(image1)
            net, inp = torch.split(cnet.float(), [hdim, cdim], dim=1)
            # hidden state stays channels-last in the loop
            return to_channels_last(torch.tanh(net)), to_channels_last(torch.relu(inp))

        # the context encoder reads image1 only: on the second stream beside the feature encoder and the volume build (core/streams.py)
        overlap = streams.OVERLAP and image1.is_cuda
        if overlap:
            with torch.cuda.stream(streams.fork(image1.device)):
                net, inp = context()
        with autocast("cuda", enabled=amp):
            fmap1, fmap2 = self.fnet([image1, image2])
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        if self.args.alternate_corr:
            corr_fn = AlternateCorrBlock(fmap1, fmap2, radius=self.args.corr_radius)
        else:
            corr_fn = CorrBlock(fmap1, fmap2, radius=self.args.corr_radius)
        if overlap:
            streams.join(image1.device, net, inp)
        else:
            net, inp = context()

        # The loop carries the FLOW, not coords1 = coords0 + flow (raft.py:121-131): the lookup adds the pixel grid itself,
        # the update block and the upsampler want the flow anyway, so an iteration has one framework op (flow + delta)
        # instead of three.  Same gradient structure: the flow entering an iteration is detached, delta_flow reaches the
        # loss through the upsampled prediction.
        B, _, Hi, Wi = image1.shape
        if flow_init is not None:
            flow = flow_init.float()
        else:
            flow = torch.zeros(B, 2, Hi // 8, Wi // 8, device=image1.device)

        flow_predictions = []
        flow_up = None
        # training: the mask head and the upsampler of all iterations run as one launch each after the loop (update.HeadBatch)
        hb = self.update_block.head_batch(iters, net) if not test_mode else None
        # ... and the motion encoder's backward of all iterations likewise (update.MotionBatch: the lookups write into its slots)
        mb = self.update_block.motion_batch(iters, net) if not test_mode else None
        flows = []
        for itr in range(iters):
            flow = flow.detach()
            corr = corr_fn(flow, channels_last=True, is_flow=True, **({"out": mb.corr[mb.n]} if mb is not None else {}))
            # test_mode returns only the last upsampled flow (raft.py:141-142): the mask convolution and the upsampler of
            # the other iterations are skipped -- same outputs, the reference computes them and drops them (raft.py:134-139)
            want_up = not test_mode or itr == iters - 1
            net, up_mask, delta_flow = self.update_block.forward_cl(net, inp, corr, flow, need_mask=want_up, head_batch=hb, motion_batch=mb)
            flow = flow + delta_flow
            if hb is not None:
                flows.append(flow)
                continue
            if not want_up:
                continue
            if up_mask is None:
                flow_up = upflow8(flow)
            else:
                flow_up = convex_upsample(flow, up_mask, channels_last=True)
            flow_predictions.append(flow_up)

        if hb is not None:
            flow_predictions = hb.finish(flows)
        if test_mode:
            return flow, flow_up
        return flow_predictions
