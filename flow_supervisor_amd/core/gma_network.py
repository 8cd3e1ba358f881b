"""RAFT-GMA forward (pytorch/core/gma_network.py:26-129) on the HIP hot path (benchmark config 5)."""
import os

import torch
import torch.nn as nn
from torch.amp import autocast

from . import streams
from .corr import CorrBlock
from .extractor import BasicEncoder
from .gma import Attention
from .raft import convex_upsample
from .update import GMAUpdateBlock, to_channels_last
from .utils.utils import coords_grid, upflow8
from .._lib import on_tensor_device


class RAFTGMA(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        args.corr_levels = 4
        args.corr_radius = 4
        if "dropout" not in self.args:
            self.args.dropout = 0
        if "mixed_precision" not in self.args:
            self.args.mixed_precision = False
        from .utils.utils import warn_mixed_precision
        warn_mixed_precision(self.args)
        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=args.dropout)
        self.cnet = BasicEncoder(output_dim=hdim + cdim, norm_fn="batch", dropout=args.dropout)
        self.cnet.out_channels_last = True   # the context features reach the update block channels_last (extractor._Encoder.forward)
        self.update_block = GMAUpdateBlock(self.args, hidden_dim=hdim)
        self.att = Attention(args=self.args, dim=cdim, heads=self.args.num_heads, max_pos_size=160, dim_head=cdim)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def initialize_flow(self, img):
        N, C, H, W = img.shape
        c = coords_grid(N, H // 8, W // 8, device=img.device)
        return c, c.clone()

    @on_tensor_device
    def upsample_flow(self, flow, mask):
        return convex_upsample(flow, mask)

    # -- pieces shared with GMAL2L ------------------------------------------------------
    def _features(self, a, b):
        with autocast("cuda", enabled=False):        # (args.mixed_precision: utils.warn_mixed_precision)
            f1, f2 = self.fnet([a, b])
        return f1.float(), f2.float()

    def _context(self, img):
        """-> (net, inp) channels-last and the attention map of inp."""
        with autocast("cuda", enabled=False):        # (args.mixed_precision: utils.warn_mixed_precision)
            cnet = self.cnet(img)
        net, inp = torch.split(cnet.float(), [self.hidden_dim, self.context_dim], dim=1)
        net = to_channels_last(torch.tanh(net))
        inp = to_channels_last(torch.relu(inp))
        return net, inp, self.att.forward_cl(inp, records=True)     # (consumed by update_block.forward_cl only)

    @on_tensor_device
    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False):
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        # context encoder + attention on the second stream beside the feature encoder and the volume build (core/streams.py)
        overlap = streams.OVERLAP and image1.is_cuda
        if overlap:
            with torch.cuda.stream(streams.fork(image1.device)):
                net, inp, attention = self._context(image1)
        fmap1, fmap2 = self._features(image1, image2)
        corr_fn = CorrBlock(fmap1, fmap2, radius=self.args.corr_radius)
        if overlap:
            streams.join(image1.device, net, inp, attention)
        else:
            net, inp, attention = self._context(image1)

        # as in RAFT.forward the loop carries the flow (the lookups add the pixel grid themselves), and in training the mask head
        # and the upsampler of all iterations run as one launch each after the loop (update.HeadBatch)
        B, _, Hi, Wi = image1.shape
        flow = flow_init.float() if flow_init is not None else torch.zeros(B, 2, Hi // 8, Wi // 8, device=image1.device)

        flow_predictions = []
        flow_up = None
        hb = self.update_block.head_batch(iters, net) if not test_mode else None
        mb = self.update_block.motion_batch(iters, net) if not test_mode else None        # (update.MotionBatch)
        flows = []
        for itr in range(iters):
            flow = flow.detach()
            corr = corr_fn(flow, channels_last=True, is_flow=True, **({"out": mb.corr[mb.n]} if mb is not None else {}))
            want_up = not test_mode or itr == iters - 1          # test_mode keeps only the last flow_up (gma_network.py:127-128)
            net, up_mask, delta_flow = self.update_block.forward_cl(net, inp, corr, flow, attention, need_mask=want_up, head_batch=hb,
                                                                    motion_batch=mb)
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
