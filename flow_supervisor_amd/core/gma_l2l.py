"""Flow-supervisor forward for the GMA variant (pytorch/core/gma_l2l.py:26-129).

Same two-phase schedule as core/l2l.py.  As in the reference, the second phase keeps calling ``update_block``
(gma_l2l.py:112); ``grad_update_block`` exists (and is part of the state_dict) but is not used by forward.
"""
import torch
import torch.nn.functional as F

from . import streams
from .corr import CorrBlock
from .extractor import _prepare_packs
from .gma import mark_records
from .gma_network import RAFTGMA
from .l2l import _crop_back, _offsets, _pad_state
from .raft import convex_upsample
from .update import GMAUpdateBlock
from .utils.utils import upflow8
from .._lib import on_tensor_device


class GMAL2L(RAFTGMA):
    def __init__(self, args):
        super().__init__(args)
        self.grad_update_block = GMAUpdateBlock(self.args, hidden_dim=self.hidden_dim)

    @on_tensor_device
    def forward(self, image1, image2, ci1=None, ci2=None, ox=None, oy=None, iters=12, flow_init=None,
                upsample=True, test_mode=False, supervisor_grad=True, sup_grad_samples=None):
        norm = lambda im: (2 * (im / 255.0) - 1.0).contiguous()
        image1, image2 = norm(image1), norm(image2)
        if ci1 is not None:
            ci1, ci2 = norm(ci1), norm(ci2)
        if not test_mode and ci1 is None:
            raise NameError("GMAL2L.forward in training mode needs the uncropped pair ci1/ci2 and offsets ox/oy")

        def uncropped(B_):
            """Second feature pair, context and attention of the uncropped frames (gma_l2l.py:91-99)."""
            k = sup_grad_samples
            k = k if (k is not None and 0 < k < B_ and torch.is_grad_enabled()) else None
            if k is not None:
                # (extension, see core/l2l.py) only the first k samples' supervisor predictions receive gradient: the
                # uncropped frames of the others are encoded without a graph
                ta1, ta2 = self._features(ci1[:k], ci2[:k])
                with torch.no_grad():
                    tb1, tb2 = self._features(ci1[k:], ci2[k:])
                t1, t2 = torch.cat([ta1, tb1]), torch.cat([ta2, tb2])
            else:
                t1, t2 = self._features(ci1, ci2)
            with torch.no_grad():             # (detached by the reference: no graph, no saved activations)
                _, inp2, att2 = self._context(ci1)
            return t1, t2, inp2, att2, k

        early = None
        if streams.OVERLAP and not test_mode and ci1 is not None and image1.is_cuda and iters // 2 < iters:
            # second stream (core/streams.py, as in core/l2l.py): the student's context + attention, then the uncropped frames
            _prepare_packs(self.fnet)
            _prepare_packs(self.cnet)
            with torch.cuda.stream(streams.fork(image1.device)):
                net, inp, attention = self._context(image1)
                ctx_done = streams.mark(image1.device)
                with torch.set_grad_enabled(torch.is_grad_enabled() and supervisor_grad):
                    early = uncropped(image1.shape[0])
        fmap1, fmap2 = self._features(image1, image2)
        corr_fn = CorrBlock(fmap1, fmap2, radius=self.args.corr_radius)
        if early is not None:
            streams.join(image1.device, net, inp, attention, event=ctx_done)
        else:
            net, inp, attention = self._context(image1)
        # the loop carries the flow (see core/l2l.py): lookups add the pixel grid themselves
        B, _, Hi, Wi = image1.shape
        flow = flow_init.float() if flow_init is not None else torch.zeros(B, 2, Hi // 8, Wi // 8, device=image1.device)

        flow_predictions = []
        flow_up = None
        half = iters // 2
        crop = None
        grad_mode = torch.is_grad_enabled()
        # training: mask head + upsampler of a phase's iterations as one launch each after the loop (update.HeadBatch)
        hb = self.update_block.head_batch(half, net) if not test_mode else None
        mb = self.update_block.motion_batch(half, net) if not test_mode else None      # (update.MotionBatch; the student's phase)
        hb2, flows, flows2 = None, [], []
        try:
            for itr in range(iters):
                if itr == half and not supervisor_grad and not test_mode:
                    # (extension, default off) the caller's loss does not reach the supervisor's predictions -- sequence_loss_unsup
                    # only reads the last one, detached (train.py:110-111) -- so the second half records no graph
                    torch.set_grad_enabled(False)
                flow = flow.detach()
                if test_mode or itr != half:          # (at the switch the reference looks up the crop's volume and drops it)
                    slot = {"out": mb.corr[mb.n]} if (mb is not None and itr < half and not test_mode) else {}
                    corr = corr_fn(flow, channels_last=True, is_flow=True, **slot)
                if not (test_mode or itr < half) and itr == half:
                    if ci1 is not None:
                        crop = (_offsets(ox, net.shape[0]), _offsets(oy, net.shape[0]), tuple(image1.shape[-2:]))
                        net, flow = _pad_state(net, flow, crop[0], crop[1], crop[2], tuple(ci1.shape[-2:]))
                        if early is not None:
                            streams.join(net.device, *early[:4])
                            tfmap1, tfmap2, inp, attention, k = early
                            early = None
                        else:
                            tfmap1, tfmap2, inp, attention, k = uncropped(net.shape[0])
                        corr_fn = CorrBlock(tfmap1, tfmap2, radius=self.args.corr_radius, grad_samples=k)
                        corr = corr_fn(flow, channels_last=True, is_flow=True)
                    net, corr, inp, flow = net.detach(), corr.detach(), inp.detach(), flow.detach()
                    attention = mark_records(attention.detach(), like=attention)
                    hb2 = self.update_block.head_batch(iters - half, net)
                want_up = not test_mode or itr == iters - 1          # test_mode keeps only the last flow_up (gma_l2l.py:126-127)
                cur = None if test_mode else (hb if itr < half else hb2)
                net, up_mask, delta_flow = self.update_block.forward_cl(net, inp, corr, flow, attention, need_mask=want_up, head_batch=cur,
                                                                        grad_samples=sup_grad_samples if itr >= half else None,
                                                                        motion_batch=mb if (itr < half and not test_mode) else None)

                flow = flow + delta_flow
                if cur is not None:
                    (flows if itr < half else flows2).append(flow)
                    continue
                if not want_up:
                    continue
                if up_mask is None:
                    flow_up = upflow8(flow)
                else:
                    flow_up = convex_upsample(flow, up_mask, channels_last=True)
                if not test_mode and itr >= half:
                    flow_up = _crop_back(flow_up, *crop)
                flow_predictions.append(flow_up)
        finally:
            torch.set_grad_enabled(grad_mode)
        # the deferred mask head + upsampler of each phase (after the caller's gradient mode is back: the unlabelled pass of the
        # flow-supervisor step switches it off for the supervisor's half).  A phase that ran without a batch has its predictions in
        # the list already; the student's come first.
        if hb is not None and flows:
            flow_predictions = hb.finish(flows) + flow_predictions
        if hb2 is not None and flows2:
            flow_predictions = flow_predictions + [_crop_back(p, *crop) for p in hb2.finish(flows2)]

        if test_mode:
            return flow, flow_up
        return flow_predictions
