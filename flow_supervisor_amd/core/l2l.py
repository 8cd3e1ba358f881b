"""Flow-supervisor two-phase forward (pytorch/core/l2l.py:24-133) on the HIP hot path.

First half of the iterations: the student `update_block` refines flow on the augmented crop.  At the half-way
point the hidden state and flow are zero-padded out to the uncropped frame, a second correlation volume is built
from the uncropped pair, everything is detached, and `grad_update_block` (the supervisor) carries on; its
predictions are cropped back to the student's window.  Both halves run the same kernels as RAFT.forward; the
padding happens directly on the channels-last hidden state.
"""
import torch
import torch.nn.functional as F
from torch.amp import autocast

from . import streams
from .corr import AlternateCorrBlock, CorrBlock
from .extractor import _prepare_packs
from .raft import RAFT, convex_upsample
from .update import BasicUpdateBlock, to_channels_last
from .utils.utils import upflow8
from .._lib import on_tensor_device


def _offsets(v, B):
    """Crop offsets per sample.  The reference reads `ox[0]` / `oy[0]` for the whole batch (l2l.py:87-88): a tensor or a
    python int means that.  A python list / tuple gives every sample its own offset (extension: lets the labelled and the
    unlabelled sample of the flow-supervisor step share one batched forward, train.SemiTrainStep)."""
    if isinstance(v, (list, tuple)):
        if len(v) != B:
            raise ValueError(f"{len(v)} crop offsets for a batch of {B}")
        return [int(x) for x in v]
    return [int(v) if isinstance(v, int) else int(v[0])] * B


def _pad_state(net, flow, ox_l, oy_l, orig, targ):
    """Zero-pad the hidden state (channels-last) and the flow from the crop's grid into the uncropped frame's grid
    (l2l.py:90-93), sample by sample when the offsets differ."""
    (orig_h, orig_w), (targ_h, targ_w) = orig, targ

    def pads(ox_, oy_):
        return ox_ // 8, (targ_w - ox_ - orig_w) // 8, oy_ // 8, (targ_h - oy_ - orig_h) // 8
    if len(set(zip(ox_l, oy_l))) == 1:
        l, r, t, b = pads(ox_l[0], oy_l[0])
        return F.pad(net, (0, 0, l, r, t, b)), F.pad(flow, (l, r, t, b))
    nets, flows = [], []
    for i, (ox_, oy_) in enumerate(zip(ox_l, oy_l)):
        l, r, t, b = pads(ox_, oy_)
        nets.append(F.pad(net[i:i + 1], (0, 0, l, r, t, b)))
        flows.append(F.pad(flow[i:i + 1], (l, r, t, b)))
    return torch.cat(nets, 0), torch.cat(flows, 0)


class _CropBackFn(torch.autograd.Function):
    """Per-sample windows of the full-frame prediction as ONE autograd node: a strided copy per sample forward, one zero
    fill + a copy per sample backward (slicing + cat costs five framework launches per prediction in backward: two
    zero-filled full frames, two copies, one add)."""

    @staticmethod
    def forward(ctx, flow_up, ox_l, oy_l, orig):
        h, w = orig
        out = torch.empty(flow_up.shape[0], flow_up.shape[1], h, w, device=flow_up.device, dtype=flow_up.dtype)
        for i, (ox_, oy_) in enumerate(zip(ox_l, oy_l)):
            out[i].copy_(flow_up[i, :, oy_: oy_ + h, ox_: ox_ + w])
        ctx.win = (tuple(ox_l), tuple(oy_l), h, w, tuple(flow_up.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        ox_l, oy_l, h, w, shape = ctx.win
        d = torch.zeros(shape, device=g.device, dtype=g.dtype)
        for i, (ox_, oy_) in enumerate(zip(ox_l, oy_l)):
            d[i, :, oy_: oy_ + h, ox_: ox_ + w].copy_(g[i])
        return d, None, None, None


def _crop_back(flow_up, ox_l, oy_l, orig):
    """The supervisor's full-frame prediction cut back to the student's window (l2l.py:124-125)."""
    orig_h, orig_w = orig
    if len(set(zip(ox_l, oy_l))) == 1:
        return flow_up[:, :, oy_l[0]: oy_l[0] + orig_h, ox_l[0]: ox_l[0] + orig_w]
    return _CropBackFn.apply(flow_up, ox_l, oy_l, orig)


class L2L(RAFT):
    def __init__(self, args):
        super().__init__(args)
        self.grad_update_block = BasicUpdateBlock(self.args, hidden_dim=self.hidden_dim)   # l2l.py:27

    @on_tensor_device
    def forward(self, image1, image2, ci1=None, ci2=None, ox=None, oy=None, iters=24, flow_init=None,
                upsample=True, test_mode=False, supervisor_grad=True, sup_grad_samples=None):
        norm = lambda im: (2 * (im / 255.0) - 1.0).contiguous()
        image1, image2 = norm(image1), norm(image2)
        if ci1 is not None:
            ci1, ci2 = norm(ci1), norm(ci2)
        hdim, cdim = self.hidden_dim, self.context_dim
        amp = False          # args.mixed_precision: no autocast here, see utils.warn_mixed_precision (fp32 storage everywhere, at full speed)
        if not test_mode and ci1 is None:
            # the reference reads oy_/ox_ in the second half without having set them (l2l.py:124-125)
            raise NameError("L2L.forward in training mode needs the uncropped pair ci1/ci2 and offsets ox/oy")

        def features(a, b):
            with autocast("cuda", enabled=amp):
                f1, f2 = self.fnet([a, b])
            return f1.float(), f2.float()

        def context(a):
            with autocast("cuda", enabled=amp):
                c = self.cnet(a)
            return torch.split(c.float(), [hdim, cdim], dim=1)

        def uncropped(B_):
            """Second feature pair + context of the uncropped frames (l2l.py:95-101): (tfmap1, tfmap2, inp, k)."""
            k = sup_grad_samples
            if k is not None and 0 < k < B_ and torch.is_grad_enabled():
                # (extension) the caller's loss reaches the supervisor's predictions of the first k samples only
                # (the flow-supervisor step batches its labelled and its unlabelled sample: train.SemiTrainStep): the
                # uncropped frames of the others are encoded without a graph -- their gradient would be zeros
                # pushed through the whole feature encoder
                ta1, ta2 = features(ci1[:k], ci2[:k])
                with torch.no_grad():
                    tb1, tb2 = features(ci1[k:], ci2[k:])
                t1, t2 = torch.cat([ta1, tb1]), torch.cat([ta2, tb2])
            else:
                t1, t2 = features(ci1, ci2)
            with torch.no_grad():         # (detached by the reference, l2l.py:104: no graph, no saved activations)
                _, inp2 = context(ci1)
                inp2 = to_channels_last(torch.relu(inp2))
            return t1, t2, inp2, (k if (k is not None and 0 < k < B_) else None)

        early = None
        if streams.OVERLAP and not test_mode and ci1 is not None and image1.is_cuda and iters // 2 < iters:
            # the uncropped frames' encodings depend on the inputs only: issued now on the second stream, joined at the switch
            # iteration where the reference computes them (core/streams.py; 56.0 -> 58.8 pairs/s at one pair per GPU)
            _prepare_packs(self.fnet)                   # (the packed weights both streams read are built on this one first)
            _prepare_packs(self.cnet)
            with torch.cuda.stream(streams.fork(image1.device)):
                # the student's own context first (the loop waits for it, and only for it: the event), then the uncropped frames
                net, inp = context(image1)
                net, inp = to_channels_last(torch.tanh(net)), to_channels_last(torch.relu(inp))
                ctx_done = streams.mark(image1.device)
                with torch.set_grad_enabled(torch.is_grad_enabled() and supervisor_grad):
                    early = uncropped(image1.shape[0])
        fmap1, fmap2 = features(image1, image2)
        if self.args.alternate_corr:
            corr_fn = AlternateCorrBlock(fmap1, fmap2, radius=self.args.corr_radius)
        else:
            corr_fn = CorrBlock(fmap1, fmap2, radius=self.args.corr_radius)
        if early is not None:
            streams.join(image1.device, net, inp, event=ctx_done)
        else:
            net, inp = context(image1)
            net = to_channels_last(torch.tanh(net))
            inp = to_channels_last(torch.relu(inp))

        # As in RAFT.forward the loop carries the FLOW, not coords1 = coords0 + flow (l2l.py:66-70, 110-122): the lookups add
        # the pixel grid themselves, so an iteration has one framework op (flow + delta) instead of three; same gradient
        # structure -- the flow entering an iteration is detached.
        B, _, Hi, Wi = image1.shape
        flow = flow_init.float() if flow_init is not None else torch.zeros(B, 2, Hi // 8, Wi // 8, device=image1.device)

        flow_predictions = []
        flow_up = None
        half = iters // 2
        crop = None
        grad_mode = torch.is_grad_enabled()
        # training: mask head + upsampler of a phase's iterations as one launch each after the loop (update.HeadBatch), one batch
        # per phase (the two phases run different blocks on different grids)
        hb = self.update_block.head_batch(half, net) if not test_mode else None
        mb = self.update_block.motion_batch(half, net) if not test_mode else None      # (update.MotionBatch; the student's phase)
        hb2, mb2, flows, flows2 = None, None, [], []
        try:
            for itr in range(iters):
                if itr == half and not supervisor_grad and not test_mode:
                    # (extension, default off) the caller's loss does not reach the supervisor's predictions -- sequence_loss_unsup
                    # only reads the last one, detached (train.py:110-111) -- so the second half records no graph
                    torch.set_grad_enabled(False)
                flow = flow.detach()
                if test_mode or itr != half:          # (at the switch the reference looks up the crop's volume and drops it, l2l.py:73/102)
                    cur_mb = None if test_mode else (mb if itr < half else mb2)
                    slot = {"out": cur_mb.corr[cur_mb.n]} if cur_mb is not None else {}
                    corr = corr_fn(flow, channels_last=True, is_flow=True, **slot)
                want_up = not test_mode or itr == iters - 1          # test_mode keeps only the last flow_up (l2l.py:130-131)
                if test_mode or itr < half:
                    net, up_mask, delta_flow = self.update_block.forward_cl(net, inp, corr, flow, need_mask=want_up, head_batch=hb,
                                                                            motion_batch=None if test_mode else mb)
                else:
                    if itr == half:
                        if ci1 is not None:
                            crop = (_offsets(ox, net.shape[0]), _offsets(oy, net.shape[0]), tuple(image1.shape[-2:]))
                            net, flow = _pad_state(net, flow, crop[0], crop[1], crop[2], tuple(ci1.shape[-2:]))   # (l2l.py:90-93)
                            if early is not None:
                                streams.join(net.device, *early[:3])
                                tfmap1, tfmap2, inp, k = early
                                early = None
                            else:
                                tfmap1, tfmap2, inp, k = uncropped(net.shape[0])
                            corr_fn = CorrBlock(tfmap1, tfmap2, radius=self.args.corr_radius, grad_samples=k)   # second volume (l2l.py:101)
                            corr = corr_fn(flow, channels_last=True, is_flow=True)
                        net, corr, inp, flow = net.detach(), corr.detach(), inp.detach(), flow.detach()
                        hb2 = self.grad_update_block.head_batch(iters - half, net)
                        # (the switch iteration's own correlation features are detached: it runs outside the batch)
                        mb2 = self.grad_update_block.motion_batch(iters - half - 1, net, grad_samples=sup_grad_samples)
                    net, up_mask, delta_flow = self.grad_update_block.forward_cl(net, inp, corr, flow, head_batch=hb2, grad_samples=sup_grad_samples,
                                                                                 motion_batch=mb2 if itr > half else None)

                flow = flow + delta_flow
                if not test_mode and (hb if itr < half else hb2) is not None:
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
