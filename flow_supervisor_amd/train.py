"""The training step the benchmark times: forward (12 GRU iterations), sequence loss, backward,
gradient all-reduce, clip, AdamW.  Callers of the hot path, written with framework ops."""
import torch

import os

from .core import streams
from .parallel import FlatAdamW, FlatGradients


MAX_FLOW = 400          # pytorch/train.py:55
SUP_GRAD_SAMPLES = True  # batched flow-supervisor step: no graph / backward for the unlabelled samples' supervisor half (known-zero gradients)


class _SeqLossFn(torch.autograd.Function):
    """All predictions in one fsraft_sequence_loss launch: loss, its gradients and the EPE statistics."""

    @staticmethod
    def forward(ctx, weights, gt, valid, max_flow, eps, metric_idx, *preds):
        import ctypes
        from . import _lib as L
        preds = [p.contiguous() for p in preds]
        L.require_cuda_f32(*preds)
        B, _, H, W = preds[0].shape
        n = len(preds)
        need = [ctx.needs_input_grad[6 + i] for i in range(n)]
        if all(need) and all(p.shape == preds[0].shape for p in preds):
            # one buffer, the gradients back to back: a producer that made the predictions in one launch (update.HeadBatch) takes
            # them back as one tensor
            dps = list(torch.empty((n,) + tuple(preds[0].shape), device=preds[0].device, dtype=torch.float32).unbind(0))
        else:
            dps = [torch.empty_like(p) if nd else None for p, nd in zip(preds, need)]
        out = torch.zeros(6, device=preds[0].device, dtype=torch.float32)
        a_p = (ctypes.c_void_p * n)(*[p.data_ptr() for p in preds])
        a_d = (ctypes.c_void_p * n)(*[d.data_ptr() if d is not None else None for d in dps])
        a_w = (ctypes.c_float * n)(*[float(w) for w in weights])
        gt = gt.contiguous().float() if gt is not None else None
        valid = valid.contiguous().float() if valid is not None else None
        L.check(L.load().fsraft_sequence_loss(ctypes.cast(a_p, L._PP), ctypes.cast(a_d, L._PP), a_w, n,
                                              -1 if metric_idx is None else int(metric_idx), L.ptr(gt), L.ptr(valid),
                                              float(max_flow), float(eps), B, H, W, L.ptr(out), L.stream()), "sequence_loss")
        ctx.dps = dps
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g, _gstats):
        dps, ctx.dps = ctx.dps, None
        live = [d for d in dps if d is not None]
        torch._foreach_mul_(live, g)              # one multi-tensor kernel instead of one multiply per prediction
        return (None,) * 6 + tuple(dps)


class _SemiLossFn(torch.autograd.Function):
    """Both losses of the batched flow-supervisor step on the UNSLICED predictions [2*bs, 2, H, W]: samples [0, bs) against the
    ground truth (sequence_loss), samples [bs, 2*bs) against the last prediction of their own half (sequence_loss_unsup), as two
    fsraft_sequence_loss launches on pointer offsets that write straight into one gradient tensor per prediction.  Slicing
    the predictions instead costs autograd five framework launches per prediction and half in backward (two zero-filled
    batches, two copies, one add): ~240 launches of a ~2000-launch step."""

    @staticmethod
    def forward(ctx, bs, w_sup, w_unsup, gt, valid, max_flow, eps, *preds):
        import ctypes
        from . import _lib as L
        preds = [p.contiguous() for p in preds]
        L.require_cuda_f32(*preds)
        B2, _, H, W = preds[0].shape
        n = len(preds)
        if B2 != 2 * bs or len(w_sup) != n or len(w_unsup) != n:
            raise ValueError("batched flow-supervisor loss: predictions must hold bs labelled then bs unlabelled samples")
        # (one buffer, the gradients back to back: update.HeadBatch takes a phase's gradients back as one tensor)
        dps = list(torch.empty((n,) + tuple(preds[0].shape), device=preds[0].device, dtype=torch.float32).unbind(0))
        out = torch.zeros(12, device=preds[0].device, dtype=torch.float32)
        gt = gt.contiguous().float()
        valid = valid.contiguous().float() if valid is not None else None
        L.require_cuda_f32(gt)
        half = bs * 2 * H * W * 4                     # bytes from sample 0 to sample bs
        lib = L.load()
        for k, (w, ofs) in enumerate(((w_sup, 0), (w_unsup, half))):
            a_p = (ctypes.c_void_p * n)(*[p.data_ptr() + ofs for p in preds])
            a_d = (ctypes.c_void_p * n)(*[d.data_ptr() + ofs for d in dps])
            a_w = (ctypes.c_float * n)(*[float(x) for x in w])
            if k == 0:
                g, v, mf = L.ptr(gt), L.ptr(valid), float(max_flow)
            else:                                     # pseudo label: the last prediction of the unlabelled samples (train.py:110-111)
                g, v, mf = ctypes.c_void_p(preds[-1].data_ptr() + ofs), None, float("inf")
            L.check(lib.fsraft_sequence_loss(ctypes.cast(a_p, L._PP), ctypes.cast(a_d, L._PP), a_w, n, -1, g, v, mf, float(eps),
                                             bs, H, W, ctypes.c_void_p(out.data_ptr() + 24 * k), L.stream()), "sequence_loss")
        ctx.dps, ctx.bs = dps, bs
        return out[0].clone(), out[6].clone()

    @staticmethod
    def backward(ctx, g_sup, g_unsup):
        dps, ctx.dps = ctx.dps, None
        bs = ctx.bs
        if g_sup is not None:
            torch._foreach_mul_([d[:bs] for d in dps], g_sup)
        else:
            for d in dps:
                d[:bs].zero_()
        if g_unsup is not None:
            torch._foreach_mul_([d[bs:] for d in dps], g_unsup)
        else:
            for d in dps:
                d[bs:].zero_()
        return (None,) * 7 + tuple(dps)


def semi_sequence_losses(flow_preds, bs, flow_gt, valid, gamma=0.8, gamma2=1.0, unsup_weight=1.0, max_flow=MAX_FLOW, gamma_unsup=0.8):
    """(sequence_loss of samples [0, bs), sequence_loss_unsup of samples [bs, 2*bs)) of batched flow-supervisor predictions:
    same values and gradients as the two functions on the two slices (pytorch/train.py:60-129), without the slices.
    gamma_unsup: the unlabelled loss has its OWN decay -- the reference's step calls sequence_loss_unsup without a gamma
    (train.py:276: always the default 0.8) while sequence_loss gets args.gamma (0.85 in the GMA recipe)."""
    nm = len(flow_preds)
    n = nm // 2
    w_sup = [gamma ** (n - i - 1) for i in range(n)] + [gamma2 ** (n - i - 1) for i in range(nm - n)]
    w_unsup = [unsup_weight * gamma_unsup ** (n - i - 1) for i in range(n)] + [0.0] * (nm - n)
    return _SemiLossFn.apply(bs, w_sup, w_unsup, flow_gt, valid, max_flow, 1e-3, *flow_preds)


def weighted_sequence_loss(flow_preds, weights, flow_gt=None, valid=None, max_flow=MAX_FLOW, eps=1e-3, metric_idx=None):
    """sum_i weights[i] * mean(mask * sqrt((pred_i - gt)^2 + eps^2)) on the fused kernel.  Returns (loss, stats) with
    stats = [loss, epe_sum, n<1px, n<3px, n<5px, n_valid] of prediction `metric_idx` (device tensor, no sync)."""
    return _SeqLossFn.apply(list(weights), flow_gt, valid, max_flow, eps, metric_idx, *flow_preds)


def sequence_loss(flow_preds, flow_gt, valid, gamma=0.8, gamma2=1.0, max_flow=MAX_FLOW, metrics=True):
    """pytorch/train.py:60-96, same signature and return value (loss, metrics dict; metrics=False: (loss, None) without the
    host read-back of the statistics).  The first half of the predictions
    (the student's, in the flow-supervisor forward) is weighted gamma^(n-i-1), the second half (the supervisor's)
    gamma2^(n-i-1) with n = len(flow_preds) // 2; Charbonnier penalty with eps = 1e-3; pixels with valid < 0.5 or
    |gt| >= max_flow are excluded; metrics (epe, 1px, 3px, 5px) come from prediction n-1 over valid > 0.5."""
    nm = len(flow_preds)
    n = nm // 2
    weights = [gamma ** (n - i - 1) for i in range(n)] + [gamma2 ** (n - i - 1) for i in range(nm - n)]
    loss, stats = weighted_sequence_loss(flow_preds, weights, flow_gt, valid, max_flow, 1e-3, n - 1)
    if not metrics:
        return loss, None
    _, epe_sum, n1, n3, n5, nv = stats.tolist()
    nv = max(nv, 1.0)
    return loss, {"epe": epe_sum / nv, "1px": n1 / nv, "3px": n3 / nv, "5px": n5 / nv}


def sequence_loss_unsup(flow_preds, flow_gt, valid, gamma=0.8, unsup_weight=1.0, max_flow=MAX_FLOW, metrics=True):
    """pytorch/train.py:99-129, same signature and return value: the unlabelled half of the flow-supervisor step.  The pseudo
    label is the LAST prediction (the supervisor's), detached; only the first half of the predictions (the student's) is
    penalised, weight unsup_weight * gamma^(n-i-1), Charbonnier eps = 1e-3, no mask.  Metrics (epe, 1px, 3px, 5px) are
    those of prediction n-1 against flow_gt over valid > 0.5, as in the reference (metrics=False skips them and their
    host read-back: the benchmark's steps stay free of synchronisation)."""
    nm = len(flow_preds)
    n = nm // 2
    pseudo = flow_preds[-1].detach()
    weights = [unsup_weight * gamma ** (n - i - 1) for i in range(n)]
    loss, _ = weighted_sequence_loss(list(flow_preds[:n]), weights, pseudo, None, float("inf"), 1e-3, None)
    if not metrics:
        return loss, None
    with torch.no_grad():
        _, stats = weighted_sequence_loss([flow_preds[n - 1].detach()], [0.0], flow_gt, valid, max_flow, 1e-3, 0)
    _, epe_sum, n1, n3, n5, nv = stats.tolist()
    nv = max(nv, 1.0)
    return loss, {"epe": epe_sum / nv, "1px": n1 / nv, "3px": n3 / nv, "5px": n5 / nv}


def raft_sequence_loss(flow_preds, flow_gt=None, gamma=0.8):
    """The benchmark objective: sum_i gamma^(n-1-i) * mean(sqrt((pred_i - gt)^2 + 1e-6)), every pixel valid,
    gt defaulting to zero flow."""
    n = len(flow_preds)
    return weighted_sequence_loss(flow_preds, [gamma ** (n - i - 1) for i in range(n)], flow_gt, None, float("inf"), 1e-3)[0]


class TrainStep:
    def __init__(self, model, lr=4e-4, wdecay=1e-5, eps=1e-8, clip=1.0, iters=12, capturable=False):
        self.model = model
        self.iters = iters
        self.clip = clip
        named = list(model.named_parameters())
        self.grads = FlatGradients([p for _, p in named], [n for n, _ in named])
        fused = self.grads.flat.is_cuda
        # clip + AdamW as one kernel over flat buffers on the GPU (FSRAFT_FLAT_ADAMW=0: torch's multi-tensor AdamW)
        self.flat_opt = fused and os.environ.get("FSRAFT_FLAT_ADAMW", "1") != "0"
        if self.flat_opt:
            self.opt = FlatAdamW(self.grads, lr=lr, weight_decay=wdecay, eps=eps)
        else:
            self.opt = torch.optim.AdamW(self.grads.params, lr=lr, weight_decay=wdecay, eps=eps, fused=fused,
                                         capturable=bool(capturable and fused))

    def set_lr(self, lr):
        """Learning rate of the next step (what `scheduler.step()` does through param_groups; for hipGraph replays call this
        -- or opt.sync_lr() after the scheduler -- BEFORE the replay: the kernel reads a device scalar)."""
        if self.flat_opt:
            self.opt.set_lr(lr)
        else:
            for g in self.opt.param_groups:
                if isinstance(g["lr"], torch.Tensor):
                    g["lr"].fill_(float(lr))
                else:
                    g["lr"] = float(lr)

    def _update(self):
        if self.flat_opt:
            self.opt.step(clip=self.clip)
        else:
            self.grads.clip_norm_(self.clip)
            self.opt.step()

    # The step in three parts, for callers that replay it as TWO hipGraphs with the collective issued eagerly between them
    # (bench.py at N > 1: no RCCL call inside a capture): forward_backward() gathers the gradients into the flat buffer
    # without exchanging them, exchange() is the one all-reduce of the step, update() is clip + AdamW.  __call__ runs the
    # overlapped route (bucket all-reduces from the backward hooks).
    def forward_backward(self, *args, **kw):
        """Forward, loss, backward; gradients of this rank in grads.flat, NOT exchanged.  Same arguments as __call__."""
        out = self._forward_backward(*args, exchange=False, **kw)
        self.grads.finish()
        return out

    def exchange(self):
        """The all-reduce (sum over ranks of local_batch / global_batch weighted gradients) of the whole flat buffer."""
        self.grads.exchange_all()

    def update(self):
        """clip_grad_norm_ + AdamW on the (exchanged) flat gradient buffer."""
        self._update()

    def _forward_backward(self, image1, image2, flow_gt=None, global_batch=None, exchange=True, flow_init=None):
        if global_batch is None:
            self.grads.begin(exchange=exchange)
        else:
            self.grads.begin(image1.shape[0], global_batch, exchange=exchange)
        # flow_init: RAFT.forward's warm start (pytorch/core/raft.py:118-119), [B,2,H/8,W/8]
        preds = self.model(image1, image2, iters=self.iters, **({} if flow_init is None else {"flow_init": flow_init}))
        loss = raft_sequence_loss(preds, flow_gt)
        with streams.accumulate_grad_warning_off():
            loss.backward()                       # bucket hooks start each all-reduce as its gradients complete
        return loss.detach()

    def __call__(self, *args, **kw):
        """image1, image2[, flow_gt, global_batch]: this rank's shard.  global_batch: pairs over all ranks (default: equal shards)."""
        out = self._forward_backward(*args, **kw)
        self.grads.all_reduce_mean_()
        self._update()
        return out


class SemiTrainStep(TrainStep):
    """The flow-supervisor optimisation step (pytorch/train.py:246-284) around L2L / GMAL2L: a labelled sample through
    the two-phase forward (student on the crop, supervisor on the uncropped frame) with `sequence_loss`, backward; an
    unlabelled sample through the same forward with `sequence_loss_unsup` (the supervisor's last prediction is the
    student's pseudo label), backward; then ONE clip + AdamW step on the accumulated gradients.  Both backward passes feed
    the same gradient buckets (FlatGradients.begin(backward_passes=2)).

    sup / unsup: (image1, image2, ci1, ci2, ox, oy, flow, valid) -- the crop pair, the uncropped pair, the crop's offsets
    (python ints or CPU tensors keep the step free of device synchronisation; the reference passes CUDA tensors and
    syncs on them), ground-truth flow and validity of the crop."""

    def __init__(self, model, lr=5e-6, wdecay=0.0, eps=1e-8, clip=1.0, iters=12, gamma=0.8, unsup_lambda=1.0,
                 capturable=False, batched=None):
        super().__init__(model, lr=lr, wdecay=wdecay, eps=eps, clip=clip, iters=iters, capturable=capturable)
        self.gamma, self.unsup_lambda = gamma, unsup_lambda
        # batched: the labelled and the unlabelled sample go through ONE forward / backward as a batch of two (their crop
        # offsets differ: L2L.forward takes a list).  Every op of the model is per-sample (InstanceNorm; BatchNorm frozen), each
        # loss is evaluated on its own slice of the predictions, and the parameter gradient of the sum of the two losses is
        # the sum of the two passes' gradients -- the same step as the reference's two passes up to summation order, with
        # kernels that see twice the pixels (at the recipe's batch size of 1 a launch fills a fifth of the chip).
        self.batched = (os.environ.get("FSRAFT_SEMI_BATCHED", "1") != "0") if batched is None else bool(batched)
        self._cat = None
        import inspect
        self._sup_kw = "sup_grad_samples" in inspect.signature(model.forward).parameters and SUP_GRAD_SAMPLES

    def _batched_inputs(self, sup, unsup):
        """cat of the two samples.  The key holds every input's identity AND version counter, so a caller that refreshes
        preallocated input buffers in place (the pattern of hipGraph replays) is not served the first batch again; while a
        stream capture is running the concatenation is always performed, into persistent buffers, so that the copies are
        nodes of the graph and every replay re-reads the caller's buffers."""
        srcs = sup[:4] + unsup[:4]
        capturing = srcs[0].is_cuda and torch.cuda.is_current_stream_capturing()
        key = tuple((id(t), t._version) for t in srcs)
        if self._cat is not None and self._cat[0] == key and not capturing:
            return self._cat[1]
        shapes = tuple((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]) for a, b in zip(sup[:4], unsup[:4]))
        if self._cat is not None and tuple(tuple(t.shape) for t in self._cat[1]) == shapes and self._cat[1][0].device == srcs[0].device:
            bufs = self._cat[1]                   # same addresses as in an earlier capture
        else:
            bufs = tuple(torch.empty(sh, device=a.device, dtype=a.dtype) for sh, a in zip(shapes, sup[:4]))
        for buf, a, b in zip(bufs, sup[:4], unsup[:4]):
            buf[:a.shape[0]].copy_(a)
            buf[a.shape[0]:].copy_(b)
        # (while capturing the copies above were only recorded: the buffers hold nothing yet, so the key is not remembered and the
        # next eager call copies again instead of being served never-filled buffers -- ADVICE r4)
        self._cat = (None if capturing else key, bufs, srcs)
        return bufs

    def _bn_training(self):
        return any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training for m in self.model.modules())

    def _forward_backward(self, sup, unsup, global_batch=None, exchange=True):
        if not self.batched:
            return self._sequential(sup, unsup, global_batch, exchange)
        if self._bn_training():
            # one batch of two is the reference's two passes only while every op is per-sample: a BatchNorm in training mode
            # (the chairs stage: pytorch/train.py:203-204 freezes it for every other stage) would take its statistics over
            # both samples and update its running statistics once instead of twice
            return self._sequential(sup, unsup, global_batch, exchange)
        bs = sup[0].shape[0]
        if global_batch is None:
            self.grads.begin(exchange=exchange)
        else:
            self.grads.begin(bs, global_batch, exchange=exchange)
        from .core.l2l import _offsets
        im1, im2, ci1, ci2 = self._batched_inputs(sup, unsup)
        ox = _offsets(sup[4], bs) + _offsets(unsup[4], unsup[0].shape[0])
        oy = _offsets(sup[5], bs) + _offsets(unsup[5], unsup[0].shape[0])
        kw = {"sup_grad_samples": bs} if self._sup_kw else {}        # the unlabelled samples' supervisor predictions carry no gradient
        preds = self.model(im1, im2, ci1, ci2, ox, oy, iters=2 * self.iters, **kw)
        if preds[0].is_cuda and unsup[0].shape[0] == bs:
            loss, loss_u = semi_sequence_losses(preds, bs, sup[6], sup[7], self.gamma, unsup_weight=self.unsup_lambda)
        else:
            loss, _ = sequence_loss([p[:bs] for p in preds], sup[6], sup[7], self.gamma, metrics=False)
            loss_u, _ = sequence_loss_unsup([p[bs:] for p in preds], unsup[6], unsup[7], unsup_weight=self.unsup_lambda, metrics=False)
        with streams.accumulate_grad_warning_off():
            (loss + loss_u).backward()
        del preds
        return loss.detach(), loss_u.detach()

    def _sequential(self, sup, unsup, global_batch=None, exchange=True):
        """The reference's order: two forward / backward passes (pytorch/train.py:270-277)."""
        if global_batch is None:
            self.grads.begin(backward_passes=2, exchange=exchange)
        else:
            self.grads.begin(sup[0].shape[0], global_batch, backward_passes=2, exchange=exchange)
        im1, im2, ci1, ci2, ox, oy, flow, valid = sup
        preds = self.model(im1, im2, ci1, ci2, ox, oy, iters=2 * self.iters)
        loss, _ = sequence_loss(preds, flow, valid, self.gamma, metrics=False)
        with streams.accumulate_grad_warning_off():
            loss.backward()
        del preds
        im1, im2, ci1, ci2, ox, oy, flow, valid = unsup
        preds = self.model(im1, im2, ci1, ci2, ox, oy, iters=2 * self.iters, supervisor_grad=False)
        loss_u, _ = sequence_loss_unsup(preds, flow, valid, unsup_weight=self.unsup_lambda, metrics=False)
        with streams.accumulate_grad_warning_off():
            loss_u.backward()
        del preds
        return loss.detach(), loss_u.detach()
