"""Data parallelism for the RAFT step (row (e) of SURVEY.md section 8): one process per GPU,
image pairs sharded on the batch dimension, ONE exchange per optimizer step -- an all-reduce
(sum, then 1/world) of the fp32 gradients over RCCL/xGMI.  Replaces the reference's
single-process nn.DataParallel (pytorch/train.py:192: per step broadcast of all parameters,
scatter, gather of 12-24 full-resolution outputs to GPU 0, reduce of gradients to GPU 0) and
tf.distribute.MirroredStrategy (train.py:75-78).

Gradients live in one flat fp32 buffer (after the exchange every ``p.grad`` is a view into it), laid out in
three buckets -- update block, context encoder, feature encoder, the order in which backward finishes them --
and each bucket's all-reduce is issued asynchronously from a post-accumulate hook as soon as its last gradient
exists, so that the exchange of the update block's 4.7 M parameters runs under the encoders' backward
(SURVEY.md 5 / 8e).  21 MB over 7 xGMI links of ~153 GB/s is ~0.1-0.3 ms, so what the overlap buys is the
latency of three collectives, not bandwidth.  Gradient clipping and AdamW then run identically on every rank
on identical tensors, so no second collective is needed.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(device_type=None, backend=None, timeout_s=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank).
    backend: nccl (= RCCL on ROCm) for GPU tensors, gloo for CPU tests; backend="gloo" with GPU tensors (several ranks
    sharing one device, which RCCL refuses -- the 2-process GPU test) stages the exchange through host memory.
    timeout_s: bound on the rendezvous and on every collective (default: torch's, 10 min for nccl) -- a rank that never
    arrives then ends the job with an exception on the others instead of a hang (bench.py sets it)."""
    kw = {}
    if timeout_s is not None:
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if device_type is None:
            device_type = "cuda" if torch.cuda.is_available() else "cpu"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if device_type == "cuda" and backend == "gloo":
            torch.cuda.set_device(local)
            dist.init_process_group("gloo", rank=rank, world_size=world, **kw)
        elif device_type == "cuda":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), **kw)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_batch(global_batch, rank, world):
    """Contiguous split of `global_batch` pairs over ranks; sizes differ by at most one."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


class FlatGradients:
    """All gradients of `params` in one contiguous fp32 buffer, exchanged in BUCKETS that overlap the rest of backward.

    Backward of RAFT reaches the update block's parameters first (one packed-arena unpack after the last of the 12
    iterations), then the context encoder, then -- behind the volume backward -- the feature encoder.  Parameters are
    grouped into buckets by the top-level module they belong to, laid out bucket by bucket in the flat buffer, and a
    bucket's all-reduce (RCCL, asynchronous, on the collective's own stream) starts from a post-accumulate hook the moment
    its last gradient has arrived, while autograd continues with the encoders.  `finish()` -- called
    by all_reduce_mean_ -- launches whatever is left and waits.

    Gradients are not accumulated INTO the buffer by autograd: `begin()` sets every .grad to None, so autograd hands each
    parameter its gradient tensor as is, and one multi-tensor copy per bucket moves them in (instead of one zero fill plus
    ~150 separate accumulation kernels per step).  After finish(), p.grad are views into the buffer again."""

    def __init__(self, params, names=None):
        params = list(params)
        names = list(names) if names is not None else [str(i) for i in range(len(params))]
        keep = [(n, p) for n, p in zip(names, params) if p.requires_grad]
        order = {}
        for n, _ in keep:                                    # buckets in first-appearance order of the top-level module ...
            order.setdefault(n.split(".", 1)[0], len(order))
        # ... laid out so that the module whose gradients arrive first (update block) comes first when it exists
        rank = lambda k: (0 if "update" in k else 1 if k.startswith("cnet") else 2, order[k])
        keys = sorted(order, key=rank)
        self.buckets = [[p for n, p in keep if n.split(".", 1)[0] == k] for k in keys]
        self.params = [p for b in self.buckets for p in b]
        pad = lambda k: (k + 63) // 64 * 64              # every tensor starts on a 256-byte boundary (FlatAdamW points the
        n = sum(pad(p.numel()) for p in self.params)        # parameters at the same offsets of its own buffer; the pad stays zero)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.views, self.slices, self.offsets, o = {}, [], {}, 0
        for b in self.buckets:
            o0 = o
            for p in b:
                self.views[p] = self.flat[o:o + p.numel()].view_as(p)
                self.offsets[p] = o
                o += pad(p.numel())
            self.slices.append((o0, o))
        self._bucket_of = {p: i for i, b in enumerate(self.buckets) for p in b}
        self._pending = None
        self._handles = []
        self._weight = 1.0
        self._active = False
        self._deferred = False
        self._caller_stream = None
        self.missing = []                 # parameters that received no gradient in the last step (FlatAdamW leaves them alone)
        # FSRAFT_DP_BUCKETS=0: no exchange from the backward hooks -- ONE all-reduce of the whole flat buffer in finish()
        self.bucketed = os.environ.get("FSRAFT_DP_BUCKETS", "1") != "0"
        # FSRAFT_DP_FORCE_COLLECTIVE=1: issue the all-reduces at world size 1 too (a -m gpu test runs the RCCL stream
        # ordering of the hook-time launches on a one-GPU box this way)
        self.force_collective = os.environ.get("FSRAFT_DP_FORCE_COLLECTIVE", "0") == "1"
        for p in self.params:
            p.grad = self.views[p]
            p.register_post_accumulate_grad_hook(self._hook)

    # ---- per step -------------------------------------------------------------------------------------------------
    def begin(self, local_batch=None, global_batch=None, backward_passes=1, exchange=True):
        """Call before backward: arms the bucket hooks for this step.  backward_passes: how many backward() calls feed this
        optimizer step (the flow-supervisor step runs a labelled and an unlabelled forward/backward before
        optimizer.step(), pytorch/train.py:270-277); a bucket is exchanged once its gradients have arrived that many
        times, parameters that take part in fewer passes hold their bucket back until finish().
        exchange=False: the hooks (and finish()) only gather the gradients into the flat buffer; the caller runs the
        collective itself with exchange_all() -- the route of a step replayed as two hipGraphs with the all-reduce issued
        eagerly between them (TrainStep.forward_backward / exchange / update)."""
        for p in self.params:
            p.grad = None
        self.missing = []
        self._pending = [len(b) * int(backward_passes) for b in self.buckets]
        self._done = [False] * len(self.buckets)
        self._handles = []
        world = dist.get_world_size() if dist.is_initialized() else 1
        self._weight = (1.0 / world) if local_batch is None else float(local_batch) / float(global_batch)
        self._active = True
        self._deferred = not exchange
        self._caller_stream = torch.cuda.current_stream(self.flat.device) if self.flat.is_cuda else None

    def zero_(self):
        """(kept for callers of the round-1 interface) equivalent to begin()."""
        self.begin()

    def _hook(self, p):
        if not self._active:
            return
        i = self._bucket_of[p]
        if self._done[i]:
            # the bucket was already copied out (and reduced): this gradient would be added, unreduced, on one rank only
            raise RuntimeError("FlatGradients: a gradient arrived for a bucket that was already exchanged -- more backward "
                               "passes than begin(backward_passes=...) announced (gradient accumulation, a second loss, "
                               "retain_graph); call begin(backward_passes=k) or finish() between them")
        self._pending[i] -= 1
        if self._pending[i] == 0 and self.bucketed:
            self._launch(i)

    def _launch(self, i):
        """Move bucket i's gradients into the flat buffer (one multi-tensor copy) and start its exchange."""
        if self._done[i]:
            return
        self._done[i] = True
        if self.flat.is_cuda:
            # (a hook runs on its gradient's stream; the bucket's other gradients may have been produced on others: core/streams.py)
            from .core import streams
            streams.order_current_behind_all(self.flat.device, self._caller_stream)
        have = [p for p in self.buckets[i] if p.grad is not None and p.grad is not self.views[p]]
        missing = [p for p in self.buckets[i] if p.grad is None]
        if have:
            torch._foreach_copy_([self.views[p] for p in have], [p.grad for p in have])
        if missing:                                   # (parameters that took no part in this step, e.g. biases in front of InstanceNorm)
            torch._foreach_zero_([self.views[p] for p in missing])
            self.missing += missing
        for p in self.buckets[i]:
            p.grad = self.views[p]
        if self.bucketed and not self._deferred and self._exchanging():
            a, b = self.slices[i]
            self._exchange(self.flat[a:b])

    def _exchanging(self):
        return dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collective)

    def _exchange(self, seg):
        seg.mul_(self._weight)
        if _host_staged(seg):
            _all_reduce_sum_(seg)
        else:
            self._handles.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Launch the buckets whose hooks did not complete (unused parameters) and wait for every exchange."""
        if not self._active:
            return
        for i in range(len(self.buckets)):
            self._launch(i)
        if not self.bucketed and not self._deferred and self._exchanging():
            self._exchange(self.flat)
        for h in self._handles:
            h.wait()
        self._handles = []
        self._active = False

    def exchange_all(self):
        """The whole flat buffer in ONE blocking all-reduce (after a begin(exchange=False) step's finish()): what a step split
        into two hipGraphs issues eagerly between them.  The weight of begin() applies."""
        if self._active:
            self.finish()
        if self._exchanging():
            self._exchange(self.flat)
            for h in self._handles:
                h.wait()
            self._handles = []

    def all_reduce_mean_(self, local_batch=None, global_batch=None):
        """Gradient of the GLOBAL-batch mean loss from per-rank gradients of local-batch mean losses: every rank's
        gradient is weighted by local_batch / global_batch before the sum (equal shards: 1 / world, what DataParallel's
        reduce + batch-mean loss give).  With begin() called before backward the buckets are already in flight and this
        only waits; without it (plain use) it performs the whole exchange here."""
        if not self._active:                      # begin() was not called: gradients were accumulated into the views by autograd
            if dist.is_initialized() and dist.get_world_size() > 1:
                w = 1.0 / dist.get_world_size() if local_batch is None else float(local_batch) / float(global_batch)
                self.flat.mul_(w)
                _all_reduce_sum_(self.flat)
            return
        self.finish()

    def clip_norm_(self, max_norm):
        """clip_grad_norm_(params, max_norm) on the flat buffer (pytorch/train.py:280)."""
        total = self.flat.norm()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        self.flat.mul_(coef)
        return total


class FlatAdamW(torch.optim.Optimizer):
    """`clip_grad_norm_` + `torch.optim.AdamW.step()` (pytorch/train.py:137, 280-282) for the parameters of a FlatGradients, as ONE
    elementwise kernel over flat buffers (csrc/optim.hip): the parameters are moved into a flat fp32 buffer (`p.data` becomes a
    view of it: names, shapes and the model's state_dict are unchanged), the moments live in two more.  Same update rule as
    torch's fused AdamW, operation for operation; the step count and the learning rate are device scalars, so the step can
    be captured in a hipGraph.

    It IS a `torch.optim.Optimizer` (one param group), so the reference's `StepLR(optimizer, ...)` / `scheduler.step()`
    (pytorch/train.py:139, 283) attaches to it: `step()` copies `param_groups[0]["lr"]` to the device scalar when it
    changed (`sync_lr()` does the same for hipGraph replays, which do not re-run Python).  `state_dict()` /
    `load_state_dict()` carry the moments, the step count and the hyper-parameters (checkpoint / resume).  Parameters
    that received no gradient in a step are left alone with their moments, as torch's AdamW skips `p.grad is None`."""

    def __init__(self, grads, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        flat = grads.flat
        if not flat.is_cuda:
            raise RuntimeError("FlatAdamW runs on the GPU (csrc/optim.hip); use torch.optim.AdamW on the CPU")
        super().__init__(grads.params, dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps),
                                            weight_decay=float(weight_decay)))
        self.grads = grads
        self.params = grads.params
        self.p = torch.empty_like(flat)
        self.p.zero_()
        with torch.no_grad():
            for q in self.params:
                o, k = grads.offsets[q], q.numel()
                self.p[o:o + k].copy_(q.detach().reshape(-1))
                q.data = self.p[o:o + k].view_as(q)
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count = torch.zeros(1, device=flat.device, dtype=torch.float32)
        self.lr = torch.full((1,), float(lr), device=flat.device, dtype=torch.float32)
        self._lr_host = float(lr)
        self._state = torch.empty(4, device=flat.device, dtype=torch.float32)
        self._skip = None                 # uint8 per 64-element block: parameters without a gradient this step
        self._skip_key = None

    # -- hyper-parameters live in the (single) param group, like any torch optimizer
    @property
    def betas(self):
        return self.param_groups[0]["betas"]

    @property
    def eps(self):
        return self.param_groups[0]["eps"]

    @property
    def weight_decay(self):
        return self.param_groups[0]["weight_decay"]

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = float(lr)
        self.sync_lr()

    def sync_lr(self):
        """param_groups[0]['lr'] (what an lr_scheduler writes) -> the device scalar the kernel reads."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self.lr.fill_(lr)
            self._lr_host = lr

    def _check_bound(self):
        base = self.p.data_ptr()
        for q in self.params:
            if q.data_ptr() != base + 4 * self.grads.offsets[q]:
                raise RuntimeError("FlatAdamW: a parameter no longer lives in the optimizer's flat buffer (model.to() / .float() / "
                                   "load_state_dict(assign=True) or a re-created Parameter after the optimizer was built); "
                                   "the kernel would update memory the module does not read.  Rebuild TrainStep / FlatAdamW.")

    def _skip_table(self):
        missing = self.grads.missing
        if not missing:
            return None
        key = tuple(id(q) for q in missing)
        if key != self._skip_key:
            if self._skip is None:
                self._skip = torch.zeros((self.p.numel() + 63) // 64, device=self.p.device, dtype=torch.uint8)
            else:
                self._skip.zero_()
            for q in missing:
                o = self.grads.offsets[q]
                self._skip[o // 64:(o + q.numel() + 63) // 64] = 1
            self._skip_key = key
        return self._skip

    @torch.no_grad()
    def step(self, closure=None, clip=None):
        """One update from grads.flat; clip: max gradient norm (None: no clipping).  Returns the gradient norm (0-dim) or None
        (with a closure: its loss, like torch optimizers)."""
        import ctypes
        from . import _lib as L
        from . import ops
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._check_bound()
        self.sync_lr()
        g = self.grads.flat
        norm = g.norm() if clip is not None else None
        skip = self._skip_table()
        b1, b2 = self.betas
        L.check(L.load().fsraft_adamw_flat(L.ptr(self.p), L.ptr(g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), g.numel(),
                                           L.ptr(self.step_count), L.ptr(norm), ctypes.c_float(float(clip) if clip is not None else 0.0),
                                           L.ptr(self.lr), ctypes.c_float(b1), ctypes.c_float(b2),
                                           ctypes.c_float(self.eps), ctypes.c_float(self.weight_decay), L.ptr(self._state),
                                           L.ptr(skip), L.stream()),
                "adamw_flat")
        ops.parameters_updated(self.params)       # the packed-weight caches key on Parameter._version
        return loss if closure is not None else norm

    def zero_grad(self, set_to_none=True):
        """Gradients live in FlatGradients (begin() resets them before every step); kept for `optimizer.zero_grad()` callers."""
        for q in self.params:
            q.grad = None

    def state_dict(self):
        """Moments, step count and hyper-parameters.  `layout` (numel per parameter, in buffer order) guards load_state_dict
        against a different model."""
        g = self.param_groups[0]
        return {"step": float(self.step_count.item()), "exp_avg": self.exp_avg.detach().clone(),
                "exp_avg_sq": self.exp_avg_sq.detach().clone(), "layout": [int(q.numel()) for q in self.params],
                "param_groups": [{k: v for k, v in g.items() if k != "params"}]}

    def load_state_dict(self, sd):
        if list(sd["layout"]) != [int(q.numel()) for q in self.params]:
            raise ValueError("FlatAdamW.load_state_dict: the checkpoint's parameter layout differs from this model's")
        with torch.no_grad():
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.step_count.fill_(float(sd["step"]))
        for k, v in sd["param_groups"][0].items():
            self.param_groups[0][k] = v
        self.sync_lr()


def _host_staged(t):
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce_sum_(t):
    if _host_staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def broadcast_parameters(module, src=0):
    """One-time parameter sync at start-up (instead of DataParallel's per-step replicate)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        # Received into a staging tensor and copied INTO the parameter under no_grad: `copy_` bumps the tensor's version
        # counter, which the packed-weight caches of the update block / encoders key on (a c10d broadcast into the
        # parameter, or any write through `.data`, leaves the counter alone and those caches stale).
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                h = t.cpu() if _host_staged(t) else t.detach().clone()
                dist.broadcast(h, src)
                t.copy_(h)


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        if _host_staged(t):
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
