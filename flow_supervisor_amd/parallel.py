"""Data parallelism for the RAFT step (row (e) of SURVEY.md section 8): one process per GPU,
image pairs sharded on the batch dimension, ONE exchange per optimizer step -- an all-reduce
(sum, then 1/world) of the fp32 gradients over RCCL/xGMI.  Replaces the reference's
single-process nn.DataParallel (pytorch/train.py:192: per step broadcast of all parameters,
scatter, gather of 12-24 full-resolution outputs to GPU 0, reduce of gradients to GPU 0) and
tf.distribute.MirroredStrategy (train.py:75-78).

Gradients live in one flat fp32 buffer (every ``p.grad`` is a view into it), so the exchange is
a single 21 MB collective with no packing copies: on 8 fully connected MI355X (7 xGMI links of
~153 GB/s each) that is ~0.1-0.3 ms against a >100 ms step, which is why there is no bucketing
or overlap machinery here -- it would be hiding 0.2 % of the step.  Gradient clipping and AdamW
then run identically on every rank on identical tensors, so no second collective is needed.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(device_type=None, backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank).
    backend: nccl (= RCCL on ROCm) for GPU tensors, gloo for CPU tests; backend="gloo" with GPU tensors (several ranks
    sharing one device, which RCCL refuses -- the 2-process GPU test) stages the exchange through host memory."""
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
            dist.init_process_group("gloo", rank=rank, world_size=world)
        elif device_type == "cuda":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local


def shard_batch(global_batch, rank, world):
    """Contiguous split of `global_batch` pairs over ranks; sizes differ by at most one."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


class FlatGradients:
    """All gradients of `params` as views into one contiguous fp32 buffer."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean_(self, local_batch=None, global_batch=None):
        """Gradient of the GLOBAL-batch mean loss from per-rank gradients of local-batch mean losses: every rank's
        gradient is weighted by local_batch / global_batch before the sum (equal shards: 1 / world, what DataParallel's
        reduce + batch-mean loss give).  `shard_batch` hands out unequal shards when the global batch does not divide."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            w = 1.0 / dist.get_world_size() if local_batch is None else float(local_batch) / float(global_batch)
            self.flat.mul_(w)
            _all_reduce_sum_(self.flat)

    def clip_norm_(self, max_norm):
        """clip_grad_norm_(params, max_norm) on the flat buffer (pytorch/train.py:280)."""
        total = self.flat.norm()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        self.flat.mul_(coef)
        return total


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
