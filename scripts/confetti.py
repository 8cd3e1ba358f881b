#!/usr/bin/env python3
"""Where the framework's own small launches of one train step come from.
Part 1: every kernel of one profiled step that is not one of this library's (at::native, memcpy, memset, MIOpen helpers),
counted by name.  Part 2: every non-view ATen op of one step seen by a dispatch mode, grouped by the innermost frame of
this repository on the Python stack (ops issued by the autograd engine itself have no such frame).
usage: python scripts/confetti.py [raft|gma|alt|l2l]   (GPU box)"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from flow_supervisor_amd.train import TrainStep  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "raft"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
if variant == "gma":
    from flow_supervisor_amd.core.gma_network import RAFTGMA
    model = RAFTGMA(argparse.Namespace(mixed_precision=False, num_heads=1, position_only=False, position_and_content=False))
elif variant == "l2l":
    from flow_supervisor_amd.core.l2l import L2L
    model = L2L(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False))
else:
    from flow_supervisor_amd.core.raft import RAFT
    model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=variant == "alt"))
model = model.to(dev).train()
model.freeze_bn()
if variant == "l2l":                      # the flow-supervisor step of bench.py --variant l2l (one labelled + one unlabelled sample)
    from flow_supervisor_amd.train import SemiTrainStep
    sstep = SemiTrainStep(model, lr=5e-6, wdecay=0.0, iters=12, gamma=0.8, unsup_lambda=1.0)

    def sample(oy, ox):
        f1, f2 = torch.rand(1, 3, 432, 1024, device=dev) * 255, torch.rand(1, 3, 432, 1024, device=dev) * 255
        c1, c2 = f1[:, :, oy:oy + 368, ox:ox + 768].contiguous(), f2[:, :, oy:oy + 368, ox:ox + 768].contiguous()
        return (c1, c2, f1, f2, ox, oy, torch.randn(1, 2, 368, 768, device=dev), torch.ones(1, 368, 768, device=dev))
    sup, unsup = sample(40, 136), sample(16, 200)
    step = lambda _a, _b: sstep(sup, unsup)
    im1 = im2 = None
else:
    step = TrainStep(model, lr=1.6e-5, iters=12)
    B, H, W = (1, 376, 1248) if variant == "alt" else (4, 440, 1024)
    im1 = torch.rand(B, 3, H, W, device=dev) * 255
    im2 = torch.rand(B, 3, H, W, device=dev) * 255
for _ in range(3):
    step(im1, im2)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(im1, im2)
    torch.cuda.synchronize()

by_kernel = collections.Counter()
us = collections.Counter()
n_ours = 0
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU:
        continue
    nm = ev.name
    if nm.startswith("(anonymous namespace)") or nm.startswith("void (anonymous namespace)"):
        n_ours += 1
        continue
    by_kernel[nm[:100]] += 1
    us[nm[:100]] += ev.device_time
print(f"kernels of this library: {n_ours}; everything else: {sum(by_kernel.values())} launches, {sum(us.values())/1e3:.2f} ms")
for k, n in by_kernel.most_common(40):
    print(f"{n:5d} {us[k]:8.0f} us  {k}")

VIEWS = ("view", "reshape", "as_strided", "select", "slice", "transpose", "permute", "expand", "detach", "alias", "unsqueeze",
         "squeeze", "empty", "t.default", "unbind", "split", "chunk", "narrow", "_local_scalar", "sym_", "size", "stride",
         "storage_offset", "is_", "numel", "dim", "_to_copy", "lift_fresh", "unfold", "contiguous", "resize_", "set_", "record_stream")


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            site = "(no repository frame: autograd engine / optimizer)"
            for fr in reversed(traceback.extract_stack(limit=24)):
                if fr.filename.startswith(ROOT) and "/scripts/" not in fr.filename:
                    site = f"{fr.filename[len(ROOT) + 1:]}:{fr.lineno} {fr.name}"
                    break
            self.c[(site, name)] += 1
        return func(*args, **(kwargs or {}))


with Sites() as s:
    step(im1, im2)
    torch.cuda.synchronize()
print()
print(f"non-view ATen ops of one step: {sum(s.c.values())}")
for (site, name), n in s.c.most_common(70):
    print(f"{n:5d}  {name:42s} {site}")
