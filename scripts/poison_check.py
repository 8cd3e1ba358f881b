#!/usr/bin/env python3
"""Does any kernel read a torch.empty buffer before writing it?  One eager train step from fixed parameters with the
allocator's free blocks filled with NaN, against the same step with them filled with zeros."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch
from flow_supervisor_amd.core.raft import RAFT
from flow_supervisor_amd.train import TrainStep

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train()
model.freeze_bn()
B, H, W = 2, 184, 320
g = torch.Generator(device=dev).manual_seed(1)
im1 = torch.rand(B, 3, H, W, device=dev, generator=g) * 255
im2 = torch.rand(B, 3, H, W, device=dev, generator=g) * 255
step = TrainStep(model, lr=0.0, iters=4)          # lr 0: every step starts from the same parameters
names = [n for n, _ in model.named_parameters()]
for _ in range(2):
    step(im1, im2)


def poison(value):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = [torch.full((64 << 20,), value, device=dev) for _ in range(24)]      # 24 x 256 MB
    small = [torch.full((n,), value, device=dev) for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20) for _ in range(64)]
    del blocks, small
    torch.cuda.synchronize()


res = {}
for tag, v in (("zeros", 0.0), ("nan", float("nan")), ("big", 1e30)):
    poison(v)
    loss = float(step(im1, im2))
    res[tag] = (loss, {n: step.grads.views[p].clone() for n, p in zip(names, model.parameters()) if p in step.grads.views})
    print(tag, "loss", loss)
for tag in ("nan", "big"):
    bad = [(n, (res[tag][1][n] - res["zeros"][1][n]).abs().max().item() / (res["zeros"][1][n].abs().max().item() + 1e-20)) for n in res["zeros"][1]]
    bad = [(n, e) for n, e in bad if not (e < 1e-3)]
    print(tag, "gradients that differ from the zero-filled run:", len(bad), bad[:12])
