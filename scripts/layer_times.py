#!/usr/bin/env python3
"""Per-shape times of the convolution launches of one eager train step (bench configuration by default): which layers the
conv_igemm / conv_wgrad families' time sits in.  usage (GPU box, repo root): python scripts/layer_times.py [bench-like args]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import ops  # noqa: E402
from flow_supervisor_amd.core import streams  # noqa: E402
from flow_supervisor_amd.core.raft import RAFT  # noqa: E402
from flow_supervisor_amd.train import TrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--height", type=int, default=440)
ap.add_argument("--width", type=int, default=1024)
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--variant", default="raft", choices=["raft", "l2l"], help="l2l: the flow-supervisor step (one labelled + one "
                "unlabelled sample per --batch, crops 368x768 of 432x1024 frames unless --height/--width say otherwise)")
a = ap.parse_args()
if a.variant == "l2l" and (a.batch, a.height, a.width) == (4, 440, 1024):
    a.batch, a.height, a.width = 1, 432, 1024
dev = torch.device("cuda")
streams.OVERLAP = False        # one stream: a launch's events bracket that launch alone
torch.manual_seed(0)
if a.variant == "l2l":
    from flow_supervisor_amd.core.l2l import L2L
    model = L2L(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train()
else:
    model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train()
model.freeze_bn()
g = torch.Generator(device=dev).manual_seed(1234)
im1 = torch.rand(a.batch, 3, a.height, a.width, device=dev, generator=g) * 255.0
im2 = (torch.roll(im1, shifts=(3, -5), dims=(2, 3)) + 2.0 * torch.randn(a.batch, 3, a.height, a.width, device=dev, generator=g)).clamp(0, 255)
if a.variant == "l2l":
    from flow_supervisor_amd.train import SemiTrainStep
    sstep = SemiTrainStep(model, lr=5e-6, wdecay=0.0, iters=a.iters, gamma=0.8, unsup_lambda=1.0)
    ch, cw, B = 368, 768, a.batch

    def sample(f1, f2, oy, ox):
        c1 = f1[:, :, oy:oy + ch, ox:ox + cw].contiguous()
        c2 = f2[:, :, oy:oy + ch, ox:ox + cw].contiguous()
        flow = torch.randn(B, 2, ch, cw, device=dev, generator=g) * 4.0
        valid = (torch.rand(B, ch, cw, device=dev, generator=g) > 0.1).float()
        return (c1, c2, f1, f2, ox, oy, flow, valid)
    sup, unsup = sample(im1, im2, 40, 136), sample(im2, im1, 16, 200)

    def step(_a, _b):
        return sstep(sup, unsup)
else:
    step = TrainStep(model, lr=1.6e-5, iters=a.iters)
for _ in range(3):
    step(im1, im2)
torch.cuda.synchronize()
timer = ops.KernelTimer(detail=True)
ops.TIMER = timer
NS = 3
for _ in range(NS):
    step(im1, im2)
torch.cuda.synchronize()
ops.TIMER = None
sm = timer.summary()
tot = {f: v for f, v in sm.items() if "|" not in f}
print("family totals (ms per step):", {f: round(v["ms_total"] / NS, 3) for f, v in sorted(tot.items(), key=lambda kv: -kv[1]["ms_total"])})
rows = [(f, v) for f, v in sm.items() if "|" in f]
rows.sort(key=lambda kv: -kv[1]["ms_total"])
print(f"{'ms/step':>8s} {'n/step':>6s} {'us each':>8s} {'TF/s':>6s}  family | shape")
for f, v in rows:
    print(f"{v['ms_total'] / NS:8.3f} {v['launches'] / NS:6.1f} {1e3 * v['ms_avg']:8.1f} {v['flops'] / (v['ms_total'] * 1e-3) / 1e12:6.1f}  {f}")
