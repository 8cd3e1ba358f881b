#!/usr/bin/env python3
"""Per-shape times of the convolution launches of one eager train step (bench configuration by default): which layers the
conv_igemm / conv_wgrad families' time sits in.  usage (GPU box, repo root): python scripts/layer_times.py [bench-like args]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import ops  # noqa: E402
from flow_supervisor_amd.core.raft import RAFT  # noqa: E402
from flow_supervisor_amd.train import TrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--height", type=int, default=440)
ap.add_argument("--width", type=int, default=1024)
ap.add_argument("--iters", type=int, default=12)
a = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train()
model.freeze_bn()
g = torch.Generator(device=dev).manual_seed(1234)
im1 = torch.rand(a.batch, 3, a.height, a.width, device=dev, generator=g) * 255.0
im2 = (torch.roll(im1, shifts=(3, -5), dims=(2, 3)) + 2.0 * torch.randn(a.batch, 3, a.height, a.width, device=dev, generator=g)).clamp(0, 255)
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
