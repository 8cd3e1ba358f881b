#!/usr/bin/env python3
"""The encoders' 7x7 stride-2 stem (3 -> 64) on MIOpen: NCHW tensors against channels_last tensors, forward and
backward-weights, kernels by name.  usage: python scripts/stem_micro.py   (GPU box)"""
import os
import sys

os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch
import torch.nn.functional as F
from torch.profiler import ProfilerActivity, profile

dev = "cuda"
torch.manual_seed(0)
for B in (8, 4):
    x = torch.rand(B, 3, 440, 1024, device=dev)
    w = (torch.randn(64, 3, 7, 7, device=dev) * 0.05).requires_grad_()
    for tag, xx, ww in (("nchw", x, w), ("channels_last", x.contiguous(memory_format=torch.channels_last), w)):
        def run():
            wl = ww.contiguous(memory_format=torch.channels_last) if tag == "channels_last" else ww
            y = F.conv2d(xx, wl, None, 2, 3)
            g = torch.ones_like(y)
            (gw,) = torch.autograd.grad(y, ww, g)
            return y, gw
        for _ in range(3):
            y, gw = run()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                y, gw = run()
            torch.cuda.synchronize()
        tot = 0.0
        rows = {}
        for ev in prof.events():
            if ev.device_type == torch.autograd.DeviceType.CUDA:
                rows[ev.name[:90]] = rows.get(ev.name[:90], 0.0) + ev.device_time / 3
                tot += ev.device_time / 3
        print(f"B={B} {tag}: {tot:.0f} us per forward + backward-weights; output contiguous as {'channels_last' if y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous() else 'nchw'}")
        for k, v in sorted(rows.items(), key=lambda kv: -kv[1])[:8]:
            print(f"      {v:8.1f} us  {k}")
        if tag == "nchw":
            ref = (y.clone(), gw.clone())
        else:
            print("      max |y - y_nchw|", (y - ref[0]).abs().max().item(), " max |gw - gw_nchw| / max|gw|", ((gw - ref[1]).abs().max() / ref[1].abs().max()).item())
