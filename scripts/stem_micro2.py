#!/usr/bin/env python3
"""csrc/stem.hip at the bench shapes: forward and weight gradient, us per call.  usage: python scripts/stem_micro2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow_supervisor_amd import ops

dev = "cuda"
for B in (8, 4):
    x = torch.rand(B, 3, 440, 1024, device=dev)
    w = torch.randn(64, 3, 7, 7, device=dev) * 0.05
    y = ops.stem_fwd(x, w)
    dy = torch.randn_like(y)
    for name, fn in (("fwd", lambda: ops.stem_fwd(x, w)), ("wgrad", lambda: ops.stem_wgrad(x, dy))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        gb = 4.0 * (x.numel() + y.numel()) / 1e9
        print(f"B={B} {name:5s} {us:7.1f} us   {gb / us * 1e6 / 1e3:5.2f} TB/s of the {gb * 1e3:.0f} MB it has to move")
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn(); torch.cuda.synchronize()
        for ev in prof.events():
            if ev.device_type == torch.autograd.DeviceType.CUDA:
                print(f"         {ev.device_time:8.1f} us  {ev.name[:80]}")
