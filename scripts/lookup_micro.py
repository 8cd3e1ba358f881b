#!/usr/bin/env python3
"""Lookup fwd/bwd timing at the bench shape (B=4, 55x128, r=4) for each workgroup size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow_supervisor_amd import _lib, ops
from flow_supervisor_amd.core.utils.utils import coords_grid
B, C, H, W, r = 4, 256, 55, 128, 4
dev = "cuda"
f1 = torch.randn(B, C, H, W, device=dev); f2 = torch.randn(B, C, H, W, device=dev)
levels = ops.corr_build(f1, f2, 4)
coords = coords_grid(B, H, W, device=dev) + (torch.rand(B, 2, H, W, device=dev) - 0.5) * 8
dlv = [torch.zeros_like(l) for l in levels]
nq = B * H * W
fwd_b = 4.0 * nq * (400 + 2 + 324); bwd_b = 4.0 * nq * (324 + 2 + 800)
for nhwc in (True, False):
    for qb in ((8, 16, 208, 308) if nhwc else (8, 32)):
        _lib.load().fsraft_set_lookup_qb(qb)
        out = ops.corr_lookup_fwd(levels, coords, r, nhwc=nhwc)
        g = torch.randn_like(out)
        ops.corr_lookup_bwd_(dlv, coords, g, r, nhwc=nhwc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): ops.corr_lookup_fwd(levels, coords, r, nhwc=nhwc)
        torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20): ops.corr_lookup_bwd_(dlv, coords, g, r, nhwc=nhwc)
        torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 20
        print(f"nhwc={int(nhwc)} QB={qb:3d}  fwd {tf*1e6:7.1f} us {fwd_b/tf/1e9:7.0f} GB/s ({fwd_b/tf/8e12*100:4.1f}% of 8 TB/s)   bwd {tb*1e6:7.1f} us {bwd_b/tb/1e9:7.0f} GB/s ({bwd_b/tb/8e12*100:4.1f}%)")
