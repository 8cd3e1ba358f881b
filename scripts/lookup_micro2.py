#!/usr/bin/env python3
"""Tiled-row lookup forward at the bench shape under the inputs the training step gives it (flow input, small / zero flow)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow_supervisor_amd import ops
from flow_supervisor_amd.core.utils.utils import coords_grid
B, H, W, C, r = 4, 55, 128, 256, 4
dev = "cuda"
torch.manual_seed(0)
f1, f2 = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
vol, lay = ops.corr_build_tiled(f1, f2, 4)
big = torch.empty(600 << 20, device=dev, dtype=torch.uint8)      # evicts the caches between launches


def timeit(fn, n=10, flush=False):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        if flush:
            big.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


for name, flow in (("random +-8 px", (torch.rand(B, 2, H, W, device=dev) - 0.5) * 16), ("zero flow", torch.zeros(B, 2, H, W, device=dev)),
                   ("random +-0.5 px", torch.rand(B, 2, H, W, device=dev) - 0.5)):
    coords = coords_grid(B, H, W, device=dev) + flow
    for flush in (False, True):
        a = timeit(lambda: ops.corr_lookup_tiled_fwd(vol, lay, coords, r), flush=flush)
        b = timeit(lambda: ops.corr_lookup_tiled_fwd(vol, lay, flow, r, is_flow=True), flush=flush)
        print(f"{name:16s} cache flush {flush!s:5s}: coords input {a:6.1f} us   flow input {b:6.1f} us")
