#!/usr/bin/env python3
"""Fused alt-corr lookup at the KITTI shape (1 x 47 x 156, C = 256): tile kernel vs wave-per-query kernel, smooth and rough flow."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402

dev = "cuda"
lib = _lib.load()
B, C, H, W, r = 1, 256, 47, 156, 4
torch.manual_seed(0)
f1 = torch.randn(B, C, H, W, device=dev)
f2 = torch.randn(B, C, H, W, device=dev)
f1c = ops.nchw_to_nhwc(f1)
lv, x = [], f2
for _ in range(4):
    lv.append(ops.nchw_to_nhwc(x))
    x = F.avg_pool2d(x, 2, stride=2)


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, flow in (("smooth flow (3.3, -1.7) + 0.3 px noise", torch.tensor([3.3, -1.7], device=dev).view(1, 2, 1, 1) + 0.3 * torch.randn(B, 2, H, W, device=dev)),
                   ("rough flow (8 px noise)", 8.0 * torch.randn(B, 2, H, W, device=dev)),
                   ("zero flow", torch.zeros(B, 2, H, W, device=dev))):
    outs, ts = [], []
    for tile in (0, 1):
        lib.fsraft_set_alt_tile(tile)
        outs.append(ops.altcorr_fused_fwd(f1c, lv, flow, r, is_flow=True))
        ts.append(timeit(lambda: ops.altcorr_fused_fwd(f1c, lv, flow, r, is_flow=True)))
    lib.fsraft_set_alt_tile(1)
    err = (outs[0] - outs[1]).abs().max().item()
    print(f"{name:42s} wave-per-query {ts[0]:7.1f} us   tile {ts[1]:7.1f} us   max diff {err:.2e}")
    assert err < 2e-4
