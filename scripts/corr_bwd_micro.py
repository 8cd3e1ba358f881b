#!/usr/bin/env python3
"""The correlation path's backward at the bench shape (4 x 55x128, C = 256, 12 lookups with step-like flows) under the k-slice
settings of the two list GEMMs: per-family times from the bench's own KernelTimer."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import ops  # noqa: E402
from flow_supervisor_amd.core.corr import CorrBlock  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
B, C, H, W, T = 4, 256, 55, 128, 12
f1 = torch.randn(B, C, H, W, device=dev, requires_grad=True)
f2 = torch.randn(B, C, H, W, device=dev, requires_grad=True)
base = torch.randn(B, 2, H, W, device=dev) * 2.0
flows = [base + 0.25 * i + 0.1 * torch.randn(B, 2, H, W, device=dev) for i in range(T)]
gouts = [torch.randn(B, H, W, 324, device=dev) for _ in range(T)]


def step():
    blk = CorrBlock(f1, f2, radius=4)
    outs = [blk(fl, channels_last=True, is_flow=True) for fl in flows]
    torch.autograd.backward(outs, gouts)
    f1.grad = f2.grad = None


for nt, tn in ((1, 2), (2, 2), (1, 3), (1, 1), (2, 3), (3, 2)):
    ops.NT_LIST_KSPLIT, ops.TN_LIST_KSPLIT = nt, tn
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    tm = ops.KernelTimer()
    ops.TIMER = tm
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ops.TIMER = None
    sm = tm.summary()
    print(f"NT k-slices {nt}, TN k-slices {tn}: " + "  ".join(f"{k} {v['ms_total'] / 5 * 1e3:7.1f} us" for k, v in sm.items()))
