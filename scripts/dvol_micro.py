#!/usr/bin/env python3
"""corr_dvol_kernel (the gradient volume of a step in one pass) at the benchmark shape: store cache policies A/B, plus the two
record GEMMs that read it right after (their time depends on what the stores left in the caches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)
B, C, H, W, r, T = 4, 256, 55, 128, 4, 12
lay = ops.VolLayout.get(H, W, 4)
douts = [torch.randn(B, H, W, 324, device=dev) for _ in range(T)]
flows = [torch.randn(B, 2, H, W, device=dev) * 3 for _ in range(T)]
f1 = torch.randn(B, C, H, W, device=dev)
f2 = torch.randn(B, C, H, W, device=dev)
f1r = ops.fmap_records(f1)


def timeit(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ref = None
res = {0: [], 1: [], 2: []}
resb = {0: [], 1: [], 2: []}
for rnd in range(4):
    for pol in (0, 1, 2):
        lib.fsraft_set_dvol_policy(pol)
        dv = ops.corr_dvol_build(douts, flows, lay, B, r, records=True, is_flow=True)
        if ref is None:
            ref = dv.clone()
        else:
            assert torch.equal(dv, ref), pol
        res[pol].append(timeit(lambda: ops.corr_dvol_build(douts, flows, lay, B, r, records=True, is_flow=True), 5))
        resb[pol].append(timeit(lambda: (ops.corr_dvol_build(douts, flows, lay, B, r, records=True, is_flow=True),
                                         ops.corr_build_bwd_tiled(f1, f2, dv, lay, records=True, f1r=f1r)), 3))
for pol in (0, 1, 2):
    print(f"policy {pol}: dvol {sorted(res[pol])[1]:7.1f} us   dvol + build backward {sorted(resb[pol])[1]:7.1f} us")
lib.fsraft_set_dvol_policy(0)
