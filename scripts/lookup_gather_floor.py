#!/usr/bin/env python3
"""What the memory system delivers for the tiled lookup's access pattern (VERDICT r3 next #3c): the lookup kernel with its window
loads alone -- same addresses, same masks, same two queries of look-ahead per wave, nothing done with the data
(fsraft_set_lookup_policy(100)) -- against the full kernel, at the bench shape, every launch on a fresh flow field (as in the
train step: a lookup never meets its windows in the caches) and behind a cache flush."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402

lib = _lib.load()
B, H, W, C, r = 4, 55, 128, 256, 4
dev = "cuda"
torch.manual_seed(0)
f1, f2 = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
vol, lay = ops.corr_build_tiled(f1, f2, 4)
flows = [torch.randn(B, 2, H, W, device=dev) * 3.0 for _ in range(12)]
big = torch.empty(600 << 20, device=dev, dtype=torch.uint8)
ALG = B * H * W * (4 * 100 * 4 + 8 + 324 * 4)          # SURVEY.md 8d bytes of one lookup
READ = B * H * W * 4 * 100 * 4                           # ... of which window reads


def run(policy, flush):
    lib.fsraft_set_lookup_policy(policy)
    for f in flows[:2]:
        ops.corr_lookup_tiled_fwd(vol, lay, f, r, is_flow=True)
    torch.cuda.synchronize()
    tot = 0.0
    for f in flows:
        if flush:
            big.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.corr_lookup_tiled_fwd(vol, lay, f, r, is_flow=True)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / len(flows) * 1e3


for flush in (False, True):
    res = {}
    for rnd in range(3):
        for pol in (-1, 100):
            res.setdefault(pol, []).append(run(pol, flush))
    full, gather = sorted(res[-1])[1], sorted(res[100])[1]
    print(f"cache flush {flush!s:5s}: full lookup {full:6.1f} us = {ALG / full / 1e6:5.2f} TB/s algorithmic ({ALG / full / 8e6:.3f} of the HBM roof)   "
          f"window loads alone {gather:6.1f} us = {READ / gather / 1e6:5.2f} TB/s of window bytes; the loads are {gather / full:.2f} of the kernel")
lib.fsraft_set_lookup_policy(-1)
