"""Debug helper (not collected by pytest): device memory still allocated after each eager train step -- must stay flat.
usage: python tests/debug_leak.py [raft|l2l]"""
import argparse
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd.core.raft import RAFT  # noqa: E402
from flow_supervisor_amd.train import TrainStep  # noqa: E402

m = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).cuda().train()
m.freeze_bn()
st = TrainStep(m, lr=1e-5, iters=12)
a, b = torch.rand(2, 3, 256, 384, device="cuda") * 255, torch.rand(2, 3, 256, 384, device="cuda") * 255
prev = None
for i in range(14):
    st(a, b)
    torch.cuda.synchronize()
    gc.collect()
    now = torch.cuda.memory_allocated()
    if i >= 2:
        print(f"step {i}: allocated {now / 2**20:9.2f} MiB   delta {(now - prev) / 2**20 if prev is not None else 0:8.2f} MiB")
    prev = now
# what is alive: the largest live tensors
import collections
c = collections.Counter()
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.is_cuda:
            c[(tuple(o.shape), o.dtype)] += 1
    except Exception:
        pass
for (shape, dt), n in sorted(c.items(), key=lambda kv: -kv[1] * max(1, torch.Size(kv[0][0]).numel()))[:12]:
    print(n, shape, dt)
