"""Debug helper (not collected by pytest): device memory still allocated after each eager train step -- must stay flat.
usage: python tests/debug_leak.py [raft|alt|gma|l2l]   (LEAK_NO_GC=0 keeps the collector on)"""
import argparse
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd.core.raft import RAFT  # noqa: E402
from flow_supervisor_amd.train import TrainStep  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "raft"
if which == "l2l":
    from flow_supervisor_amd.core.l2l import L2L
    from flow_supervisor_amd.train import SemiTrainStep
    m = L2L(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).cuda().train()
    m.freeze_bn()
    sst = SemiTrainStep(m, lr=1e-5, wdecay=0.0, iters=4)

    def sample(oy, ox):
        f1, f2 = torch.rand(1, 3, 256, 384, device="cuda") * 255, torch.rand(1, 3, 256, 384, device="cuda") * 255
        c1, c2 = f1[:, :, oy:oy + 192, ox:ox + 256].contiguous(), f2[:, :, oy:oy + 192, ox:ox + 256].contiguous()
        return (c1, c2, f1, f2, ox, oy, torch.randn(1, 2, 192, 256, device="cuda"), torch.ones(1, 192, 256, device="cuda"))
    sup, unsup = sample(8, 16), sample(24, 40)
    st = lambda a, b: sst(sup, unsup)
elif which == "gma":
    from flow_supervisor_amd.core.gma_network import RAFTGMA
    m = RAFTGMA(argparse.Namespace(mixed_precision=False, num_heads=1, position_only=False, position_and_content=False)).cuda().train()
    m.freeze_bn()
    st = TrainStep(m, lr=1e-5, iters=4)
else:
    m = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=which == "alt")).cuda().train()
    m.freeze_bn()
    st = TrainStep(m, lr=1e-5, iters=12)
a, b = torch.rand(2, 3, 256, 384, device="cuda") * 255, torch.rand(2, 3, 256, 384, device="cuda") * 255
gc.collect()
if os.environ.get("LEAK_NO_GC", "1") == "1":
    gc.disable()          # growth without the cyclic collector = tensors sitting in reference cycles
prev = None
for i in range(14):
    st(a, b)
    torch.cuda.synchronize()
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
