#!/usr/bin/env python3
"""Host-side cost of one train step: time to ENQUEUE a step (no synchronisation) against the GPU time of the step.
If the first approaches the second the step is host-bound (matters with 8 processes per node sharing the host's cores)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch
from flow_supervisor_amd.core.raft import RAFT
from flow_supervisor_amd.train import TrainStep
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train()
model.freeze_bn()
step = TrainStep(model, lr=1.6e-5, iters=12)
im1 = torch.rand(4, 3, 440, 1024, device=dev) * 255
im2 = torch.rand(4, 3, 440, 1024, device=dev) * 255
for _ in range(3):
    step(im1, im2)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(5):
    t0 = time.perf_counter()
    step(im1, im2)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print(f"enqueue {1e3 * sum(enq) / 5:.1f} ms per step, step {1e3 * sum(tot) / 5:.1f} ms  (host threads: {torch.get_num_threads()}, cores: {os.cpu_count()})")
