#!/usr/bin/env python3
"""hipGraph replay of the train step against eager steps from the same start: loss sequence and final parameters.
usage: python scripts/graph_check.py   (GPU box)"""
import argparse
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch  # noqa: E402

from flow_supervisor_amd.core.raft import RAFT  # noqa: E402
from flow_supervisor_amd.train import TrainStep  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train()
model.freeze_bn()
model2 = copy.deepcopy(model)
B, H, W = 2, 184, 320
g = torch.Generator(device=dev).manual_seed(1)
im1 = torch.rand(B, 3, H, W, device=dev, generator=g) * 255
im2 = torch.rand(B, 3, H, W, device=dev, generator=g) * 255
N = 6

eager = TrainStep(model2, lr=1e-4, iters=4, capturable=True)
le, snaps = [], []
for _ in range(2 + 1 + N):
    le.append(float(eager(im1, im2)))
    snaps.append([p.detach().clone() for p in model2.parameters()])

step = TrainStep(model, lr=1e-4, iters=4, capturable=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    lw = [float(step(im1, im2)) for _ in range(2)]
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    loss = step(im1, im2)
lg = []
for _ in range(N):
    graph.replay()
    torch.cuda.synchronize()
    lg.append(float(loss))
    ref = snaps[1 + len(lg)]                                            # (the capture itself executes nothing)
    if ref is not None:
        names = [n for n, _ in model.named_parameters()]
        errs = sorted(((p.detach() - q).abs().max().item() / (q.abs().max().item() + 1e-12), n) for p, q, n in zip(model.parameters(), ref, names))
        print(f"replay {len(lg)}: worst relative parameter differences vs eager step {2 + len(lg)}:", ", ".join(f"{n} {e:.2e}" for e, n in errs[-4:]))
print("eager :", " ".join(f"{v:.6f}" for v in le))
print("graph :", " ".join(f"{v:.6f}" for v in lw), "(capture)", " ".join(f"{v:.6f}" for v in lg))
worst = max((p.detach() - q.detach()).abs().max().item() for p, q in zip(model.parameters(), model2.parameters()))
print("max |param difference| after the same number of updates:", worst)
