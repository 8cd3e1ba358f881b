#!/usr/bin/env python3
"""Two replays of the captured train step from the SAME parameters and optimizer state: do they produce the same gradients?"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch
from flow_supervisor_amd.core.raft import RAFT
from flow_supervisor_amd.train import TrainStep

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train()
model.freeze_bn()
B, H, W = 2, 184, 320
g = torch.Generator(device=dev).manual_seed(1)
im1 = torch.rand(B, 3, H, W, device=dev, generator=g) * 255
im2 = torch.rand(B, 3, H, W, device=dev, generator=g) * 255
step = TrainStep(model, lr=1e-4, iters=4, capturable=True)
probe = {}


def hook_for(name):
    def hook(mod, gin, gout):
        for i, t in enumerate(gout):
            if t is None:
                continue
            k = f"{name}.gout{i}"
            if k not in probe:
                probe[k] = torch.empty_like(t)
            probe[k].copy_(t)
    return hook


for name, mod in model.named_modules():
    if name in ("fnet", "cnet", "fnet.layer1", "fnet.layer2", "fnet.layer3", "fnet.conv2", "fnet.norm1", "fnet.conv1", "fnet.layer1.0", "fnet.layer1.1", "fnet.layer2.0", "fnet.layer2.1", "fnet.layer3.0", "fnet.layer3.1"):
        mod.register_full_backward_hook(hook_for(name))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        step(im1, im2)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    loss = step(im1, im2)


def state():
    o = step.opt                                   # parallel.FlatAdamW: everything lives in four flat buffers
    return [o.p.clone(), o.exp_avg.clone(), o.exp_avg_sq.clone(), o.step_count.clone()]


def restore(st):
    o = step.opt
    with torch.no_grad():
        o.p.copy_(st[0]); o.exp_avg.copy_(st[1]); o.exp_avg_sq.copy_(st[2]); o.step_count.copy_(st[3])


s0 = state()
names = [n for n, _ in model.named_parameters()]
outs = []
for k in range(3):
    restore(s0)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    outs.append((float(loss), {n: step.grads.views[p].clone() for n, p in zip(names, model.parameters()) if p in step.grads.views}, {k: v.clone() for k, v in probe.items()}))
    print("replay", k, "loss", outs[-1][0])
for k in (1, 2):
    errs = sorted(((outs[k][1][n] - outs[0][1][n]).abs().max().item() / (outs[0][1][n].abs().max().item() + 1e-20), n) for n in outs[0][1])
    print(f"replay {k} vs replay 0 (same start): worst relative gradient differences:", ", ".join(f"{n} {e:.2e}" for e, n in errs[-6:]))
for k in sorted(outs[0][2]):
    a, b = outs[0][2][k], outs[1][2][k]
    print(f"  {k:28s} max|replay0| {a.abs().max().item():.3e}  max|replay1 - replay0| {(a - b).abs().max().item():.3e}")
# eager gradient from the same start
restore(s0)
le = float(step(im1, im2))
ge = {n: step.grads.views[p].clone() for n, p in zip(names, model.parameters()) if p in step.grads.views}
errs = sorted(((ge[n] - outs[0][1][n]).abs().max().item() / (ge[n].abs().max().item() + 1e-20), n) for n in ge)
print("eager loss", le, "eager vs replay 0: worst:", ", ".join(f"{n} {e:.2e}" for e, n in errs[-6:]))
