import os, sys, time, argparse, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flow_supervisor_amd.core.extractor import BasicEncoder
mode = sys.argv[1]
dev = "cuda"
torch.manual_seed(0)
fnet = BasicEncoder(256, "instance").to(dev).train()
cnet = BasicEncoder(256, "batch").to(dev).train()
for m in cnet.modules():
    if isinstance(m, torch.nn.BatchNorm2d): m.eval()
x = torch.rand(8, 3, 440, 1024, device=dev)
if mode == "cl":
    fnet = fnet.to(memory_format=torch.channels_last); cnet = cnet.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
if mode == "bench":
    torch.backends.cudnn.benchmark = True
def step():
    a = fnet(x); b = cnet(x[:4])
    (a.square().mean() + b.square().mean()).backward()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print(mode, "first step", time.perf_counter() - t0, "s")
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print(mode, "ms/step", (time.perf_counter() - t0) / 5 * 1e3)
