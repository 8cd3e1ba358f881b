import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flow_supervisor_amd.core.raft import RAFT
from flow_supervisor_amd.train import TrainStep
lr = float(sys.argv[1]); graph = int(sys.argv[2]); n = int(sys.argv[3])
dev = torch.device("cuda")
torch.manual_seed(0)
model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train(); model.freeze_bn()
step = TrainStep(model, lr=lr, iters=12, capturable=bool(graph))
g = torch.Generator(device=dev).manual_seed(1234)
im1 = torch.rand(4, 3, 440, 1024, device=dev, generator=g) * 255.0
im2 = (torch.roll(im1, shifts=(3, -5), dims=(2, 3)) + 2.0 * torch.randn(4, 3, 440, 1024, device=dev, generator=g)).clamp(0, 255)
out = []
if graph:
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): out.append(float(step(im1, im2)))
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    G = torch.cuda.CUDAGraph()
    with torch.cuda.graph(G, stream=side): loss = step(im1, im2)
    for _ in range(n): G.replay(); out.append(float(loss))
else:
    for _ in range(n): out.append(float(step(im1, im2)))
print("lr", lr, "graph", graph, " ".join(f"{x:.4f}" for x in out))
