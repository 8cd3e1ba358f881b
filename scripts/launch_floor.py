import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow_supervisor_amd import ops
dev = "cuda"
def graph_time(fn, reps=200):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / 5 * 1e6
x = torch.zeros(1024, device=dev)
print("add_ on 1K floats            %.2f us" % graph_time(lambda: x.add_(1.0)))
y = torch.zeros(4, 55, 128, 128, device=dev)
print("add_ on 14 MB                %.2f us" % graph_time(lambda: y.add_(1.0)))
fl = torch.zeros(4, 2, 55, 128, device=dev); dd = torch.zeros(4, 55, 128, 4, device=dev)
print("flow_to_nhwc (own, tiny)     %.2f us" % graph_time(lambda: ops.flow_to_nhwc(fl, dd, 0)))
