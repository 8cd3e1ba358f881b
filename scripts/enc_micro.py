"""Encoder forward+backward timings (ms) on the bench shapes for FSRAFT_ENCODER_CL = 0 / 1 / 2.
usage: python scripts/enc_micro.py [B H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow_supervisor_amd.core.extractor import BasicEncoder

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (4, 440, 1024)
dev = "cuda"
torch.manual_seed(0)
fnet = BasicEncoder(256, "instance").to(dev)
cnet = BasicEncoder(256, "batch").to(dev).eval()
x1 = torch.randn(B, 3, H, W, device=dev); x2 = torch.randn(B, 3, H, W, device=dev)


def run(enc, xs):
    out = enc(xs)
    out = torch.cat(out) if isinstance(out, (tuple, list)) else out
    out.square().mean().backward()


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("PYTORCH_MIOPEN_SUGGEST_NHWC =", os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC"))
for mode in os.environ.get("ENC_MODES", "0,1").split(","):
    os.environ["FSRAFT_ENCODER_CL"] = mode
    t0 = time.time()
    tf = timeit(lambda: run(fnet, [x1, x2]))
    tc = timeit(lambda: run(cnet, x1))
    print(f"mode {mode}: fnet {tf:.2f} ms  cnet {tc:.2f} ms  total {tf + tc:.2f} ms   (wall {time.time() - t0:.0f}s)", flush=True)
