"""NCHW <-> channels_last conversion of encoder-sized activations: torch strided copy vs the tiled fsraft transposes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow_supervisor_amd import ops
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for shp in [(8, 64, 220, 512), (8, 96, 110, 256), (8, 128, 55, 128), (8, 256, 55, 128)]:
    x = torch.randn(*shp, device="cuda"); B, C, H, W = shp
    xc = x.contiguous(memory_format=torch.channels_last)
    dst = torch.empty(B, H, W, C, device="cuda"); dn = torch.empty(B, C, H, W, device="cuda")
    mb = x.numel() * 8 / 1e6
    a = t(lambda: x.contiguous(memory_format=torch.channels_last)); b = t(lambda: ops.nchw_to_nhwc(x, dst))
    c = t(lambda: xc.contiguous()); d = t(lambda: ops.nhwc_to_nchw(xc.permute(0, 2, 3, 1), dst=dn))
    print(shp, f"to_cl torch {a:.0f} us ({mb/a:.2f} TB/s)  fsraft {b:.0f} us ({mb/b:.2f} TB/s) | to_nchw torch {c:.0f} us  fsraft {d:.0f} us ({mb/d:.2f} TB/s)")
