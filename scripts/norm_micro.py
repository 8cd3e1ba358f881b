"""Channels-last norm kernels at encoder sizes: time and effective bandwidth per kernel pair, for several workgroup targets.
usage: python scripts/norm_micro.py [targets...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow_supervisor_amd import _lib
from flow_supervisor_amd.core.extractor import _InstNormReluCL, _FrozenBNReluCL

lib = _lib.load()
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for tgt in [int(v) for v in sys.argv[1:]] or [2048, 4096, 8192, 16384]:
    lib.fsraft_set_norm_blocks(tgt)
    for shp in [(8, 64, 220, 512), (4, 64, 220, 512), (8, 96, 110, 256), (8, 128, 55, 128)]:
        x = torch.randn(*shp, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        res = torch.randn(*shp, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        g = torch.randn(*shp, device="cuda").contiguous(memory_format=torch.channels_last)
        mb = x.numel() * 4 / 1e6
        C = shp[1]
        w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
        y = _InstNormReluCL.apply(x, 1e-5, True, res)
        f1 = t(lambda: _InstNormReluCL.apply(x, 1e-5, True, res))
        b1 = t(lambda: torch.autograd.grad(y, (x, res), g, retain_graph=True))
        y2 = _FrozenBNReluCL.apply(x, None, w, b, rm, rv, 1e-5, True, res)
        f2 = t(lambda: _FrozenBNReluCL.apply(x, None, w, b, rm, rv, 1e-5, True, res))
        b2 = t(lambda: torch.autograd.grad(y2, (x, res), g, retain_graph=True))
        print(f"target {tgt:6d} {shp}: IN fwd {f1:6.0f} us ({4 * mb / f1:.2f} TB/s)  bwd {b1:6.0f} us ({8 * mb / b1:.2f} TB/s) | "
              f"BN fwd {f2:6.0f} us ({3 * mb / f2:.2f} TB/s)  bwd {b2:6.0f} us ({5 * mb / b2:.2f} TB/s)", flush=True)
