#!/usr/bin/env python3
"""Error of one convolution layer (forward and weight gradient) in the exact-fp32 and the split arithmetic against float64,
at several operand magnitudes: `python scripts/conv_accuracy.py [amp_x amp_w ...]`.
Reports max |err| / rms(result) -- the error relative to the size of the results, which is what a dot product can promise."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402
from flow_supervisor_amd.ops import Dst, V  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
B, H, W = 2, 40, 64
LAYERS = [("3x3 256->192", 3, 3, 256, 192), ("1x5 384->256", 1, 5, 384, 256), ("1x1 324->256", 1, 1, 324, 256)]
amps = [(1.0, 0.05), (1e-3, 0.05), (300.0, 0.05), (1.0, 1e-4), (1e-6, 1e-3), (3e4, 1.0)]
lib = _lib.load()
for name, kh, kw, cin, cout in LAYERS:
    for ax, aw in amps:
        # heavy-tailed activations (a ReLU of a Gaussian times a log-normal factor): wide dynamic range inside the tensor
        x = (torch.randn(B, H, W, cin, device=dev).relu() * torch.exp(torch.randn(B, H, W, cin, device=dev))) * ax
        w = torch.randn(cout, cin, kh, kw, device=dev) * aw
        dy = torch.randn(B, H, W, cout, device=dev) * ax
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1)
        # weight gradient reference: dW[co][ci][t] = sum_p dy[p, co] x[p + t, ci]
        xd = x.permute(0, 3, 1, 2).double().requires_grad_(False)
        wd = w.double().requires_grad_(True)
        (F.conv2d(xd, wd, padding=(kh // 2, kw // 2)) * dy.permute(0, 3, 1, 2).double()).sum().backward()
        refw = wd.grad
        row = f"{name:13s} |x|~{ax:7.0e} |w|~{aw:6.0e}"
        for mode in (0, 1):
            ops.set_arithmetic(mode)
            wpk = ops.pack_weight(w, [cin], 0)
            wps = ops.pack_weight(w, [cin], 10)
            out = ops.tracked(torch.zeros(B, H, W, cout, device=dev))
            ops.conv_forward([V(x, cin)], wpk, None, B, H, W, kh, kw, cout, [Dst.nhwc(out)], wpk_split=wps)
            e = (out.double() - ref).abs().max() / ref.pow(2).mean().sqrt()
            got = ops.amax_of(out).item()
            assert 0.5 * out.abs().max().item() < got <= 4.0 * out.abs().max().item(), ("dst_amax", got, out.abs().max().item())
            dwpk = torch.zeros_like(wpk)
            ops.conv_wgrad(V(dy, cout), [V(x, cin)], dwpk, B, H, W, kh, kw)
            dw = ops.unpack_weight_grad(dwpk, w.shape, [cin])
            ew = (dw.double() - refw).abs().max() / refw.pow(2).mean().sqrt()
            row += f" | {'exact' if mode == 0 else 'split'} fwd {e.item():.2e} wgrad {ew.item():.2e}"
        print(row, flush=True)
ops.set_arithmetic(1)
