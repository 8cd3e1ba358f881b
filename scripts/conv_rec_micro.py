#!/usr/bin/env python3
"""Record-activation convolution kernel (csrc/conv_rec.inc) against the round-1 implicit-GEMM kernels: same outputs,
per-layer time at the bench size (M = 4*55*128 pixels) and on ragged shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402
from flow_supervisor_amd.ops import Dst, V  # noqa: E402

dev = "cuda"
lib = _lib.load()
LAYERS = [  # name, kh, kw, src channels, Cout
    ("c1 1x1 324->256", 1, 1, [324], 256),
    ("c2 3x3 256->192", 3, 3, [256], 192),
    ("cv 3x3 256->126", 3, 3, [256], 126),
    ("zr 1x5 256->256", 1, 5, [128, 128], 256),
    ("zr 5x1 256->256", 5, 1, [128, 128], 256),
    ("q  1x5 256->128", 1, 5, [128, 128], 128),
    ("hd 3x3 128->512", 3, 3, [128], 512),
    ("m2 1x1 256->576", 1, 1, [256], 576),
    ("dg m2 1x1 576->256", 1, 1, [576], 256),
    ("dg hd 3x3 512->128", 3, 3, [512], 128),
    ("dg q 1x5 128->256", 1, 5, [128], 256),
    ("dg zr 1x5 256->256", 1, 5, [256], 256),
    ("dg cv 3x3 126->256", 3, 3, [126], 256),
    ("dg c2 3x3 192->256", 3, 3, [192], 256),
    ("dg c1 1x1 256->324", 1, 1, [256], 324),
    ("f2 3x3 128->64", 3, 3, [128], 64),
    ("f1 1x1 98->128", 1, 1, [98], 128),
    ("gma zr 1x5 384->256", 1, 5, [128, 256], 256),
]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for (B, H, W) in ((2, 13, 21), (4, 55, 128)):
    M = B * H * W
    print(f"-- B={B} H={H} W={W}")
    for name, kh, kw, cs, cout in LAYERS:
        cin = sum(cs)
        bufs = [torch.randn(B, H, W, (c + 3) // 4 * 4, device=dev) for c in cs]
        for b, c in zip(bufs, cs):
            b[..., c:] = 0
        w = torch.randn(cout, cin, kh, kw, device=dev) * 0.05
        bias = torch.randn(cout, device=dev)
        wpk = ops.pack_weight(w, cs, 0) if ops.exact_mode() or cout <= 32 else None
        wps = ops.pack_weight(w, cs, 10)
        outs = []
        ts = []
        for rec in (0, 1, 2):
            srcs = [V(b, c, 0, ops.to_records(b, pad=rec == 2) if rec else None) for b, c in zip(bufs, cs)]
            out = torch.zeros(B, H, W, (cout + 3) // 4 * 4, device=dev)
            lib.fsraft_set_tuning(25, 2 if rec else 0)

            def run():
                ops.conv_forward(srcs, wpk if wpk is not None else wps, bias, B, H, W, kh, kw, cout, [Dst.nhwc(out)], relu=True, wpk_split=wps)
            run()
            torch.cuda.synchronize()
            outs.append(out.clone())
            if M > 10000:
                ts.append(timeit(run))
        lib.fsraft_set_tuning(25, 1)
        err = max((outs[0] - o).abs().max().item() for o in outs[1:]) / max(outs[0].abs().max().item(), 1e-9)
        fl = 2.0 * M * cout * cin * kh * kw
        extra = f"  round-1 {ts[0]*1e6:7.1f} us {fl/ts[0]/1e12:6.1f} TF   rec {ts[1]*1e6:7.1f} us   rec, odd-line pitch {ts[2]*1e6:7.1f} us {fl/ts[2]/1e12:6.1f} TF ({fl/ts[2]/1e12/833.3*100:4.1f} %)" if ts else ""
        print(f"{name:24s} rel diff {err:.2e}{extra}")
        assert err < 2e-5, (name, err)
