#!/usr/bin/env python3
"""Micro-benchmark of the update-block GEMM shapes at the bench size (M = 4*55*128 pixels).
Prints per-layer time and fp32-MFMA efficiency; meant to run under rocprofv3 as well."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import ops  # noqa: E402
from flow_supervisor_amd.ops import Dst, V  # noqa: E402

B, H, W = (int(v) for v in os.environ.get("CONV_MICRO_BHW", "4,55,128").split(","))
GRAPH = os.environ.get("CONV_MICRO_GRAPH", "0") == "1"
NSEG = int(os.environ.get("CONV_MICRO_NSEG", "1"))   # wgrad: this many (dY, X) segments in one multi launch
M = B * H * W
dev = "cuda"
LAYERS = [  # name, kh, kw, src channels, Cout
    ("c1 1x1 324->256", 1, 1, [324], 256),
    ("c2 3x3 256->192", 3, 3, [256], 192),
    ("cv 3x3 256->126", 3, 3, [256], 126),
    ("zr 1x5 384->256", 1, 5, [128, 128, 128], 256),
    ("q  1x5 384->128", 1, 5, [128, 128, 128], 128),
    ("hd 3x3 128->512", 3, 3, [128], 512),
    ("m2 1x1 256->576", 1, 1, [256], 576),
    ("fh2 3x3 256->2", 3, 3, [256], 2),
    # data-gradient shapes of the same layers (K = Cout * taps of the forward layer, N = its input channels)
    ("dg m2 1x1 576->256", 1, 1, [576], 256),
    ("dg fh2 3x3 2->256", 3, 3, [2], 256),
    ("dg hd 3x3 512->128", 3, 3, [512], 128),
    ("dg q 1x5 128->256", 1, 5, [128], 256),
    ("dg zr 1x5 256->256", 1, 5, [256], 256),
    ("dg cv 3x3 126->256", 3, 3, [126], 256),
    ("dg f2 3x3 64->128", 3, 3, [64], 128),
    ("dg f1 1x1 128->98", 1, 1, [128], 98),
    ("dg c2 3x3 192->256", 3, 3, [192], 256),
    ("dg c1 1x1 256->324", 1, 1, [256], 324),
    ("f2 3x3 128->64", 3, 3, [128], 64),
    ("e1 3x3 64->64 (encoder layer1)", 3, 3, [64], 64),
    ("e2 3x3 96->96 (encoder layer2)", 3, 3, [96], 96),
    ("e3 3x3 128->128 (encoder layer3)", 3, 3, [128], 128),
    ("f1 1x1 98->128", 1, 1, [98], 128),
    ("zrc 1x5 256->256 (ctx split)", 1, 5, [128, 128], 256),
    ("qc 1x5 256->128 (ctx split)", 1, 5, [128, 128], 128),
    # diagnostics: same K as zr, different reuse structure
    ("x1 1x1 1920->256", 1, 1, [1920], 256),
    ("x5 1x5 384->256 one src", 1, 5, [384], 256),
    ("x9 3x3 224->256", 3, 3, [224], 256),
    ("xw 1x5 384->512", 1, 5, [384], 512),
    # fixed cost of a launch: one k-tile, 128 / 256 / 512 outputs
    ("y1 1x1 32->128", 1, 1, [32], 128),
    ("y2 1x1 32->256", 1, 1, [32], 256),
    ("y3 1x1 32->512", 1, 1, [32], 512),
    ("y4 1x1 256->256", 1, 1, [256], 256),
    ("y5 1x1 512->256", 1, 1, [512], 256),
    ("y6 1x1 1024->256", 1, 1, [1024], 256),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
from flow_supervisor_amd import _lib  # noqa: E402
for kv in sys.argv[3:]:                      # tuning overrides as key=value (fsraft_set_tuning keys)
    k, v = kv.split("=")
    _lib.load().fsraft_set_tuning(int(k), int(v))
print("tuning:", sys.argv[3:])
for name, kh, kw, cs, cout in LAYERS:
    if only and not name.startswith(only):
        continue
    cin = sum(cs)
    srcs = [V(torch.randn(B, H, W, (c + 3) // 4 * 4, device=dev), c) for c in cs]
    w = torch.randn(cout, cin, kh, kw, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    wpk = ops.pack_weight(w, cs, 0)
    wps = ops.pack_weight(w, cs, 10)
    wpf = ops.fragment_order(wps)
    out = torch.zeros(B, H, W, (cout + 3) // 4 * 4, device=dev)
    if os.environ.get("CONV_MICRO_TRACK", "1") == "1":
        ops.tracked(out)                 # (as in the step: the epilogue raises the destination's amax word)
    dy = torch.randn(B, H, W, (cout + 3) // 4 * 4, device=dev)
    dys = [V(torch.randn_like(dy), cout) for _ in range(NSEG)]
    xss = [[V(torch.randn_like(v.t), v.C) for v in srcs] for _ in range(NSEG)]
    dwpk = torch.zeros_like(wpk)
    if os.environ.get("CONV_MICRO_SHAREWORD") == "1":      # one amax word for all segments' dY and one per source (as a step-wide word would be)
        wdy = ops.amax_of_tensors([v.t for v in dys])
        for v in dys:
            v.amax = wdy
        for si in range(len(cs)):
            wsrc = ops.amax_of_tensors([xs[si].t for xs in xss])
            for xs in xss:
                xs[si].amax = wsrc
    for kind in ("fwd", "wgrad"):
        def run():
            if kind == "fwd":
                ops.conv_forward(srcs, wpk, bias, B, H, W, kh, kw, cout, [Dst.nhwc(out)], relu=True, wpk_split=wps, wpk_frag=wpf)
            else:
                ops.conv_wgrad_multi(dys, xss, dwpk, B, H, W, kh, kw)
        run()
        torch.cuda.synchronize()
        if GRAPH:                      # `reps` launches captured in one hipGraph: GPU time per launch without the host's ~10 us per call
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(reps):
                    run()
            gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps / 5
        else:
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
        fl = 2.0 * M * cout * cin * kh * kw * (NSEG if kind == 'wgrad' else 1)
        print(f"{name:18s} {kind:5s} {dt * 1e6:8.1f} us  {fl / dt / 1e12:6.1f} TF  ({fl / dt / 1e12 / 157.3 * 100:4.1f}% of fp32 MFMA peak)")
