#!/usr/bin/env python3
"""Stage-removal experiment on the split-bf16 conv kernel (needs `make -C flow_supervisor_amd/csrc ablate`).
usage: FSRAFT_LIB_PATH=flow_supervisor_amd/csrc/build/ablate/libfsraft_ablate.so python scripts/ablate.py [buf]
Times the zr-shaped (1x5, 3 sources, 384 -> 256) and hd-shaped (3x3, 128 -> 512) GEMMs with pipeline stages removed:
bit 0 no LDS staging, bit 1 no MFMA, bit 2 no fragment reads + no MFMA, bit 3 no global loads."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402
from flow_supervisor_amd.ops import Dst, V  # noqa: E402

lib = _lib.load()
lib.fsraft_set_ablate.argtypes = [ctypes.c_int]
lib.fsraft_set_tuning(5, int(sys.argv[1]) if len(sys.argv) > 1 else 0)     # buffer-addressed loaders
# (argv[2] used to select a tap-inner k order experiment; removed from the library)
lib.fsraft_set_tuning(7, int(sys.argv[3]) if len(sys.argv) > 3 else 0)     # XCD-aware tile mapping
MASKS = [int(m) for m in sys.argv[4].split(",")] if len(sys.argv) > 4 else None
B, H, W = 4, 55, 128
M = B * H * W
dev = "cuda"
for name, kh, kw, cs, cout in (("zr 1x5 3x128->256", 1, 5, [128, 128, 128], 256), ("hd 3x3 128->512", 3, 3, [128], 512)):
    srcs = [V(torch.randn(B, H, W, c, device=dev), c) for c in cs]
    w = torch.randn(cout, sum(cs), kh, kw, device=dev) * 0.05
    wpk, wps = ops.pack_weight(w, cs, 0), ops.pack_weight(w, cs, 10)
    out = torch.zeros(B, H, W, cout, device=dev)
    for mask, what in ((0, "full"), (1, "no staging"), (2, "no MFMA"), (4, "no frag reads, no MFMA"), (8, "no global loads"),
                       (9, "no loads, no staging"), (6, "no reads/MFMA (=4|2)"), (5, "loads only"), (13, "nothing but loop"), (16, "A staged by copy (pre-split emulation)")):
        if MASKS is not None and mask not in MASKS:
            continue
        lib.fsraft_set_ablate(mask)

        def run():
            ops.conv_forward(srcs, wpk, None, B, H, W, kh, kw, cout, [Dst.nhwc(out)], wpk_split=wps)
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        print(f"{name:20s} mask {mask:2d} {what:28s} {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us")
