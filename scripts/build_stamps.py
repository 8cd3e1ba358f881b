#!/usr/bin/env python3
"""Where a workgroup of the volume build spends its time: corr_build_rec_t_kernel<STAMP> (experiment object, built by
`hipcc -DFSRAFT_EXPERIMENTS corr_build.hip` into libfsraft_buildexp.so) records s_memtime at the phase boundaries of every
workgroup and the id of its CU.  Prints the median phases and, per CU, the gaps between consecutive workgroups."""
import ctypes
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402

_lib.load()
exp = ctypes.CDLL(os.path.join(ROOT, "flow_supervisor_amd", "libfsraft_buildexp.so"))
dev = "cuda"
B, C, H, W = 4, 256, 55, 128
torch.manual_seed(0)
f1 = torch.randn(B, C, H, W, device=dev)
f2 = torch.randn(B, C, H, W, device=dev)
recs = (ops.fmap_records(f1), ops.fmap_records(f2))
lay = ops.VolLayout.get(H, W, 4)
vol = torch.empty(B * H * W, lay.P, device=dev)
nwg = ((W + 31) // 32) * ((H + 7) // 8) * ((H * W + 127) // 128) * B
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
fn = exp.fsraft_corr_build_rec_stamps
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
for mode, label in ((0, "as shipped"), (64, "stores ablated")):
  for _ in range(3):
    rc = fn(recs[0].data_ptr(), recs[1].data_ptr(), vol.data_ptr(), 4 | mode << 8, B, C, H, W, stamps.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(5):
    fn(recs[0].data_ptr(), recs[1].data_ptr(), vol.data_ptr(), 4 | mode << 8, B, C, H, W, stamps.data_ptr(), torch.cuda.current_stream().cuda_stream)
  e1.record()
  torch.cuda.synchronize()
  print(f"== {label}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch")
  s = stamps.cpu().numpy().reshape(nwg, 8).astype(np.int64)
  t0, t1, t2, t3, t4, r0, r1, hw = (s[:, i] for i in range(8))
  clk = (t4 - t0) / np.maximum(r1 - r0, 1) * 100e6          # s_memrealtime ticks at 100 MHz
  print(f"{nwg} workgroups; shader clock during a workgroup: median {np.median(clk)/1e9:.2f} GHz")
  us = lambda c: c / np.median(clk) * 1e6
  for name, d in (("setup (descriptors, offsets)", t1 - t0), ("k-loop (8 k-tiles, incl. first-tile latency)", t2 - t1),
                  ("epilogue issue (pooling + 45 stores/wave)", t3 - t2), ("store drain (vmcnt 0)", t4 - t3), ("whole workgroup", t4 - t0)):
    print(f"  {name:48s} median {np.median(us(d)):6.2f} us   p10 {np.percentile(us(d), 10):6.2f}   p90 {np.percentile(us(d), 90):6.2f}")
