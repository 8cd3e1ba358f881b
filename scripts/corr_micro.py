#!/usr/bin/env python3
"""Correlation path at the bench shape (B=4, 55x128, C=256, r=4, 12 lookups): tiled-row kernels against the row-major
kernels of round 1 (same numbers, checked here), each timed in isolation with events.
usage: python scripts/corr_micro.py [B H W]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import ops  # noqa: E402
from flow_supervisor_amd.core.utils.utils import coords_grid  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (4, 55, 128)
C, r, T = 256, 4, 12
dev = "cuda"
torch.manual_seed(0)
f1 = torch.randn(B, C, H, W, device=dev)
f2 = torch.randn(B, C, H, W, device=dev)
coords = [coords_grid(B, H, W, device=dev) + (torch.rand(B, 2, H, W, device=dev) - 0.5) * 16 for _ in range(T)]
nq = B * H * W


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


levels = ops.corr_build(f1, f2, 4)
vol, lay = ops.corr_build_tiled(f1, f2, 4)
P = sum(h * w for h, w in zip(lay.h, lay.w))
for l in range(4):
    err = (lay.level_view(vol, l) - levels[l]).abs().max().item()
    assert err < 1e-5, (l, err)
recs = (ops.fmap_records(f1), ops.fmap_records(f2))
vol_r, _ = ops.corr_build_tiled(f1, f2, 4, recs=recs)
for l in range(4):
    err = (lay.level_view(vol_r, l) - levels[l]).abs().max().item()
    assert err < 1e-5, ("record build", l, err)
t_rec = timeit(lambda: ops.corr_build_tiled(f1, f2, 4, recs=recs))
t_cvt = timeit(lambda: (ops.fmap_records(f1), ops.fmap_records(f2)))
print(f"build on the record core {t_rec*1e6:8.1f} us (+ {t_cvt*1e6:.1f} us for the two feature-map conversions)")
t_old = timeit(lambda: ops.corr_build(f1, f2, 4))
t_new = timeit(lambda: ops.corr_build_tiled(f1, f2, 4))
bb = 4.0 * B * (2 * H * W * C + H * W * P)
print(f"build      row-major {t_old*1e6:8.1f} us   tiled {t_new*1e6:8.1f} us   ({bb/t_new/1e9:6.0f} GB/s algorithmic, {bb/t_new/8e12*100:4.1f} % of 8 TB/s)")

out_old = ops.corr_lookup_fwd(levels, coords[0], r, nhwc=True)
out_new = ops.corr_lookup_tiled_fwd(vol, lay, coords[0], r)
assert (out_old - out_new).abs().max().item() < 1e-5, (out_old - out_new).abs().max().item()
t_old = timeit(lambda: ops.corr_lookup_fwd(levels, coords[0], r, nhwc=True), 20)
t_new = timeit(lambda: ops.corr_lookup_tiled_fwd(vol, lay, coords[0], r), 20)
fb = 4.0 * nq * (400 + 2 + 324)
print(f"lookup fwd row-major {t_old*1e6:8.1f} us   tiled {t_new*1e6:8.1f} us   ({fb/t_new/1e9:6.0f} GB/s algorithmic, {fb/t_new/8e12*100:4.1f} %)")

douts = [torch.randn(B, H, W, 324, device=dev) for _ in range(T)]
dlv = [torch.zeros_like(l) for l in levels]
for c, g in zip(coords, douts):
    ops.corr_lookup_bwd_(dlv, c, g, r, nhwc=True)
dvol = ops.corr_dvol_build(douts, coords, lay, B, r)
for l in range(4):
    err = (lay.level_view(dvol, l) - dlv[l]).abs().max().item()
    assert err < 2e-4, (l, err)
z = dvol.clone()
for l in range(4):      # pad cells must be zero: remove the valid cells and look at what is left
    n = lay.th[l] * lay.tw[l] * 16
    t = z[:, lay.off[l]:lay.off[l] + n].view(nq, lay.th[l], lay.tw[l], 4, 4).permute(0, 1, 3, 2, 4).reshape(nq, lay.th[l] * 4, lay.tw[l] * 4)
    assert t[:, lay.h[l]:, :].abs().max().item() == 0 if lay.th[l] * 4 > lay.h[l] else True
    assert t[:, :, lay.w[l]:].abs().max().item() == 0 if lay.tw[l] * 4 > lay.w[l] else True


def old_bwd_lookups():
    for d in dlv:
        d.zero_()
    for c, g in zip(coords, douts):
        ops.corr_lookup_bwd_(dlv, c, g, r, nhwc=True)


t_old = timeit(old_bwd_lookups, 3)
t_new = timeit(lambda: ops.corr_dvol_build(douts, coords, lay, B, r), 5)
lb = 4.0 * nq * (T * (324 + 2 + 800) + P)
print(f"lookup bwd (x{T}) zero+RMW {t_old*1e6:8.1f} us   dvol_build {t_new*1e6:8.1f} us   ({lb/t_new/1e9:6.0f} GB/s algorithmic, {lb/t_new/8e12*100:4.1f} %)")

dvol_r = ops.corr_dvol_build(douts, coords, lay, B, r, records=True)
t_rec = timeit(lambda: ops.corr_dvol_build(douts, coords, lay, B, r, records=True), 5)
print(f"                dvol_build as records {t_rec*1e6:8.1f} us")
d1o, d2o = ops.corr_build_bwd(f1, f2, [d.clone() for d in dlv])
d1r, d2r = ops.corr_build_bwd_tiled(f1, f2, dvol_r, lay, records=True)
print("record path: dfmap1 rel err", ((d1o - d1r).norm() / d1o.norm()).item(), " dfmap2 rel err", ((d2o - d2r).norm() / d2o.norm()).item())
t_r = timeit(lambda: ops.corr_build_bwd_tiled(f1, f2, dvol_r, lay, records=True), 3)
print(f"                build bwd on the record GEMMs {t_r*1e6:8.1f} us")
d1n, d2n = ops.corr_build_bwd_tiled(f1, f2, dvol, lay)
print("dfmap1 rel err", ((d1o - d1n).norm() / d1o.norm()).item(), " dfmap2 rel err", ((d2o - d2n).norm() / d2o.norm()).item())
t_old = timeit(lambda: ops.corr_build_bwd(f1, f2, [d.clone() for d in dlv]), 3) - timeit(lambda: [d.clone() for d in dlv], 3)
t_new = timeit(lambda: ops.corr_build_bwd_tiled(f1, f2, dvol, lay), 3)
gb = 4.0 * B * (H * W * P + 4 * H * W * C)
print(f"build bwd  unpool+2 GEMMs {t_old*1e6:8.1f} us   f2cat+2 GEMMs+unpool(fmap) {t_new*1e6:8.1f} us   ({gb/t_new/1e9:6.0f} GB/s algorithmic)")
