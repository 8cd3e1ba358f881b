#!/usr/bin/env python3
"""AlternateCorrBlock lookup: the tile GEMM kernel on the matrix pipe (csrc/altcorr.hip altcorr_mfma_fwd_kernel, split arithmetic on
records) against the fp32 tile kernel (altcorr_tile_fwd_kernel) and against CorrBlock's volume lookup, for smooth flow,
flow with discontinuities (window positions outside a tile's region), rough flow, flow that leaves the image; several
shapes incl. ragged tiles; then timing at the KITTI shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402

dev = "cuda"
lib = _lib.load()
torch.manual_seed(0)


def setup(B, C, H, W, nlev=4):
    f1 = torch.randn(B, C, H, W, device=dev)
    f2 = torch.randn(B, C, H, W, device=dev)
    f1c = ops.nchw_to_nhwc(f1)
    lv, x = [], f2
    for _ in range(nlev):
        lv.append(ops.nchw_to_nhwc(x))
        x = F.avg_pool2d(x, 2, stride=2)
    recs = (ops.to_records(f1c.view(B, -1, C)), [ops.to_records(f.view(B, -1, C)) for f in lv])
    return f1, f2, f1c, lv, recs


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def flows(B, H, W):
    step = torch.zeros(B, 2, H, W, device=dev)
    step[:, 0, :, W // 2:] = 11.0                       # a motion boundary through the tiles of one column
    step[:, 1, H // 2:] -= 7.0
    return (("smooth (3.3, -1.7) + 0.3 px noise", torch.tensor([3.3, -1.7], device=dev).view(1, 2, 1, 1) + 0.3 * torch.randn(B, 2, H, W, device=dev)),
            ("motion boundaries (11 / -7 px steps)", step + 0.2 * torch.randn(B, 2, H, W, device=dev)),
            ("rough (8 px noise)", 8.0 * torch.randn(B, 2, H, W, device=dev)),
            ("leaving the image (+-200 px)", 200.0 * torch.randn(B, 2, H, W, device=dev)),
            ("zero", torch.zeros(B, 2, H, W, device=dev)))


for (B, C, H, W, nlev) in ((1, 256, 47, 156, 4), (2, 128, 17, 19, 4), (1, 64, 8, 12, 3), (2, 256, 46, 62, 4), (1, 96, 5, 7, 2)):
    f1, f2, f1c, lv, recs = setup(B, C, H, W, nlev)
    vol, lay = ops.corr_build_tiled(f1, f2, nlev, recs=(ops.fmap_records(f1), ops.fmap_records(f2))) if C % 32 == 0 else (None, None)
    for name, flow in flows(B, H, W):
        a = ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True)                    # fp32 tile kernel
        m = ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True, recs=recs)         # matrix pipe
        e = (a - m).abs().max().item() / max(a.abs().max().item(), 1e-6)
        line = f"B={B} C={C} {H}x{W} L={nlev}  {name:38s} |mfma - fp32 tile| / max = {e:.2e}"
        if vol is not None:
            v = ops.corr_lookup_tiled_fwd(vol, lay, flow, 4, True)
            ev = (v - m).abs().max().item() / max(v.abs().max().item(), 1e-6)
            line += f"   |mfma - volume lookup| / max = {ev:.2e}"
            assert ev < 3e-5, line
        print(line)
        assert e < 3e-5, line

B, C, H, W = 1, 256, 47, 156
f1, f2, f1c, lv, recs = setup(B, C, H, W)
for name, flow in flows(B, H, W):
    res = {0: [], 1: []}
    for rnd in range(3):
        res[0].append(timeit(lambda: ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True)))
        res[1].append(timeit(lambda: ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True, recs=recs)))
    print(f"47x156  {name:38s} fp32 tile kernel {sorted(res[0])[1]:7.1f} us   matrix pipe {sorted(res[1])[1]:7.1f} us")

# the per-launch dispatch (fsraft_altcorr_mfma_fwd's regime buffer): uncovered fraction u, the kernel chosen, the time of the dispatched
# call (statistic kernel + the chosen lookup + the other one's empty launch) beside both kernels alone, over a sweep of flow roughness
print("sigma of the flow noise [1/8-resolution cells] | uncovered queries u | chosen | fp32 tile / matrix pipe / dispatched  [us]")
reg = torch.zeros(8, dtype=torch.int32, device=dev)
for sigma in (0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 6.0, 8.0, 16.0, 200.0):
    flow = torch.tensor([3.3, -1.7], device=dev).view(1, 2, 1, 1) + sigma * torch.randn(B, 2, H, W, device=dev)
    t0 = sorted(timeit(lambda: ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True)) for _ in range(3))[1]
    t1 = sorted(timeit(lambda: ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True, recs=recs)) for _ in range(3))[1]
    t2 = sorted(timeit(lambda: ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True, recs=recs, regime=reg)) for _ in range(3))[1]
    r = reg.tolist()
    a = ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True)
    d = ops.altcorr_fused_fwd(f1c, lv, flow, 4, is_flow=True, recs=recs, regime=reg)
    e = (a - d).abs().max().item() / a.abs().max().item()
    assert e < 3e-5 and r[1] == 0 and r[2] == 0 and r[3] == 0, (e, r)
    worse = t2 / min(t0, t1)
    print(f"  {sigma:6.1f}   u = {r[4] / max(r[5], 1):5.3f} ({r[4]} of {r[5]})   {'fp32 tile' if r[0] else 'matrix pipe'}   {t0:7.1f} / {t1:7.1f} / {t2:7.1f}   dispatched / best = {worse:.2f}")
