#!/usr/bin/env python3
"""corr_dvol_kernel (the gradient volume of a step in one pass) at the benchmark shape: store cache policies A/B, plus the two
record GEMMs that read it right after (their time depends on what the stores left in the caches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)
B, C, H, W, r, T = 4, 256, 55, 128, 4, 12
lay = ops.VolLayout.get(H, W, 4)
douts = [torch.randn(B, H, W, 324, device=dev) for _ in range(T)]
flows = [torch.randn(B, 2, H, W, device=dev) * 3 for _ in range(T)]
f1 = torch.randn(B, C, H, W, device=dev)
f2 = torch.randn(B, C, H, W, device=dev)
f1r = ops.fmap_records(f1)


def timeit(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ref = None
cases = (("smooth: 12 lookups drifting 0.3 px per iteration", [f + 0.3 * i for i, f in enumerate([flows[0]] * T)]),
         ("independent 3 px noise per lookup", flows),
         ("jumping: +-40 px between lookups (work-list route)", [torch.randn(B, 2, H, W, device=dev) * 40 for _ in range(T)]))
for name, fl in cases:
    lib.fsraft_set_dvol_box(0)
    ref = ops.corr_dvol_build(douts, fl, lay, B, r, records=True, is_flow=True)
    ref32 = ops.corr_dvol_build(douts, fl, lay, B, r, records=False, is_flow=True)
    lib.fsraft_set_dvol_box(1)
    got = ops.corr_dvol_build(douts, fl, lay, B, r, records=True, is_flow=True)
    got32 = ops.corr_dvol_build(douts, fl, lay, B, r, records=False, is_flow=True)
    d = (got32 - ref32).abs()
    nz = int((ref32 != 0).sum()), int((got32 != 0).sum())
    print(f"   max |box - row| = {d.max().item():.3e} (max |ref| {ref32.abs().max().item():.3e}); nonzeros ref / box: {nz}; cells that differ: {int((d > 1e-6).sum())}")
    if d.max().item() > 1e-5:
        idx = (d > 1e-5).nonzero()[:6].tolist()
        for qq, pp in idx:
            lvl = max(l for l in range(4) if pp >= lay.off[l])
            rel = pp - lay.off[lvl]; tl = rel >> 4
            print("      q", qq, "p", pp, "level", lvl, "tile", tl, "cell", rel & 15, "ref", ref32[qq, pp].item(), "box", got32[qq, pp].item())
    assert d.max().item() <= 1e-5 * max(ref32.abs().max().item(), 1.0), name
    res = {0: [], 1: []}
    for rnd in range(3):
        for box in (0, 1):
            lib.fsraft_set_dvol_box(box)
            res[box].append(timeit(lambda: ops.corr_dvol_build(douts, fl, lay, B, r, records=True, is_flow=True), 5))
    print(f"{name:55s} row-segment kernel {sorted(res[0])[1]:7.1f} us   wave-per-query kernel + work list {sorted(res[1])[1]:7.1f} us")
# as in the train step: only the records the two list GEMMs read are written (fsraft_corr_bwd_ktiles' wmask)
for name, fl in cases[:2]:
    kt = ops.corr_bwd_ktiles(fl, lay, B, r, is_flow=True)
    res = {1: []}
    lib.fsraft_set_dvol_box(1)
    for rnd in range(3):
        res[1].append(timeit(lambda: ops.corr_dvol_build(douts, fl, lay, B, r, records=True, is_flow=True, wmask=kt.wmask), 5))
    frac = sum(bin(v & 0xffffffff).count("1") for v in kt.wmask.tolist()) / (B * -(-H * W // 32) * (lay.P // 32))
    print(f"{name:40s} records written {frac:.3f}: wave-per-query kernel {sorted(res[1])[1]:7.1f} us")
# chunked use (AlternateCorrBlock's backward): queries [q0, q0 + nq)
lib.fsraft_set_dvol_box(0)
ref32 = ops.corr_dvol_build(douts, flows, lay, B, r, records=False, is_flow=True)
ref = ops.corr_dvol_build(douts, flows, lay, B, r, records=True, is_flow=True, q0=5000, nq=2048)
lib.fsraft_set_dvol_box(1)
got = ops.corr_dvol_build(douts, flows, lay, B, r, records=True, is_flow=True, q0=5000, nq=2048)
print("chunk records differing words:", (got.view(torch.int32) != ref.view(torch.int32)).float().mean().item())
assert (ops.corr_dvol_build(douts, flows, lay, B, r, records=False, is_flow=True, q0=5000, nq=2048) - ref32[5000:5000 + 2048]).abs().max().item() < 1e-5, "chunk"
print("chunk of 2048 queries from 5000: identical")
