#!/usr/bin/env python3
"""A/B of the two record-operand volume builds (csrc/corr_build.hip): corr_build_rec_kernel (queries on M, tile parked in
LDS, ten barriers in the epilogue) vs corr_build_rec_t_kernel (targets on M, every level stored straight from the
accumulators).  Correctness cell by cell over the existing cells of every level (against the other kernel and against fp64), then interleaved
timing at the benchmark shape and two ragged ones."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
if lib.fsraft_set_tuning(24, 0) != 0 and len(sys.argv) > 1:
    sys.exit("the transposed-role / persistent build kernels live in the experiment build only: make -C flow_supervisor_amd/csrc ablate; "
             "FSRAFT_LIB_PATH=flow_supervisor_amd/csrc/build/ablate/libfsraft_ablate.so python scripts/build_t_micro.py <kernel>")
KERNEL = int(sys.argv[1]) if len(sys.argv) > 1 else 0        # 1: one tile per workgroup, stores from the accumulators; 2 / 3 / 4: persistent (DEFER 1 / 0 / 2)
torch.manual_seed(0)


def build(f1, f2, which, nlev=4):
    lib.fsraft_set_build_kernel(which)
    recs = (ops.fmap_records(f1), ops.fmap_records(f2))
    return ops.corr_build_tiled(f1, f2, nlev, recs=recs)


for (B, C, H, W, nlev) in ((1, 256, 8, 32, 4), (2, 128, 17, 19, 4), (1, 256, 46, 62, 4), (2, 256, 55, 128, 4), (1, 128, 16, 32, 3), (1, 256, 9, 70, 2), (1, 256, 47, 156, 4), (3, 256, 30, 40, 1)):
    f1 = torch.randn(B, C, H, W, device=dev)
    f2 = torch.randn(B, C, H, W, device=dev)
    v0, lay = build(f1, f2, 0, nlev)
    v1, _ = build(f1, f2, KERNEL, nlev)
    N = H * W
    ref = torch.bmm(f1.reshape(B, C, N).transpose(1, 2).double(), f2.reshape(B, C, N).double()).float() / C ** 0.5   # [B, N, N]
    worst = 0.0
    pyr = [ref.reshape(-1, 1, H, W)]
    for l in range(1, nlev):
        pyr.append(torch.nn.functional.avg_pool2d(pyr[-1], 2, stride=2))
    for l in range(nlev):
        h, w, th, tw, off = lay.h[l], lay.w[l], lay.th[l], lay.tw[l], lay.off[l]
        a = v0[:, off:off + th * tw * 16].reshape(-1, th, tw, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, th * 4, tw * 4)
        c = v1[:, off:off + th * tw * 16].reshape(-1, th, tw, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, th * 4, tw * 4)
        d = (a[:, :h, :w] - c[:, :h, :w]).abs().max().item()
        worst = max(worst, d)
        e = (c[:, :h, :w] - pyr[l][:, 0]).abs().max().item()
        e0 = (a[:, :h, :w] - pyr[l][:, 0]).abs().max().item()
        if e > 2e-4 or e0 > 2e-4:
            print(f"   level {l}: new vs fp64 {e:.2e}, old vs fp64 {e0:.2e}")
            bad = ((c[:, :h, :w] - pyr[l][:, 0]).abs() > 2e-4).nonzero()
            print("   first bad (query, y, x):", bad[:8].tolist(), "count", len(bad))
        assert e < 2e-4, (f"level {l} vs fp64", e)
        # (pad cells of the forward volume are never read -- corr_layout.hpp -- and cells of tiles no patch reaches are
        #  not even written; only existing cells are compared)
    print(f"B={B} C={C} {H}x{W} levels={nlev}: max |old - new| over existing cells {worst:.2e}")
    assert worst < 2e-6


def timeit(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for (B, C, H, W) in ((4, 256, 55, 128), (8, 256, 46, 62), (1, 256, 47, 156), (1, 256, 54, 128)):
    f1 = torch.randn(B, C, H, W, device=dev)
    f2 = torch.randn(B, C, H, W, device=dev)
    recs = (ops.fmap_records(f1), ops.fmap_records(f2))
    lay = ops.VolLayout.get(H, W, 4)
    vol = torch.empty(B * H * W, lay.P, device=dev)
    N = H * W
    nbytes = 4.0 * B * (2 * N * C + N * sum(h * w for h, w in zip(lay.h, lay.w)))

    def run():
        _lib.check(lib.fsraft_corr_build_rec(_lib.ptr(recs[0]), _lib.ptr(recs[1]), _lib.ptr(vol), 4, B, C, H, W, _lib.stream()), "build")
    # (kernel, store policy << 8: 1 plain, 2 sc1, 3 nt; 0 = the library's choice)
    variants = [(0, 256), (0, 512), (0, 768), (0, 0)] + ([(1, 256), (2, 256), (3, 256), (3, 768)] if len(sys.argv) > 1 else [])       # (kernel, store policy << 8: 0 plain, 1 sc1, 2 nt)
    res = {v: [] for v in variants}
    for v in variants:
        lib.fsraft_set_build_kernel(v[0] | v[1] << 8); timeit(run, 3)
    for rnd in range(5):
        for v in variants:
            lib.fsraft_set_build_kernel(v[0] | v[1] << 8)
            res[v].append(timeit(run, 10))
    line = f"B={B} {H}x{W}:"
    for v in variants:
        med = sorted(res[v])[2]
        line += f"  k{v[0]}/{('auto', 'plain', 'sc1', 'nt')[v[1] >> 8]} {med*1e6:6.1f} us ({nbytes/med/8e12*100:4.1f}%)"
    print(line)
lib.fsraft_set_build_kernel(1)
