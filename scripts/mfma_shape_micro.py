#!/usr/bin/env python3
"""A/B of the two bf16 MFMA shapes in the record GEMM (csrc/gemm_rec.hpp: rec_mainloop on v_mfma_f32_32x32x16_bf16 vs
rec_mainloop16 on v_mfma_f32_16x16x32_bf16): same LDS images, same DMA ring, same cycles per FLOP; the question is the
clock the chip holds (MI355X_MICROARCH.md, DVFS give-back item 7).  Interleaved rounds in one process, random operands."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)

for (b, M, N, K, ks) in ((1, 256, 128, 32, 1), (2, 300, 200, 96, 1), (1, 256, 128, 320, 1), (3, 70, 530, 1000, 1), (2, 257, 129, 640, 3), (1, 512, 256, 4096, 4)):
    A = torch.randn(b, M, K, device=dev)
    B = torch.randn(b, N, K, device=dev)
    ref = torch.bmm(A.double(), B.double().transpose(1, 2))
    outs = []
    for m16 in (0, 1):
        lib.fsraft_set_rec_mfma16(m16)
        got = ops.gemm_rec_nt(ops.to_records(A), ops.to_records(B), 0.5, ksplit=ks)
        err = (got.double() - 0.5 * ref).abs().max().item() / ref.abs().max().item()
        outs.append(got)
        assert err < 3e-5, (m16, err)
    print(f"b={b} M={M} N={N} K={K} ksplit={ks}: ok, max |16x16 - 32x32| = {(outs[0] - outs[1]).abs().max().item():.2e}")


def timeit(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for name, (b, M, N, K), ks, n in (("square 4096", (1, 4096, 4096, 4096), 1, 20), ("square 8192", (1, 8192, 8192, 8192), 1, 4),
                                  ("dF1 = f2cat . dV^T", (4, 256, 7040, 9600), 2, 20), ("conv-like 28160x256x1920", (1, 28160, 256, 1920), 1, 40),
                                  ("volume build (no epilogue)", (4, 7040, 7040, 256), 1, 10)):
    A = torch.randn(b, M, K, device=dev)
    B = torch.randn(b, N, K, device=dev)
    Ar, Br = ops.to_records(A), ops.to_records(B)
    out = torch.empty(b, M, N, device=dev)
    fl = 2.0 * b * M * N * K
    res = {0: [], 1: []}
    for m16 in (0, 1):
        lib.fsraft_set_rec_mfma16(m16)
        timeit(lambda: ops.gemm_rec_nt(Ar, Br, ksplit=ks, out=out), 3)
    for rnd in range(5):
        for m16 in (0, 1):
            lib.fsraft_set_rec_mfma16(m16)
            res[m16].append(timeit(lambda: ops.gemm_rec_nt(Ar, Br, ksplit=ks, out=out), n))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"{name:30s} 32x32x16: {med[0]*1e6:8.1f} us {fl/med[0]/1e12:6.1f} TF | 16x16x32: {med[1]*1e6:8.1f} us {fl/med[1]/1e12:6.1f} TF"
          f" | ratio {med[0]/med[1]:.3f}  (min {min(res[0])*1e6:.1f} / {min(res[1])*1e6:.1f} us)")
    del A, B, Ar, Br, out
lib.fsraft_set_rec_mfma16(0)
