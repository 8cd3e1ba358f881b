#!/usr/bin/env python3
"""Record-operand GEMM (csrc/gemm_rec.hip): correctness against fp64 on ragged shapes, then timing at the shapes of the
correlation path next to the round-1 split GEMM (fp32 operands converted while staging)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flow_supervisor_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)


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


for (b, M, N, K, ks) in ((1, 256, 128, 32, 1), (2, 300, 200, 96, 1), (1, 256, 128, 320, 1), (3, 70, 530, 1000, 1), (2, 257, 129, 640, 3), (1, 512, 256, 4096, 4)):
    A = torch.randn(b, M, K, device=dev)
    B = torch.randn(b, N, K, device=dev)
    ref = torch.bmm(A.double(), B.double().transpose(1, 2))
    got = ops.gemm_rec_nt(ops.to_records(A), ops.to_records(B), 0.5, ksplit=ks)
    err = (got.double() - 0.5 * ref).abs().max().item() / ref.abs().max().item()
    old = ops.gemm(A, B, True, 0.5)
    err_old = (old.double() - 0.5 * ref).abs().max().item() / ref.abs().max().item()
    print(f"b={b} M={M} N={N} K={K} ksplit={ks}: rel err {err:.2e} (round-1 split core {err_old:.2e})")
    assert err < 3e-5, err

print("-- k-major (TN)")
for (b, K, M, N, ks) in ((1, 32, 256, 128, 1), (2, 100, 300, 200, 1), (1, 77, 64, 40, 1), (3, 1000, 530, 70, 2), (1, 4096, 512, 256, 4)):
    A = torch.randn(b, K, M, device=dev)
    B = torch.randn(b, K, N, device=dev)
    ref = torch.bmm(A.double().transpose(1, 2), B.double())
    got = ops.gemm_rec_tn(ops.to_records(A), ops.to_records(B), M, N, 0.5, ksplit=ks)
    err = (got.double() - 0.5 * ref).abs().max().item() / ref.abs().max().item()
    print(f"b={b} K={K} M={M} N={N} ksplit={ks}: rel err {err:.2e}")
    assert err < 3e-5, err
for name, (b, K, M, N), splits in (("dF2cat = dV^T . f1^T", (4, 7040, 9600, 256), (1, 2, 3)), ("wgrad-like K=338k 256x1280", (1, 337920, 256, 1280), (32, 64))):
    A = torch.randn(b, K, M, device=dev)
    B = torch.randn(b, K, N, device=dev)
    Ar, Br = ops.to_records(A), ops.to_records(B)
    fl = 2.0 * b * M * N * K
    t_old = timeit(lambda: ops.gemm_tn_split(A, B), 3)
    line = f"{name:32s} round-1 split tn {t_old*1e6:8.1f} us {fl/t_old/1e12:6.1f} TF |"
    for ks in splits:
        t = timeit(lambda: ops.gemm_rec_tn(Ar, Br, M, N, ksplit=ks), 5)
        line += f"  rec ks={ks} {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF ({fl/t/1e12/833.3*100:4.1f} %)"
    print(line)
    del A, B, Ar, Br
print("-- timing")
for name, (b, M, N, K), splits in (("dF1 = f2cat . dV^T", (4, 256, 7040, 9600), (1, 2, 3, 4)), ("volume build (no epilogue)", (4, 7040, 7040, 256), (1,)),
                                    ("square 4096", (1, 4096, 4096, 4096), (1,)), ("conv-like 28160x256x1920", (1, 28160, 256, 1920), (1,))):
    A = torch.randn(b, M, K, device=dev)
    B = torch.randn(b, N, K, device=dev)
    Ar, Br = ops.to_records(A), ops.to_records(B)
    fl = 2.0 * b * M * N * K
    t_old = timeit(lambda: ops.gemm(A, B, True), 5)
    line = f"{name:32s} round-1 split {t_old*1e6:8.1f} us {fl/t_old/1e12:6.1f} TF |"
    for ks in splits:
        t = timeit(lambda: ops.gemm_rec_nt(Ar, Br, ksplit=ks), 5)
        line += f"  rec ks={ks} {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF ({fl/t/1e12/833.3*100:4.1f} % of 2.5 PF / 3)"
    print(line)
    del A, B, Ar, Br
