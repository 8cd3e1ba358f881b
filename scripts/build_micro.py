import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flow_supervisor_amd import ops, _lib
B, C, H, W = 4, 256, 55, 128
f1 = torch.randn(B, C, H, W, device="cuda"); f2 = torch.randn(B, C, H, W, device="cuda")
N = H * W; P = sum(h * w for h, w in ops.pyramid_sizes(H, W))
nbytes = 4.0 * B * (2 * N * C + N * P)
for mode in (0, 1):
    _lib.load().fsraft_set_build_split(mode)
    lv = ops.corr_build(f1, f2, 4); torch.cuda.synchronize(); del lv
    t0 = time.perf_counter()
    for _ in range(10): lv = ops.corr_build(f1, f2, 4); del lv
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"build split={mode}: {dt*1e6:7.1f} us  {nbytes/dt/1e9:7.0f} GB/s ({nbytes/dt/8e12*100:4.1f}% of 8 TB/s)  {2.0*B*N*N*C/dt/1e12:6.1f} TFLOP/s algorithmic")
