import os, sys, math, torch
sys.path.insert(0, os.getcwd())
import torch.nn.functional as F
from flow_supervisor_amd import ops, _lib
from flow_supervisor_amd.ops import Dst, V
_lib.load().fsraft_set_tuning(3, 1)
torch.manual_seed(0)
for (kh, kw, cin, cout) in [(1,1,324,256),(3,3,256,192),(1,5,384,256),(5,1,384,128),(3,3,128,512),(1,1,256,576)]:
    B,H,W = 2, 9, 13
    x = torch.randn(B, cin, H, W); w = torch.randn(cout, cin, kh, kw)/math.sqrt(cin*kh*kw); b = torch.randn(cout)
    y = F.conv2d(x.double(), w.double(), b.double(), padding=(kh//2, kw//2)).float()
    y32 = F.conv2d(x, w, b, padding=(kh//2, kw//2))
    split = [cin] if cin < 64 else [cin//2//4*4, cin - cin//2//4*4]
    xs, o = [], 0
    for c in split:
        xs.append(ops.nchw_to_nhwc(x[:, o:o+c].contiguous().cuda())); o += c
    srcs = [V(t, c) for t, c in zip(xs, split)]
    wpk = ops.pack_weight(w.cuda(), split, 0); wps = ops.pack_weight(w.cuda(), split, 10)
    out = torch.zeros(B, H, W, (cout+3)//4*4, device='cuda')
    ops.conv_forward(srcs, wpk, b.cuda(), B, H, W, kh, kw, cout, [Dst.nhwc(out)], wpk_split=wps)
    got = ops.nhwc_to_nchw(out, cout).cpu()
    print(f"{kh}x{kw} {cin}->{cout}: split-bf16 max err vs fp64 {float((got-y).abs().max()):.2e}   (torch fp32 vs fp64: {float((y32-y).abs().max()):.2e})  max|y| {float(y.abs().max()):.2f}")
