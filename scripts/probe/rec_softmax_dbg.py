import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flow_supervisor_amd import ops
torch.manual_seed(0)
B, N = 2, 128
logits = torch.randn(B, N, N, device="cuda") * 3
a = logits.clone(); ops.softmax_rows_(a)
r = logits.clone(); ops.softmax_rows_rec_(r)
d = ops.to_records(a)
ri, di = r.view(torch.int32), d.view(torch.int32)
neq = (ri != di)
print("differing words", int(neq.sum()), "of", ri.numel())
idx = neq.nonzero()[:10]
print(idx)
for b, i, j in idx.tolist():
    print(b, i, j, hex(ri[b, i, j].item() & 0xffffffff), hex(di[b, i, j].item() & 0xffffffff), "word-in-record", j % 32)
