import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import argparse, torch
from _util import shapes
from oracle.weights import procedural_state_dict, synthetic_pair
from flow_supervisor_amd.core.raft import RAFT
m = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False, dropout=0, corr_levels=4, corr_radius=4))
m.load_state_dict(procedural_state_dict(shapes("raft_basic"), 55))
m = m.cuda().eval()
im1, im2 = (t.cuda() for t in synthetic_pair(1, 128, 192, 56))
with torch.no_grad():
    a = m(im1, im2, iters=5, test_mode=True)
    b = m(im1, im2, iters=5, test_mode=True)
    p = m(im1, im2, iters=5)
    q = m(im1, im2, iters=5)
print("test_mode twice: low", (a[0]-b[0]).abs().max().item(), "up", (a[1]-b[1]).abs().max().item())
print("train preds twice:", [(x-y).abs().max().item() for x, y in zip(p, q)])
print("test vs train last:", (a[1]-p[-1]).abs().max().item())
for it in (1, 2):
    with torch.no_grad():
        a = m(im1, im2, iters=it, test_mode=True); p = m(im1, im2, iters=it)
    print(it, "iters: test vs train", (a[1]-p[-1]).abs().max().item())
