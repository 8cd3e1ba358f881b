import os, sys
sys.path.insert(0, os.getcwd())
import torch
from flow_supervisor_amd import _lib, ops
lib = _lib.load()
torch.manual_seed(0)
B, H, W, r, T = 4, 55, 128, 4, 12
lay = ops.VolLayout.get(H, W, 4)
douts = [torch.randn(B, H, W, 324, device="cuda") for _ in range(T)]
base = torch.randn(B, 2, H, W, device="cuda") * 3
fl = [base + 0.3 * i for i in range(T)]
kt = ops.corr_bwd_ktiles(fl, lay, B, r, is_flow=True)
def timeit(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lib.fsraft_set_dvol_box(2)
for rnd in range(2):
  for pol in (4, 6, 7):
    lib.fsraft_set_dvol_policy(pol)
    t = timeit(lambda: ops.corr_dvol_build(douts, fl, lay, B, r, records=True, is_flow=True, wmask=kt.wmask))
    print("policy", pol, f"{t:.1f} us")
lib.fsraft_set_dvol_policy(0)
