"""How far the image gradient of the encoder moves between the channels_last fsraft path and the all-MIOpen NCHW path, in
exact and split-bf16 arithmetic, next to the effect of a 1e-6 relative input perturbation (conditioning of the test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flow_supervisor_amd.core.extractor import BasicEncoder
from flow_supervisor_amd import _lib
DEV = "cuda"
for split in (1, 0):
    lib = _lib.load()
    lib.fsraft_set_tuning(3, 1 if split else 0); lib.fsraft_set_tuning(4, 2 if split else 0)
    torch.manual_seed(21)
    enc = BasicEncoder(output_dim=128, norm_fn="instance").to(DEV)
    x = torch.randn(2, 3, 72, 104, device=DEV)
    outs = {}
    for mode in ("0", "1"):
        os.environ["FSRAFT_ENCODER_CL"] = mode
        enc.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        a, b = enc([xi[:1], xi[1:]])
        (a.square().sum() + (b * 0.5).sum()).backward()
        outs[mode] = xi.grad.clone()
    # same NCHW path twice with a 1e-6 relative perturbation of the input: how much does dx move by itself?
    os.environ["FSRAFT_ENCODER_CL"] = "0"
    xi = (x * (1 + 1e-6 * torch.randn_like(x))).requires_grad_(True)
    a, b = enc([xi[:1], xi[1:]])
    (a.square().sum() + (b * 0.5).sum()).backward()
    ref = outs["0"].double(); d1 = (outs["1"].double() - ref); dp = (xi.grad.double() - ref)
    mx = ref.abs().max()
    for nm, d in (("cl-vs-nchw", d1), ("nchw perturbed 1e-6", dp)):
        big = d.abs() > 1e-3 * mx
        print(f"split={split} {nm}: relL2 {d.norm() / ref.norm():.3e}  frac>1e-3max {big.float().mean():.3e}  relL2 of the rest {(d * ~big).norm() / ref.norm():.3e}")
