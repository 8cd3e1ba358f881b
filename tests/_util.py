import json
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def shapes(name):
    return json.load(open(os.path.join(G, name + "_shapes.json")))


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, atol, rtol=2e-5, what=""):
    a = (a if isinstance(a, torch.Tensor) else T(a)).detach().float().cpu()
    b = (b if isinstance(b, torch.Tensor) else T(b)).detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    lim = atol + rtol * ref
    log = os.environ.get("FSRAFT_PARITY_LOG")         # margins of every comparison, for profiles/*parity_margins*
    if log:
        test = os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0]
        with open(log, "a") as f:
            f.write(f"{test}\t{what}\terr {err:.3e}\tlimit {lim:.3e} (atol {atol:g} + {rtol:g} * max|ref| {ref:.3g})\tused {err / lim if lim else 0:.3f}\n")
    assert err <= lim, f"{what}: max abs err {err:.3e} > {lim:.3e}"
    return err
