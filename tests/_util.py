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


def close(a, b, atol, rtol=1e-4, what=""):
    a = (a if isinstance(a, torch.Tensor) else T(a)).detach().float().cpu()
    b = (b if isinstance(b, torch.Tensor) else T(b)).detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    err = (a - b).abs().max().item()
    lim = atol + rtol * b.abs().max().item()
    assert err <= lim, f"{what}: max abs err {err:.3e} > {lim:.3e}"
    return err
