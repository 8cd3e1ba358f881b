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


def word_scale(word):
    """The power-of-two scale the kernels derive from an amax word (csrc/split_arith.hpp fs_scale_of_amax), restated."""
    import math
    a = float(word.item()) if isinstance(word, torch.Tensor) else float(word)
    if a == 0.0:
        return 1.0
    return 2.0 ** min(13 - math.floor(math.log2(a)), 62)


def decode_records(r, word):
    """Rows of [32 hi | 32 lo] fp16 records of x * scale(word) -> fp32 rows of x (hi + lo, un-scaled; exact in fp64)."""
    w = r.contiguous().view(torch.float16).view(r.shape[0], -1, 2, 32).double()
    return ((w[:, :, 0] + w[:, :, 1]) / word_scale(word)).reshape(r.shape[0], -1).float()


def _log_margin(what, err, lim, detail):
    log = os.environ.get("FSRAFT_PARITY_LOG")
    if log:
        test = os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0]
        with open(log, "a") as f:
            f.write(f"{test}\t{what}\terr {err:.3e}\tlimit {lim:.3e} ({detail})\tused {err / lim if lim else 0:.3f}\n")


def grad_digest_check(named_params, g, tol, prefix="gnorm.", hprefix="ghead.", skip=(), atol_norm=1e-5):
    """Every parameter-gradient norm (and, where the fixture has it, the first 32 elements) against a reference-generated
    digest.  tol: dict(gnorm, ghead, gnorm_fnet, ghead_fnet) -- the feature encoder's gradients pass through InstanceNorm
    and the volume backward and carry ~1e-3 of summation-order noise in the REFERENCE itself (measured in round 2, docs/history),
    so `fnet.*` has its own, wider pair of limits; everything else (update blocks, context encoder, GMA attention) is held
    to the tighter pair.  norm: |ours - ref| <= rtol * ref + atol_norm (atol: biases in front of InstanceNorm have a zero
    gradient in exact arithmetic and ~1e-6 of rounding noise in the reference); head: max abs error <= rtol_head *
    max|head| + the share of rtol * ref one element carries.  The worst margin of each group goes to the parity log.
    Returns the offenders."""
    import math
    bad = []
    worst = {}
    for k, p in named_params:
        if any(t in k for t in skip) or prefix + k not in g:
            continue
        grp = "fnet" if k.startswith("fnet.") else "rest"
        rn, rh = (tol["gnorm_fnet"], tol["ghead_fnet"]) if grp == "fnet" else (tol["gnorm"], tol["ghead"])
        ref = float(g[prefix + k])
        gn = 0.0 if p.grad is None else p.grad.norm().item()
        lim = rn * max(ref, 1e-6) + atol_norm
        err = abs(gn - ref)
        if err / lim >= worst.get(("n", grp), (-1.0,))[0]:
            worst[("n", grp)] = (err / lim, rn, f"{k}: {gn:.6g} vs {ref:.6g}")
        if not err <= lim:
            bad.append((k, "norm", gn, ref))
        if hprefix and hprefix + k in g and p.grad is not None and ref > 1e-4:
            head = T(g[hprefix + k]).float()
            herr = (p.grad.reshape(-1)[:32].cpu() - head).abs().max().item()
            hlim = rh * head.abs().max().item() + rn * ref / math.sqrt(p.numel()) + 1e-7
            if herr / hlim >= worst.get(("h", grp), (-1.0,))[0]:
                worst[("h", grp)] = (herr / hlim, rh, k)
            if not herr <= hlim:
                bad.append((k, "head", herr, head.abs().max().item()))
    for (kind, grp), (used, r, what) in sorted(worst.items()):
        _log_margin(f"{prefix if kind == 'n' else hprefix}* worst of {grp} ({what})", used * r, r,
                    "relative to the reference norm" if kind == "n" else "relative to max|head|")
    return bad


def rel_check(value, ref, rtol, what):
    """|value - ref| <= rtol * |ref| for scalars (losses), logged like close()."""
    err, lim = abs(float(value) - float(ref)), rtol * abs(float(ref))
    _log_margin(what, err, lim, f"{rtol:g} * |ref| {abs(float(ref)):.6g}")
    assert err <= lim, f"{what}: {float(value)!r} vs reference {float(ref)!r} (rel {err / max(abs(float(ref)), 1e-30):.2e} > {rtol:g})"
