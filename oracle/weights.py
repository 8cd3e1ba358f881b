"""Procedural weights and inputs shared by the golden generator and the tests
(TEST INFRASTRUCTURE).  Nothing is stored: both sides regenerate the same
tensors from a seed with numpy's PCG64 in a fixed, documented order
(keys sorted lexicographically), so fixtures hold outputs only.

Regime: update-block / GMA convs use PyTorch's default conv
init U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias; encoder convs use
N(0, 2/fan_out) weights (core/extractor.py:150-157) and default-init bias; norm
weight 1 / bias 0; BatchNorm running stats 0 / 1.
"""
import math

import numpy as np
import torch


def _fan(shape):
    rf = 1
    for s in shape[2:]:
        rf *= s
    return shape[1] * rf, shape[0] * rf


def procedural_state_dict(shapes, seed):
    """shapes: {key: tuple} (e.g. from a reference module's state_dict).
    Returns {key: float32 tensor} filled deterministically."""
    rng = np.random.default_rng(seed)
    out = {}
    for key in sorted(shapes):
        shp = tuple(shapes[key])
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "rel_ind":       # GMA RelPosEmb index buffer: keeps its constructed value (load with strict=False)
            continue
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros((), dtype=torch.long)
            continue
        if leaf == "running_mean":
            out[key] = torch.zeros(shp)
            continue
        if leaf == "running_var":
            out[key] = torch.ones(shp)
            continue
        is_encoder = key.startswith("fnet.") or key.startswith("cnet.")
        if len(shp) == 4:
            fan_in, fan_out = _fan(shp)
            if is_encoder:
                w = rng.standard_normal(shp) * math.sqrt(2.0 / fan_out)
            else:
                b = 1.0 / math.sqrt(fan_in)
                w = rng.uniform(-b, b, shp)
            out[key] = torch.from_numpy(w.astype(np.float32))
        elif leaf == "bias" and (key[: -len(".bias")] + ".weight") in shapes and len(shapes[key[: -len(".bias")] + ".weight"]) == 4:
            fan_in, _ = _fan(tuple(shapes[key[: -len(".bias")] + ".weight"]))
            b = 1.0 / math.sqrt(fan_in)
            out[key] = torch.from_numpy(rng.uniform(-b, b, shp).astype(np.float32))
        elif leaf == "weight":      # norm scale
            out[key] = torch.ones(shp)
        elif leaf == "bias":        # norm shift
            out[key] = torch.zeros(shp)
        else:                       # e.g. GMA gamma / rel-pos tables
            out[key] = torch.from_numpy((rng.standard_normal(shp) * 0.1).astype(np.float32))
    return out


def synthetic_pair(batch, ht, wd, seed):
    """image1 ~ U[0,255), image2 = roll(image1, (+3 rows, -5 cols)) + N(0, 2^2).
    (SURVEY.md 8d).  Low-pass structure is added so the correlation volume has
    peaks rather than white noise."""
    rng = np.random.default_rng(seed)
    base = rng.uniform(0, 255, (batch, 3, ht // 4 + 2, wd // 4 + 2)).astype(np.float32)
    t = torch.from_numpy(base)
    img1 = torch.nn.functional.interpolate(t, size=(ht, wd), mode="bilinear", align_corners=True)
    img1 = (0.7 * img1 + 0.3 * torch.from_numpy(rng.uniform(0, 255, (batch, 3, ht, wd)).astype(np.float32))).contiguous()
    noise = torch.from_numpy((rng.standard_normal((batch, 3, ht, wd)) * 2.0).astype(np.float32))
    img2 = (torch.roll(img1, shifts=(3, -5), dims=(2, 3)) + noise).clamp(0, 255).contiguous()
    return img1, img2


def rand_tensor(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def rand_uniform(shape, seed, lo, hi):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.uniform(lo, hi, shape).astype(np.float32))


RAFT_SHAPES_CACHE = {}
