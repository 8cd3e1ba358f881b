#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (container only).

Usage (build container, where /root/reference exists):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's PyTorch tree (/root/reference/pytorch/core) is imported, fed
procedurally generated weights and inputs (oracle/weights.py, seeds below) and
its OUTPUTS are stored.  No reference source is copied; the fixtures are data.
Each fixture records the seeds/shapes needed to regenerate the inputs.
"""
import argparse
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/pytorch")
warnings.filterwarnings("ignore")

from core.corr import CorrBlock                      # noqa: E402  (reference)
from core.raft import RAFT                           # noqa: E402
from core.update import BasicUpdateBlock, SmallUpdateBlock  # noqa: E402
from core.utils.utils import InputPadder, coords_grid, upflow8  # noqa: E402

from oracle.weights import procedural_state_dict, rand_tensor, rand_uniform, synthetic_pair  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def args_ns(small):
    return argparse.Namespace(small=small, mixed_precision=False, alternate_corr=False, dropout=0,
                              corr_levels=4, corr_radius=3 if small else 4)


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


# ---------------------------------------------------------------- G1 / G2
CORR_CASES = [
    # name, B, C, H, W, radius, seed
    ("corr_tiny", 1, 8, 16, 16, 3, 101),
    ("corr_odd", 1, 32, 17, 19, 4, 102),
    ("corr_mid", 2, 256, 16, 24, 4, 103),
]


def gen_corr():
    for name, B, C, H, W, r, seed in CORR_CASES:
        f1 = rand_tensor((B, C, H, W), seed, 1.0).requires_grad_(True)
        f2 = rand_tensor((B, C, H, W), seed + 1, 1.0).requires_grad_(True)
        blk = CorrBlock(f1, f2, num_levels=4, radius=r)
        coords = coords_grid(B, H, W) + rand_uniform((B, 2, H, W), seed + 2, -6.0, 6.0)
        # make a few queries land far outside / exactly on integers
        coords[:, :, 0, 0] = -20.0
        coords[:, :, 1, 1] = torch.tensor([float(W + 7), float(H + 9)]).view(1, 2)
        coords[:, :, 2, 2] = torch.tensor([3.0, 2.0]).view(1, 2)
        out = blk(coords)
        g = rand_tensor(tuple(out.shape), seed + 3, 1.0)
        (out * g).sum().backward()
        d = {f"pyr{l}": p for l, p in enumerate(blk.corr_pyramid)}
        save(name, B=B, C=C, H=H, W=W, radius=r, seed=seed, coords=coords, out=out,
             dfmap1=f1.grad, dfmap2=f2.grad, **d)


# ---------------------------------------------------------------- G3
def gen_update():
    for small in (False, True):
        a = args_ns(small)
        blk = SmallUpdateBlock(a, hidden_dim=96) if small else BasicUpdateBlock(a, hidden_dim=128)
        shapes = shapes_of(blk)
        tag = "small" if small else "basic"
        with open(os.path.join(HERE, f"update_{tag}_shapes.json"), "w") as f:
            json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0)
        seed = 300 + int(small)
        blk.load_state_dict(procedural_state_dict(shapes, seed))
        B, H, W = 1, 12, 16
        hd, cd = (96, 64) if small else (128, 128)
        cp = 4 * (2 * a.corr_radius + 1) ** 2
        net = torch.tanh(rand_tensor((B, hd, H, W), seed + 10)).requires_grad_(True)
        inp = torch.relu(rand_tensor((B, cd, H, W), seed + 11)).requires_grad_(True)
        corr = rand_tensor((B, cp, H, W), seed + 12, 2.0).requires_grad_(True)
        flow = rand_tensor((B, 2, H, W), seed + 13, 3.0).requires_grad_(True)
        net2, mask, delta = blk(net, inp, corr, flow)
        loss = (net2 * rand_tensor(tuple(net2.shape), seed + 20)).sum() + (delta * rand_tensor(tuple(delta.shape), seed + 21)).sum()
        if mask is not None:
            loss = loss + (mask * rand_tensor(tuple(mask.shape), seed + 22)).sum()
        loss.backward()
        d = dict(B=B, H=H, W=W, seed=seed, net_out=net2, delta=delta,
                 dnet=net.grad, dinp=inp.grad, dcorr=corr.grad, dflow=flow.grad)
        if mask is not None:
            d["mask"] = mask
        for k, p in blk.named_parameters():
            # big tensors: L2 norm + a 4096-element strided sample (stride = numel // 4096)
            g = p.grad.reshape(-1)
            d["dparam_norm." + k] = g.norm()
            d["dparam." + k] = g if g.numel() <= 4096 else g[:: g.numel() // 4096][:4096].clone()
        save(f"update_{tag}", **d)


# ---------------------------------------------------------------- G4 / G6
def gen_upsample():
    model = RAFT(args_ns(False))
    N, H, W = 2, 6, 8
    flow = rand_tensor((N, 2, H, W), 401, 2.0).requires_grad_(True)
    mask = rand_tensor((N, 576, H, W), 402, 1.5).requires_grad_(True)
    up = model.upsample_flow(flow, mask)
    g = rand_tensor(tuple(up.shape), 403)
    (up * g).sum().backward()
    save("upsample", N=N, H=H, W=W, up=up, dflow=flow.grad, dmask=mask.grad)

    f = rand_tensor((2, 2, 5, 7), 411, 2.0)
    pads = {}
    for mode in ("sintel", "kitti"):
        for (h, w) in ((436, 1024), (375, 1242), (368, 496), (128, 256), (370, 1226)):
            pads[f"pad_{mode}_{h}_{w}"] = np.array(InputPadder((1, 3, h, w), mode=mode)._pad)
    save("helpers", upflow8=upflow8(f), coords_grid=coords_grid(2, 3, 5), **pads)


# ---------------------------------------------------------------- G5
E2E_CASES = [
    # name, small, B, H, W, iters, seed, stride for flow_up
    ("e2e_small_128x256", True, 1, 128, 256, 4, 501, 1),
    ("e2e_basic_368x496", False, 1, 368, 496, 12, 502, 4),
    ("e2e_basic_440x1024", False, 1, 440, 1024, 12, 503, 4),
]


def gen_e2e():
    for name, small, B, H, W, iters, seed, stride in E2E_CASES:
        model = RAFT(args_ns(small))
        shapes = shapes_of(model)
        tag = "small" if small else "basic"
        with open(os.path.join(HERE, f"raft_{tag}_shapes.json"), "w") as f:
            json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0)
        model.load_state_dict(procedural_state_dict(shapes, seed))
        model.eval()
        im1, im2 = synthetic_pair(B, H, W, seed + 1)
        with torch.no_grad():
            flow_low, flow_up = model(im1, im2, iters=iters, test_mode=True)
        save(name, small=small, B=B, H=H, W=W, iters=iters, seed=seed, stride=stride,
             flow_low=flow_low, flow_up_strided=flow_up[:, :, ::stride, ::stride].contiguous(),
             flow_up_absmean=flow_up.abs().mean())


def gen_kitti():
    """KITTI-shaped evaluation exactly as evaluate.py:133-148 runs it: pad (mode='kitti') -> 24 iters -> unpad."""
    name, B, H, W, iters, seed, stride = "e2e_basic_kitti_375x1242", 1, 375, 1242, 24, 504, 6
    model = RAFT(args_ns(False))
    model.load_state_dict(procedural_state_dict(shapes_of(model), seed))
    model.eval()
    im1, im2 = synthetic_pair(B, H, W, seed + 1)
    padder = InputPadder(im1.shape, mode="kitti")
    p1, p2 = padder.pad(im1, im2)
    with torch.no_grad():
        flow_low, flow_up = model(p1, p2, iters=iters, test_mode=True)
    flow = padder.unpad(flow_up)
    save(name, small=False, B=B, H=H, W=W, iters=iters, seed=seed, stride=stride, padded=np.array(p1.shape[-2:]),
         flow_low=flow_low, flow_strided=flow[:, :, ::stride, ::stride].contiguous(), flow_absmean=flow.abs().mean())


def gen_l2l():
    """Flow-supervisor two-phase forward (core/l2l.py): crop = window of the uncropped pair, fwd+bwd, frozen BN."""
    from core.l2l import L2L
    seed, H, W, h, w, oy, ox, iters, B = 701, 160, 256, 128, 192, 16, 40, 6, 2
    model = L2L(args_ns(False))
    shapes = shapes_of(model)
    with open(os.path.join(HERE, "l2l_basic_shapes.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0)
    model.load_state_dict(procedural_state_dict(shapes, seed))
    model.train()
    model.freeze_bn()
    ci1, ci2 = synthetic_pair(B, H, W, seed + 1)
    im1 = ci1[:, :, oy:oy + h, ox:ox + w].contiguous()
    im2 = ci2[:, :, oy:oy + h, ox:ox + w].contiguous()
    preds = model(im1, im2, ci1, ci2, torch.tensor([ox] * B), torch.tensor([oy] * B), iters=iters)
    n = len(preds)
    loss = 0.0
    for i, p in enumerate(preds):
        loss = loss + (0.8 ** (n - i - 1)) * torch.sqrt(p * p + 1e-6).mean()
    loss.backward()
    d = dict(seed=seed, H=H, W=W, h=h, w=w, oy=oy, ox=ox, iters=iters, B=B, loss=loss.detach(),
             mid=preds[iters // 2 - 1].detach()[:, :, ::2, ::2], last=preds[-1].detach()[:, :, ::2, ::2])
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        d["gnorm." + k] = g.norm()
    model.eval()
    with torch.no_grad():
        low, up = model(im1, im2, iters=iters, test_mode=True)
    d["test_low"], d["test_up"] = low, up[:, :, ::2, ::2]
    save("l2l_basic", **d)


def _ref_train_fn(name):
    """A pure-torch function of pytorch/train.py (the module imports cv2 / tensorflow and cannot be imported here): its
    FunctionDef is taken out of the syntax tree and executed with only `torch` and MAX_FLOW in scope."""
    import ast
    tree = ast.parse(open("/root/reference/pytorch/train.py").read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    scope = {"torch": torch, "MAX_FLOW": 400}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "train.py:" + name, "exec"), scope)
    return scope[name]


def l2l_recipe_inputs(seed, H=432, W=1024, h=368, w=768):
    """One labelled and one unlabelled sample of the flow-supervisor step at the reference recipe (train_semi.sh:3-6:
    --batch_size 1 --image_size 368 768): the uncropped frame is the Sintel frame floored to a multiple of 8
    (436x1024 -> 432x1024, raft_utils/augmentor.py:561-565 get_proc_size_floor), the crop sits at offsets that are
    multiples of 8 (augmentor.py:621-622) and carries photometric noise (the colour augmentation's stand-in)."""
    out = {}
    for tag, sd, oy, ox in (("sup", seed + 1, 40, 136), ("unsup", seed + 5, 16, 200)):
        ci1, ci2 = synthetic_pair(1, H, W, sd)
        im1 = (ci1[:, :, oy:oy + h, ox:ox + w] + rand_tensor((1, 3, h, w), sd + 1, 3.0)).clamp(0, 255).contiguous()
        im2 = (ci2[:, :, oy:oy + h, ox:ox + w] + rand_tensor((1, 3, h, w), sd + 2, 3.0)).clamp(0, 255).contiguous()
        flow = rand_tensor((1, 2, h, w), sd + 3, 4.0)
        valid = (rand_uniform((1, h, w), sd + 4, 0.0, 1.0) > 0.1).float()
        out[tag] = (im1, im2, ci1, ci2, torch.tensor([ox]), torch.tensor([oy]), flow, valid)
    return out


def gen_l2l_recipe(only=None):
    """The flow-supervisor optimisation step itself at the reference recipe's size: L2L / GMAL2L,
    B = 1, crop 368x768 inside the 432x1024 frame, iters = 12 + 12, labelled pass with sequence_loss + backward, unlabelled
    pass with sequence_loss_unsup + backward (pytorch/train.py:270-277), gradients of both passes accumulated.  Stored:
    both losses, strided student / supervisor predictions, every parameter's gradient norm and head after the labelled
    pass alone and after both."""
    from core.gma_l2l import GMAL2L
    from core.l2l import L2L
    seq, seq_u = _ref_train_fn("sequence_loss"), _ref_train_fn("sequence_loss_unsup")
    # (third recipe, VERDICT r3 next #4: the KITTI semi-supervised stage, train_semi.sh:14-17 -- crop 288x960; the uncropped frame is
    #  the 375x1242 KITTI frame floored to a multiple of 8, 368x1240)
    for name, cls, ns, seed, gamma, lam, (H, W, h, w) in (("l2l_recipe_basic", L2L, args_ns(False), 711, 0.8, 1.0, (432, 1024, 368, 768)),
                                                           ("l2l_recipe_gma", GMAL2L, gma_ns(), 712, 0.85, 0.25, (432, 1024, 368, 768)),
                                                           ("l2l_recipe_kitti", L2L, args_ns(False), 713, 0.8, 1.0, (368, 1240, 288, 960))):
        if only is not None and name not in only:
            continue
        model = cls(ns)
        shapes = shapes_of(model)
        with open(os.path.join(HERE, name + "_shapes.json"), "w") as f:
            json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0)
        model.load_state_dict(procedural_state_dict(shapes, seed), strict=False)
        if cls is GMAL2L:
            with torch.no_grad():
                model.update_block.aggregator.gamma.fill_(0.1)
        model.train()
        model.freeze_bn()
        inp = l2l_recipe_inputs(seed, H, W, h, w)
        d = dict(seed=seed, H=H, W=W, h=h, w=w, iters=24, B=1, gamma=gamma, unsup_lambda=lam, stride=4)
        for tag in ("sup", "unsup"):
            im1, im2, ci1, ci2, ox, oy, flow, valid = inp[tag]
            preds = model(im1, im2, ci1, ci2, ox, oy, iters=24)
            assert len(preds) == 24 and tuple(preds[-1].shape) == (1, 2, h, w)
            if tag == "sup":
                loss, metrics = seq(preds, flow, valid, gamma)
            else:
                loss, metrics = seq_u(preds, flow, valid, unsup_weight=lam)     # (train.py:276: gamma stays at its default)
            loss.backward()
            d[tag + "_loss"] = loss.detach()
            d[tag + "_epe"] = np.float64(metrics["epe"])
            d[tag + "_ox"], d[tag + "_oy"] = int(ox[0]), int(oy[0])
            for i in (0, 11, 12, 23):
                d[f"{tag}_pred{i}"] = preds[i].detach()[:, :, ::4, ::4].contiguous()
                d[f"{tag}_pred{i}_absmean"] = preds[i].detach().abs().mean()
            for k, p in model.named_parameters():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                sfx = "" if tag == "unsup" else "_sup"
                d[f"gnorm{sfx}." + k] = g.norm()
                d[f"ghead{sfx}." + k] = g.reshape(-1)[:32].clone()
            del preds, loss
        save(name, **d)
        del model


# ---------------------------------------------------------------- G7 (GMA, benchmark config 5)
def gma_ns():
    return argparse.Namespace(small=False, mixed_precision=False, dropout=0, num_heads=1, position_only=False,
                              position_and_content=False, corr_levels=4, corr_radius=4)


def _digest(d, named_params):
    for k, p in named_params:
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
        d["dparam_norm." + k] = g.norm()
        d["dparam." + k] = g if g.numel() <= 4096 else g[:: g.numel() // 4096][:4096].clone()


def gen_gma():
    from core.gma import Aggregate, Attention
    from core.gma_network import RAFTGMA
    from core.gma_update import GMAUpdateBlock
    a = gma_ns()
    # --- Attention / Aggregate ops
    B, H, W, seed = 2, 12, 16, 800
    att = Attention(args=a, dim=128, heads=1, max_pos_size=160, dim_head=128)
    agg = Aggregate(args=a, dim=128, dim_head=128, heads=1)
    sh = {"att." + k: tuple(v.shape) for k, v in att.state_dict().items()}
    sh.update({"agg." + k: tuple(v.shape) for k, v in agg.state_dict().items()})
    sd = procedural_state_dict(sh, seed)
    att.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("att.")}, strict=False)
    agg.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("agg.")})
    ctx = torch.relu(rand_tensor((B, 128, H, W), seed + 1, 1.5)).requires_grad_(True)
    fm = rand_tensor((B, 128, H, W), seed + 2).requires_grad_(True)
    A = att(ctx)
    out = agg(A, fm)
    (out * rand_tensor(tuple(out.shape), seed + 3)).sum().backward()
    d = dict(B=B, H=H, W=W, seed=seed, attn=A, out=out, dctx=ctx.grad, dfm=fm.grad)
    _digest(d, [("att." + k, p) for k, p in att.named_parameters() if "pos_emb" not in k] +
               [("agg." + k, p) for k, p in agg.named_parameters()])
    with open(os.path.join(HERE, "gma_ops_shapes.json"), "w") as f:
        json.dump({k: list(v) for k, v in sh.items()}, f, indent=0)
    save("gma_ops", **d)

    # --- GMAUpdateBlock fwd + grads
    blk = GMAUpdateBlock(a, hidden_dim=128)
    shapes = shapes_of(blk)
    with open(os.path.join(HERE, "update_gma_shapes.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0)
    seed = 810
    blk.load_state_dict(procedural_state_dict(shapes, seed))
    B, H, W = 1, 12, 16
    net = torch.tanh(rand_tensor((B, 128, H, W), seed + 10)).requires_grad_(True)
    inp = torch.relu(rand_tensor((B, 128, H, W), seed + 11)).requires_grad_(True)
    corr = rand_tensor((B, 324, H, W), seed + 12, 2.0).requires_grad_(True)
    flow = rand_tensor((B, 2, H, W), seed + 13, 3.0).requires_grad_(True)
    attn = torch.softmax(rand_tensor((B, 1, H * W, H * W), seed + 14, 2.0), -1).requires_grad_(True)
    net2, mask, delta = blk(net, inp, corr, flow, attn)
    loss = ((net2 * rand_tensor(tuple(net2.shape), seed + 20)).sum() + (delta * rand_tensor(tuple(delta.shape), seed + 21)).sum()
            + (mask * rand_tensor(tuple(mask.shape), seed + 22)).sum())
    loss.backward()
    d = dict(B=B, H=H, W=W, seed=seed, net_out=net2, delta=delta, mask=mask, dnet=net.grad, dinp=inp.grad,
             dcorr=corr.grad, dflow=flow.grad, dattn=attn.grad[:, :, ::3, ::3].contiguous(), dattn_norm=attn.grad.norm())
    _digest(d, blk.named_parameters())
    save("update_gma", **d)

    # --- end to end (eval) and one training step
    model = RAFTGMA(gma_ns())
    shapes = shapes_of(model)
    with open(os.path.join(HERE, "raft_gma_shapes.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items() if not k.endswith("rel_ind")}, f, indent=0)
    seed, H, W, iters, stride = 820, 368, 496, 12, 4
    model.load_state_dict(procedural_state_dict(shapes, seed), strict=False)
    model.eval()
    im1, im2 = synthetic_pair(1, H, W, seed + 1)
    with torch.no_grad():
        low, up = model(im1, im2, iters=iters, test_mode=True)
    save("e2e_gma_368x496", B=1, H=H, W=W, iters=iters, seed=seed, stride=stride, flow_low=low,
         flow_up_strided=up[:, :, ::stride, ::stride].contiguous(), flow_up_absmean=up.abs().mean())

    seed, H, W, iters = 830, 128, 192, 3
    model = RAFTGMA(gma_ns())
    model.load_state_dict(procedural_state_dict(shapes, seed), strict=False)
    model.train()
    model.freeze_bn()
    im1, im2 = synthetic_pair(2, H, W, seed + 1)
    preds = model(im1, im2, iters=iters)
    n = len(preds)
    loss = 0.0
    for i, p in enumerate(preds):
        loss = loss + (0.8 ** (n - i - 1)) * torch.sqrt(p * p + 1e-6).mean()
    loss.backward()
    d = dict(H=H, W=W, iters=iters, seed=seed, loss=loss.detach(), last=preds[-1].detach())
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        d["gnorm." + k] = g.norm()
    save("train_step_gma", **d)


def gen_train_step():
    """fwd+bwd of the whole model (frozen BN), small shapes: loss + parameter-grad digests."""
    for small, H, W, iters, seed in ((False, 128, 192, 3, 601), (True, 128, 192, 3, 602)):
        model = RAFT(args_ns(small))
        shapes = shapes_of(model)
        model.load_state_dict(procedural_state_dict(shapes, seed))
        model.train()
        model.freeze_bn()
        im1, im2 = synthetic_pair(2, H, W, seed + 1)
        preds = model(im1, im2, iters=iters)
        n = len(preds)
        loss = 0.0
        for i, p in enumerate(preds):
            loss = loss + (0.8 ** (n - i - 1)) * torch.sqrt(p * p + 1e-6).mean()
        loss.backward()
        d = dict(small=small, H=H, W=W, iters=iters, seed=seed, loss=loss.detach(), last=preds[-1].detach())
        for k, p in model.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            d["gnorm." + k] = g.norm()
            d["ghead." + k] = g.reshape(-1)[:32].clone()
        save("train_step_" + ("small" if small else "basic"), **d)


def _train_digest(model, preds, stride=4):
    """loss of the benchmark objective (sum_i 0.8^(n-1-i) mean sqrt(p^2 + 1e-6), zero gt) + backward digests."""
    n = len(preds)
    loss = 0.0
    for i, p in enumerate(preds):
        loss = loss + (0.8 ** (n - i - 1)) * torch.sqrt(p * p + 1e-6).mean()
    loss.backward()
    d = dict(loss=loss.detach(), last=preds[-1].detach()[:, :, ::stride, ::stride].contiguous(),
             first=preds[0].detach()[:, :, ::stride, ::stride].contiguous(), stride=stride)
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        d["gnorm." + k] = g.norm()
        d["ghead." + k] = g.reshape(-1)[:32].clone()
    return d


def gen_bench_scale():
    """The benchmark's own shapes, fwd + bwd, ONE pair, 12 iterations: exercises the 12-segment batched
    weight gradient, the 12-deep stash, the once-per-step context backward; 376x1248 is the KITTI-padded shape of config 4
    (the alt-corr path's oracle is CorrBlock itself); GMA at N = 7040 covers the K = 1536 dattn GEMM."""
    for name, H, W, seed in (("train_step_basic_440x1024", 440, 1024, 611), ("train_step_basic_376x1248", 376, 1248, 612)):
        model = RAFT(args_ns(False))
        model.load_state_dict(procedural_state_dict(shapes_of(model), seed))
        model.train()
        model.freeze_bn()
        im1, im2 = synthetic_pair(1, H, W, seed + 1)
        preds = model(im1, im2, iters=12)
        save(name, small=False, H=H, W=W, iters=12, seed=seed, B=1, **_train_digest(model, preds))
        del model, preds

    from core.gma_network import RAFTGMA
    seed, H, W = 613, 440, 1024
    model = RAFTGMA(gma_ns())
    shapes = shapes_of(model)
    model.load_state_dict(procedural_state_dict(shapes, seed), strict=False)
    with torch.no_grad():
        model.update_block.aggregator.gamma.fill_(0.1)     # zero gamma would leave the aggregate path (and dattn) unexercised
    model.eval()
    im1, im2 = synthetic_pair(1, H, W, seed + 1)
    with torch.no_grad():
        low, up = model(im1, im2, iters=12, test_mode=True)
    save("e2e_gma_440x1024", B=1, H=H, W=W, iters=12, seed=seed, stride=4, gamma=0.1, flow_low=low,
         flow_up_strided=up[:, :, ::4, ::4].contiguous(), flow_up_absmean=up.abs().mean())
    model.train()
    model.freeze_bn()
    preds = model(im1, im2, iters=12)
    save("train_step_gma_440x1024", H=H, W=W, iters=12, seed=seed, B=1, gamma=0.1, **_train_digest(model, preds))


def gen_seq_loss():
    """sequence_loss of pytorch/train.py:60-96.  train.py imports cv2 / tensorflow at module level and cannot be imported
    here, but the function is pure torch: its FunctionDef is taken out of the module's syntax tree and executed with only
    `torch` and MAX_FLOW in scope.  Stored: loss, metrics and d loss / d pred on seeded inputs that include invalid pixels,
    |gt| >= max_flow, and the student / supervisor (gamma / gamma2) halves."""
    import ast
    src = open("/root/reference/pytorch/train.py").read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "sequence_loss"][0]
    scope = {"torch": torch, "MAX_FLOW": 400}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "train.py:sequence_loss", "exec"), scope)
    ref = scope["sequence_loss"]
    d = {}
    for name, B, H, W, n, gamma, gamma2, seed in (("a", 2, 24, 40, 12, 0.8, 1.0, 901), ("b", 1, 17, 23, 6, 0.85, 0.9, 902),
                                                   ("c", 3, 8, 8, 2, 0.8, 1.0, 903)):
        preds = [rand_tensor((B, 2, H, W), seed + 10 + i, 3.0).requires_grad_(True) for i in range(n)]
        gt = rand_tensor((B, 2, H, W), seed + 1, 4.0)
        gt[:, :, 0, :3] = 500.0                                   # |gt| >= max_flow -> excluded from the loss only
        gt[:, 0, 1, 1] = 300.0; gt[:, 1, 1, 1] = 300.0            # each component < 400 but the magnitude is not
        valid = (rand_uniform((B, H, W), seed + 2, 0.0, 1.0) > 0.2).float()
        valid[:, 2, 2] = 0.5                                      # counts for the loss (>= 0.5), not for the metrics (> 0.5)
        loss, metrics = ref(preds, gt, valid, gamma=gamma, gamma2=gamma2)
        loss.backward()
        d.update({f"{name}_cfg": np.array([B, H, W, n, seed], dtype=np.int64), f"{name}_gamma": np.array([gamma, gamma2]),
                  f"{name}_loss": loss.detach(),
                  f"{name}_metrics": np.array([metrics["epe"], metrics["1px"], metrics["3px"], metrics["5px"]])})
        for i, p in enumerate(preds):
            d[f"{name}_dpred{i}"] = p.grad
    save("sequence_loss", **d)


def gen_seq_loss_unsup():
    """sequence_loss_unsup of pytorch/train.py:99-129 (the unlabelled half of the flow-supervisor step), same extraction."""
    ref = _ref_train_fn("sequence_loss_unsup")
    d = {}
    for name, B, H, W, n, gamma, lam, seed in (("a", 2, 24, 40, 24, 0.8, 1.0, 911), ("b", 1, 17, 23, 6, 0.85, 0.25, 912)):
        preds = [rand_tensor((B, 2, H, W), seed + 10 + i, 3.0).requires_grad_(True) for i in range(n)]
        gt = rand_tensor((B, 2, H, W), seed + 1, 4.0)
        valid = (rand_uniform((B, H, W), seed + 2, 0.0, 1.0) > 0.2).float()
        valid[:, 2, 2] = 0.5
        loss, metrics = ref(preds, gt, valid, gamma=gamma, unsup_weight=lam)
        loss.backward()
        d.update({f"{name}_cfg": np.array([B, H, W, n, seed], dtype=np.int64), f"{name}_gamma": np.array([gamma, lam]),
                  f"{name}_loss": loss.detach(),
                  f"{name}_metrics": np.array([metrics["epe"], metrics["1px"], metrics["3px"], metrics["5px"]])})
        for i, p in enumerate(preds):
            d[f"{name}_dpred{i}"] = p.grad if p.grad is not None else torch.zeros_like(p)
    save("sequence_loss_unsup", **d)


def gen_chairs_b8(iters=3):
    """BASELINE.json config 2 at its own batch size: RAFT, 8 pairs, 368x496, fwd + bwd; 3 iterations, and
     the configuration's own 12 as `..._b8_it12` (predictions kept at stride 8)."""
    seed, H, W, B = 621 + (0 if iters == 3 else 7), 368, 496, 8
    model = RAFT(args_ns(False))
    model.load_state_dict(procedural_state_dict(shapes_of(model), seed))
    model.train()
    model.freeze_bn()
    im1, im2 = synthetic_pair(B, H, W, seed + 1)
    preds = model(im1, im2, iters=iters)
    name = "train_step_basic_368x496_b8" + ("" if iters == 3 else f"_it{iters}")
    save(name, small=False, H=H, W=W, iters=iters, seed=seed, B=B, **_train_digest(model, preds, stride=4 if iters == 3 else 8))


def gen_bench_batch():
    """BASELINE.json config 3 at the per-GPU batch bench.py times: RAFT, FOUR pairs of 440x1024, 12
    iterations, fwd + bwd -- loss, strided predictions and the parameter-gradient digests of the benchmarked step itself."""
    seed, H, W, B = 631, 440, 1024, 4
    model = RAFT(args_ns(False))
    model.load_state_dict(procedural_state_dict(shapes_of(model), seed))
    model.train()
    model.freeze_bn()
    im1, im2 = synthetic_pair(B, H, W, seed + 1)
    preds = model(im1, im2, iters=12)
    save("train_step_basic_440x1024_b4", small=False, H=H, W=W, iters=12, seed=seed, B=B, **_train_digest(model, preds, stride=8))


def gen_warm_start():
    """forward_interpolate of the reference (core/utils/utils.py:26-54) on seeded flows: smooth + noise, a case where
    many vectors leave the image, and a constant shift (whole columns of the grid inherit their nearest landed neighbour)."""
    from core.utils.utils import forward_interpolate
    d = {}
    for name, h, w, scale, seed in (("a", 12, 16, 2.5, 801), ("b", 55, 128, 6.0, 802), ("c", 9, 7, 8.0, 803)):
        flow = rand_tensor((2, h, w), seed, scale)
        d["in_" + name] = flow
        d["out_" + name] = forward_interpolate(flow)
    shift = torch.zeros(2, 10, 12)
    shift[0] += 3.25; shift[1] -= 1.5
    d["in_shift"] = shift
    d["out_shift"] = forward_interpolate(shift)
    save("warm_start", **d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["corr", "update", "upsample", "e2e", "kitti", "l2l", "gma", "train", "warm", "seqloss", "bench",
                             "l2l_recipe", "seqloss_unsup", "chairs_b8"]
    if "warm" in which:
        gen_warm_start()
    if "corr" in which:
        gen_corr()
    if "update" in which:
        gen_update()
    if "upsample" in which:
        gen_upsample()
    if "e2e" in which:
        gen_e2e()
    if "kitti" in which:
        gen_kitti()
    if "l2l" in which:
        gen_l2l()
    if "gma" in which:
        gen_gma()
    if "train" in which:
        gen_train_step()
    if "seqloss" in which:
        gen_seq_loss()
    if "bench" in which:
        gen_bench_scale()
    if "l2l_recipe" in which:
        gen_l2l_recipe()
    if "l2l_recipe_kitti" in which:
        gen_l2l_recipe(only=("l2l_recipe_kitti",))
    if "seqloss_unsup" in which:
        gen_seq_loss_unsup()
    if "chairs_b8" in which:
        gen_chairs_b8()
    if "chairs_b8" in which or "chairs_b8_it12" in which:
        gen_chairs_b8(iters=12)
    if "bench_b4" in which:
        gen_bench_batch()
