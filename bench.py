#!/usr/bin/env python3
"""Benchmark of the RAFT hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = one full training step of RAFT (full model, 12 GRU iterations) on a batch of
synthetic 440x1024 (= padded 436x1024 Sintel) image pairs, 4 pairs per GPU: forward, sequence
loss, backward, gradient all-reduce over RCCL, gradient clipping and AdamW.  Inputs are resident
in HBM before the timed region.  Rank 0 prints ONE JSON line.

value    = image pairs processed by all ranks / wall time (max over ranks) of K steps
roofline = the dominant hand-written kernel family of the step, timed per launch with events on
           the launching stream inside the timed region (achieved = algorithmic FLOPs or bytes of
           all its launches / their summed duration); `kernels` lists the other families the same
           way, including the HBM-bound correlation build + lookup the north star asks about.
cpu_baseline = the CPU oracle (oracle/raft_torch.py, a port of the reference's PyTorch path) timed
           on this host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MIOpen's default find mode benchmarks every encoder convolution shape on first use (~35 s of start-up, measured);
# the FAST mode picks the same kernels for these shapes without the search.  (The encoders are framework callers.)
os.environ.setdefault("MIOPEN_FIND_MODE", "2")

import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0       # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_MFMA_TF = 157.3    # v_mfma_f32_32x32x2_f32 (exact fp32) dense peak, same guide
PEAK_BF16_MFMA_TF = 2500.0  # v_mfma_f32_32x32x16_f16 / _bf16 dense peak, same guide (not the 2:1-sparse figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=4)
    ap.add_argument("--height", type=int, default=440)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--variant", choices=["raft", "gma", "alt", "l2l", "gma_l2l", "dropin"], default="raft",
                    help="raft: BASELINE.json config 3 (the default, the judged line); gma: config 5 (RAFT-GMA); "
                         "dropin: config 3 through the REFERENCE's model shell over the swapped blocks (INTEGRATION.md section 1); "
                         "alt: config 4 (AlternateCorrBlock, use --height 376 --width 1248 --batch-per-gpu 1); "
                         "l2l / gma_l2l: the flow-supervisor step of the reference recipe (train_semi.sh:3-11): L2L / GMAL2L, one "
                         "labelled + one unlabelled sample per step, crop 368x768 inside a 432x1024 frame, 12 + 12 iterations, "
                         "two backward passes, one AdamW step (use --batch-per-gpu 1, the recipe's batch size)")
    ap.add_argument("--crop-height", type=int, default=368)
    ap.add_argument("--crop-width", type=int, default=768)
    ap.add_argument("--flow-regime", choices=["smooth", "rough"], default="smooth",
                    help="rough: every lookup is centred on flow_init ~ N(0, 8 px at 1/8 resolution) + the iterations' updates, i.e. "
                         "neighbouring queries read unrelated windows (the data-dependent kernels -- lookup, gradient volume, the listed "
                         "volume-backward GEMMs, the alt-corr lookup -- at their other operating point; VERDICT r5 next #5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--one-stream", action="store_true", help="every branch of the forward pass on the caller's stream (core/streams.py "
                    "off): the form the per-kernel profiles are taken in, so that a kernel's duration is its own")
    ap.add_argument("--no-extra", action="store_true", help="skip the exact-fp32 / north_star-encoder short runs and the loss check")
    ap.add_argument("--graph", type=int, default=-1,
                    help="1: capture the whole train step in a hipGraph and time replays; 0: eager; -1: auto")
    return ap.parse_args()


def cpu_baseline(height, width, iters, batch):
    """Oracle fwd+bwd on the host cores (bounded: ~30 s of CPU work): three steps at one pair, one step at the per-GPU batch."""
    from oracle import raft_torch as O
    from oracle.weights import synthetic_pair
    from flow_supervisor_amd.core.raft import RAFT
    # torch's CPU kernels oversubscribe badly on big hosts (256 threads ran this sample 30x slower
    # than 8 did); 16 threads is the sweet spot measured, and `cores` reports what was actually used.
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    im1, im2 = synthetic_pair(batch, height, width, 1234)
    # page-in at a small size so the sample measures compute, not first-touch
    O.sequence_loss_zero_gt(O.raft_forward(sd, im1[:1, :, :64, :128], im2[:1, :, :64, :128], iters=1)).backward()
    nstep = 3
    t0 = time.perf_counter()
    for _ in range(nstep):
        O.sequence_loss_zero_gt(O.raft_forward(sd, im1[:1], im2[:1], iters=iters)).backward()
    dt = (time.perf_counter() - t0) / nstep
    out = {"value": 1.0 / dt, "unit": "image-pairs/s", "cores": cores, "kind": "port",
           "sample": f"1 pair {height}x{width}, {iters} iters, fwd+bwd, {nstep} steps of oracle/raft_torch.py "
                     f"(torch CPU fp32, {cores} threads), {dt:.1f} s per step"}
    if batch > 1:
        t0 = time.perf_counter()
        O.sequence_loss_zero_gt(O.raft_forward(sd, im1, im2, iters=iters)).backward()
        dtb = time.perf_counter() - t0
        out["value_at_gpu_batch"] = batch / dtb
        out["sample"] += f"; {batch} pairs (the per-GPU batch), 1 step: {dtb:.1f} s"
    return out


def corr_isolated(dev, B, H8, W8, iters):
    """Build + `iters` lookups of the correlation path run BACK TO BACK on their own (no update block in between): inside
    the step every lookup starts from caches / TLBs the update block has just flushed, so its in-step time (roofline_corr,
    kernels.corr_lookup_fwd) is higher than what the kernel does on a warm chip.  Same byte model as roofline_corr's
    forward half (build 275.7 MB, lookup 20.4 MB per pair at 55x128)."""
    from flow_supervisor_amd import ops
    from flow_supervisor_amd.core.corr import CorrBlock
    g = torch.Generator(device=dev).manual_seed(5)
    f1 = torch.randn(B, 256, H8, W8, device=dev, generator=g)
    f2 = torch.randn(B, 256, H8, W8, device=dev, generator=g)
    flows = [torch.randn(B, 2, H8, W8, device=dev, generator=g) * 3.0 for _ in range(iters)]
    reps = 5
    with torch.no_grad():
        for timed in (False, True):
            timer = ops.KernelTimer()
            ops.TIMER = timer if timed else None
            for _ in range(reps if timed else 2):
                blk = CorrBlock(f1, f2, radius=4)
                for fl in flows:
                    blk(fl, channels_last=True, is_flow=True)
            torch.cuda.synchronize()
            ops.TIMER = None
    sm = timer.summary()
    out = {}
    tot_b = tot_ms = 0.0
    for fam in ("corr_build", "corr_lookup_fwd"):
        if fam in sm:
            ms = sm[fam]["ms_total"] / reps
            by = sm[fam]["bytes"] / reps
            tot_b += by; tot_ms += ms
            out[fam] = {"avg_launch_us": 1e3 * sm[fam]["ms_avg"], "achieved": by / (ms * 1e-3) / 1e9, "frac": by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
    return {"bound": "hbm", "achieved": tot_b / (tot_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": tot_b / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "ms": tot_ms, "kernels": out,
            "note": f"forward half of the correlation path only (1 build + {iters} lookups, {B} pairs), launched back to back"}


def loss_check(dev, variant="raft", height=440, width=1024, iters=12):
    """The first-step loss of the benchmarked configuration against the reference: the procedural weights and inputs of the
    train-step fixture generated at that very shape by running the reference (tests/golden/make_golden.py) -- RAFT at
    440x1024 / 376x1248 (also the alt-corr variant's oracle) / 8 x 368x496, RAFT-GMA at 440x1024."""
    import json
    import numpy as np
    from oracle.weights import procedural_state_dict, synthetic_pair
    from flow_supervisor_amd.train import raft_sequence_loss
    name = {("raft", 440, 1024): "train_step_basic_440x1024", ("gma", 440, 1024): "train_step_gma_440x1024",
            ("dropin", 440, 1024): "train_step_basic_440x1024",
            ("raft", 376, 1248): "train_step_basic_376x1248", ("alt", 376, 1248): "train_step_basic_376x1248",
            ("raft", 368, 496): "train_step_basic_368x496_b8"}.get((variant, height, width))
    f = os.path.join(ROOT, "tests", "golden", f"{name}.npz") if name else None
    if not f or not os.path.exists(f):
        return None
    g = np.load(f)
    if variant != "gma" and int(g["iters"]) != iters and "b8" not in name:
        return None
    seed = int(g["seed"])
    if variant == "gma":
        from flow_supervisor_amd.core.gma_network import RAFTGMA
        m = RAFTGMA(argparse.Namespace(small=False, mixed_precision=False, dropout=0, num_heads=1, position_only=False,
                                       position_and_content=False, corr_levels=4, corr_radius=4))
        shp = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "raft_gma_shapes.json"))).items()}
        m.load_state_dict(procedural_state_dict(shp, seed), strict=False)
        with torch.no_grad():
            m.update_block.aggregator.gamma.fill_(float(np.load(os.path.join(ROOT, "tests", "golden", "e2e_gma_440x1024.npz"))["gamma"]))
    else:
        from flow_supervisor_amd.core.raft import RAFT
        from flow_supervisor_amd.core.raft_dropin import ReferenceShapedRAFT
        m = (ReferenceShapedRAFT if variant == "dropin" else RAFT)(argparse.Namespace(small=False, mixed_precision=False,
                                                                                         alternate_corr=variant == "alt"))
        m.load_state_dict(procedural_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed))
    m = m.to(dev).train()
    m.freeze_bn()
    im1, im2 = synthetic_pair(int(g["B"]), int(g["H"]), int(g["W"]), seed + 1)
    with torch.no_grad():
        loss = float(raft_sequence_loss(m(im1.to(dev), im2.to(dev), iters=int(g["iters"]))))
    ref = float(g["loss"])
    rel = abs(loss - ref) / abs(ref)
    if rel > 1e-3:
        raise SystemExit(f"bench: first-step loss {loss} differs from the reference's {ref} (rel {rel:.2e})")
    out = {"loss": loss, "reference": ref, "rel_err": rel, "fixture": f"tests/golden/{name}.npz"}
    if int(g["iters"]) != iters:
        out["note"] = f"fixture generated with {int(g['iters'])} iterations (the reference's CPU run at this batch)"
    return out


def step_check(dev, batch, height=440, width=1024, iters=12):
    """The benchmarked step itself against the reference: RAFT at the per-GPU batch bench.py times (four pairs
    of 440x1024, 12 iterations), forward AND backward, on the procedural weights / inputs of tests/golden/
    train_step_basic_440x1024_b4.npz -- which holds the reference's loss and per-parameter gradient digests for that very step
    (tests/golden/make_golden.py::gen_bench_batch).  Compared: the loss, every parameter-gradient norm, the first 32 elements of
    six gradients spread over the model.  Limits = the parity suite's one tolerance table (tests/_gpu_common.py)."""
    import numpy as np
    from oracle.weights import procedural_state_dict, synthetic_pair
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.train import raft_sequence_loss
    f = os.path.join(ROOT, "tests", "golden", "train_step_basic_440x1024_b4.npz")
    if (height, width, iters, batch) != (440, 1024, 12, 4) or not os.path.exists(f):
        return None
    g = np.load(f)
    seed = int(g["seed"])
    m = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False))
    m.load_state_dict(procedural_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed))
    m = m.to(dev).train()
    m.freeze_bn()
    im1, im2 = synthetic_pair(int(g["B"]), int(g["H"]), int(g["W"]), seed + 1)
    loss = raft_sequence_loss(m(im1.to(dev), im2.to(dev), iters=iters))
    loss.backward()
    lv, ref = float(loss), float(g["loss"])
    rel = abs(lv - ref) / abs(ref)
    worst = {"gnorm": 0.0, "gnorm_fnet": 0.0, "ghead": 0.0}
    heads = ("fnet.conv1.weight", "cnet.layer3.0.conv1.weight", "update_block.encoder.convc1.weight", "update_block.gru.convz1.weight",
             "update_block.mask.2.weight", "update_block.flow_head.conv2.weight")
    for k, p in m.named_parameters():
        if p.grad is None or "gnorm." + k not in g:
            continue
        rn = float(g["gnorm." + k])
        e = abs(float(p.grad.norm()) - rn) / (rn + 1e-5)
        key = "gnorm_fnet" if k.startswith("fnet.") else "gnorm"
        worst[key] = max(worst[key], e)
        if k in heads and not k.startswith("fnet."):
            rh = torch.from_numpy(g["ghead." + k])
            worst["ghead"] = max(worst["ghead"], float((p.grad.reshape(-1)[:32].cpu() - rh).abs().max() / (rh.abs().max() + 1e-12)))
    lim = {"loss": 5e-6, "gnorm": 1.5e-3, "gnorm_fnet": 5e-3, "ghead": 1e-2}
    bad = [k for k, v in dict(worst, loss=rel).items() if v > lim[k]]
    if bad:
        raise SystemExit(f"bench: the benchmarked step differs from the reference's beyond the parity limits: {bad}, loss rel {rel:.2e}, {worst}")
    return {"loss": lv, "reference": ref, "rel_err": rel, "grad_norm_rel_err_max": worst["gnorm"], "grad_norm_rel_err_max_fnet": worst["gnorm_fnet"],
            "grad_head_rel_err_max": worst["ghead"], "limits": lim, "batch": int(g["B"]), "backward": True,
            "fixture": "tests/golden/train_step_basic_440x1024_b4.npz"}


def semi_loss_check(dev, variant, height, width, crop_h, crop_w, iters):
    """The two losses of the flow-supervisor step at the reference's recipe (labelled pass: sequence_loss, unlabelled pass:
    sequence_loss_unsup; pytorch/train.py:246-284) against tests/golden/l2l_recipe_{basic,gma}.npz -- generated by running the
    reference's L2L / GMAL2L on the same procedural weights and samples."""
    import json
    import numpy as np
    from oracle.weights import procedural_state_dict, rand_tensor, rand_uniform, synthetic_pair
    from flow_supervisor_amd.train import sequence_loss, sequence_loss_unsup
    tag = "basic" if variant == "l2l" else "gma"
    f = os.path.join(ROOT, "tests", "golden", f"l2l_recipe_{tag}.npz")
    if not os.path.exists(f):
        return None
    g = np.load(f)
    if (int(g["H"]), int(g["W"]), int(g["h"]), int(g["w"]), int(g["iters"])) != (height, width, crop_h, crop_w, 2 * iters):
        return None
    seed = int(g["seed"])
    if tag == "basic":
        from flow_supervisor_amd.core.l2l import L2L
        m = L2L(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False))
    else:
        from flow_supervisor_amd.core.gma_l2l import GMAL2L
        m = GMAL2L(argparse.Namespace(small=False, mixed_precision=False, dropout=0, num_heads=1, position_only=False,
                                      position_and_content=False, corr_levels=4, corr_radius=4))
    shp = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", f"l2l_recipe_{tag}_shapes.json"))).items()}
    m.load_state_dict(procedural_state_dict(shp, seed), strict=False)
    if tag == "gma":
        with torch.no_grad():
            m.update_block.aggregator.gamma.fill_(0.1)
    m = m.to(dev).train()
    m.freeze_bn()
    out = {"fixture": f"tests/golden/l2l_recipe_{tag}.npz"}
    H, W, h, w = height, width, crop_h, crop_w
    for which in ("sup", "unsup"):
        sd = seed + (1 if which == "sup" else 5)
        oy, ox = int(g[which + "_oy"]), int(g[which + "_ox"])
        ci1, ci2 = synthetic_pair(1, H, W, sd)
        im1 = (ci1[:, :, oy:oy + h, ox:ox + w] + rand_tensor((1, 3, h, w), sd + 1, 3.0)).clamp(0, 255).contiguous()
        im2 = (ci2[:, :, oy:oy + h, ox:ox + w] + rand_tensor((1, 3, h, w), sd + 2, 3.0)).clamp(0, 255).contiguous()
        flow = rand_tensor((1, 2, h, w), sd + 3, 4.0).to(dev)
        valid = (rand_uniform((1, h, w), sd + 4, 0.0, 1.0) > 0.1).float().to(dev)
        with torch.no_grad():
            preds = m(im1.to(dev), im2.to(dev), ci1.to(dev), ci2.to(dev), ox, oy, iters=2 * iters, supervisor_grad=which == "sup")
            if which == "sup":
                loss, _ = sequence_loss(preds, flow, valid, float(g["gamma"]))
            else:
                loss, _ = sequence_loss_unsup(preds, flow, valid, unsup_weight=float(g["unsup_lambda"]))
        loss, ref = float(loss), float(g[which + "_loss"])
        rel = abs(loss - ref) / abs(ref)
        if rel > 1e-3:
            raise SystemExit(f"bench: {which} loss {loss} differs from the reference's {ref} (rel {rel:.2e})")
        out[which] = {"loss": loss, "reference": ref, "rel_err": rel}
    return out


def set_arithmetic(split):
    """Switch every GEMM of the path between the split cores (fp16x3 products of scaled operands, the default) and the exact-fp32 MFMA cores."""
    from flow_supervisor_amd import ops
    ops.set_arithmetic(split)


def self_launch(a):
    """`python bench.py --gpus N` typed as is (no torchrun around it, WORLD_SIZE unset): start the N ranks as CHILD processes
    through torch.distributed.run before this process has made any GPU call, let rank 0's JSON line pass through on the
    inherited stdout and exit with the children's code (no exec of anything that touched the GPU).  On a box with fewer than
    N devices the ranks share the devices there are (LOCAL_RANK modulo the device count) and exchange over gloo staged through
    host memory -- RCCL refuses two ranks on one device -- which exercises every line of the N > 1 path but the RCCL call."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()                  # (does not initialise the GPU on this image)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev < a.gpus:
        env["FSRAFT_BENCH_SHARED_GPUS"] = str(max(n_dev, 1))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    from flow_supervisor_amd import ops
    from flow_supervisor_amd.core import streams
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.parallel import barrier, broadcast_parameters, init_distributed, max_over_ranks
    from flow_supervisor_amd.train import TrainStep

    shared = int(os.environ.get("FSRAFT_BENCH_SHARED_GPUS", "0"))
    if shared:                                        # fewer devices than ranks (self_launch): ranks share them, gloo through the host
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % shared)
    # Bounded start-up: the rendezvous and every collective carry a time limit, and a failure here ends THIS
    # process with a clear message and exit code 3 -- torch.distributed.run then tears the other ranks down and reports the failing
    # rank -- instead of a silent hang of the first 8-GPU run.  (Nothing is re-executed: ranks are fresh children of the launcher.)
    init_timeout = float(os.environ.get("FSRAFT_DIST_TIMEOUT_S", "180"))
    try:
        rank, world, local = init_distributed("cuda", backend="gloo" if shared else None, timeout_s=init_timeout)
    except Exception as e:       # noqa: BLE001
        print(f"bench: rank {os.environ.get('RANK', '?')} could not join the process group within {init_timeout:.0f} s "
              f"({type(e).__name__}: {e}); MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} "
              f"WORLD_SIZE={os.environ.get('WORLD_SIZE')} LOCAL_RANK={os.environ.get('LOCAL_RANK')}", file=sys.stderr, flush=True)
        sys.exit(3)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}, "
                         f"or unset WORLD_SIZE and let bench.py start its ranks itself")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    rccl = None
    if world > 1:
        # proof for the reader of the line that the collective library saw every rank: a sum of ones over the data-path backend
        import torch.distributed as dist
        ones = torch.ones(1, device=dev)
        if shared:
            ones = ones.cpu()
        t_first = time.perf_counter()
        try:
            dist.all_reduce(ones)
            seen = int(ones.item())
        except Exception as e:       # noqa: BLE001
            print(f"bench: rank {rank}: the first all-reduce over {dist.get_backend()} failed ({type(e).__name__}: {e}); "
                  f"device cuda:{local}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}", file=sys.stderr, flush=True)
            sys.exit(3)
        if seen != world:            # a collective that did not see every rank would make every later number meaningless: fail the run
            print(f"bench: rank {rank}: the all-reduce of ones returned {seen}, expected {world} ranks", file=sys.stderr, flush=True)
            sys.exit(4)
        #  the short limit above is for the rendezvous and this first collective; the steady-state collectives -- behind
        # code-object loads, a hipGraph capture, MIOpen's first calls, all of which may skew the ranks by minutes on a first
        # 8-GPU run -- get a long one, so that the watchdog does not abort a healthy run
        coll_timeout = float(os.environ.get("FSRAFT_COLLECTIVE_TIMEOUT_S", "1800"))
        try:
            import datetime
            from torch.distributed import distributed_c10d as c10d
            c10d._set_pg_timeout(datetime.timedelta(seconds=coll_timeout), dist.group.WORLD)
        except Exception as e:       # noqa: BLE001  (private API: keep the short limit and say so)
            coll_timeout = init_timeout
            print(f"bench: could not raise the collective timeout ({type(e).__name__}: {e})", file=sys.stderr)
        rccl = {"backend": dist.get_backend() + (" (= RCCL)" if dist.get_backend() == "nccl" else " staged through host memory: ranks share a device"),
                "world_size": dist.get_world_size(), "ranks_seen_by_all_reduce": seen,
                "first_all_reduce_ms": 1e3 * (time.perf_counter() - t_first), "init_timeout_s": init_timeout,
                "collective_timeout_s": coll_timeout}
    # MIOpen exhaustive find (cudnn.benchmark) costs ~7 minutes of start-up for the encoder shapes and
    # is off: the encoders are framework callers of the path, not what this benchmark is about.
    torch.backends.cudnn.benchmark = False

    if a.one_stream:
        streams.OVERLAP = False
    torch.manual_seed(0)
    semi = a.variant in ("l2l", "gma_l2l")
    if semi:
        if (a.height, a.width) == (440, 1024):
            a.height = 432                               # the recipe's uncropped frame: 436x1024 floored to a multiple of 8 (augmentor.py:561-565)
        if a.variant == "l2l":
            from flow_supervisor_amd.core.l2l import L2L
            model = L2L(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train()
        else:
            from flow_supervisor_amd.core.gma_l2l import GMAL2L
            model = GMAL2L(argparse.Namespace(mixed_precision=False, num_heads=1, position_only=False,
                                              position_and_content=False)).to(dev).train()
            with torch.no_grad():
                model.update_block.aggregator.gamma.fill_(0.1)
    elif a.variant == "gma":
        from flow_supervisor_amd.core.gma_network import RAFTGMA
        model = RAFTGMA(argparse.Namespace(mixed_precision=False, num_heads=1, position_only=False,
                                           position_and_content=False)).to(dev).train()
        with torch.no_grad():
            model.update_block.aggregator.gamma.fill_(0.1)      # zero-init gamma would leave the aggregate path unexercised
    elif a.variant == "dropin":
        # the INTEGRATION.md section 1 route: the reference's model shell (NCHW, per-iteration CorrBlock(coords) -> BasicUpdateBlock.forward
        # -> upsample_flow, coords1 carried) over the swapped blocks -- core/raft_dropin.py
        from flow_supervisor_amd.core.raft_dropin import ReferenceShapedRAFT
        model = ReferenceShapedRAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train()
    else:
        model = RAFT(argparse.Namespace(small=False, mixed_precision=False,
                                        alternate_corr=a.variant == "alt")).to(dev).train()
    model.freeze_bn()                                 # pytorch/train.py:203-204
    broadcast_parameters(model)
    # The whole step is captured in a hipGraph and replayed (724 kernel launches per step in the rocprofv3 summary of this command, profiles/r06_kernel_stats.csv; the gaps between dependent kernels of one
    # stream are what the graph removes, and with eight ranks on one host also eight Python threads competing for cores:
    # profiles/r03_host_time.txt).  Several ranks: the step is replayed as TWO graphs -- forward + loss + backward with
    # the gradients gathered into the flat buffer, then clip + AdamW + weight re-packing -- with the ONE all-reduce of the flat
    # gradient buffer issued eagerly between them (TrainStep.forward_backward / exchange / update): no RCCL call inside a
    # capture, no watchdog thread racing a capture, nothing that was not run here on two ranks.  The price is the overlap of the
    # exchange with the encoders' backward (21 MB: a fraction of a millisecond per step).  FSRAFT_BENCH_GRAPH_MULTI=2 captures
    # the whole step with the bucket all-reduces inside (verified on RCCL at world size 1 only, tests/_rccl_worker.py);
    # =0 keeps several ranks eager (overlapped bucket all-reduces from the backward hooks).
    multi = os.environ.get("FSRAFT_BENCH_GRAPH_MULTI", "1")
    use_graph = a.graph == 1 or (a.graph == -1 and (world == 1 or multi != "0"))
    split_graph = use_graph and world > 1 and multi != "2"
    # lr: a small constant (the reference's recipes: AdamW + StepLR(num_steps // 5, 0.5), pytorch/train.py:134-141, with
    # --lr 5e-6 .. 4e-4); throughput does not depend on it, the loss of synthetic steps stays finite with it
    B = a.batch_per_gpu
    g = torch.Generator(device=dev).manual_seed(1234 + rank)

    def pair():
        i1 = torch.rand(B, 3, a.height, a.width, device=dev, generator=g) * 255.0
        i2 = (torch.roll(i1, shifts=(3, -5), dims=(2, 3)) + 2.0 * torch.randn(B, 3, a.height, a.width, device=dev, generator=g)).clamp(0, 255)
        return i1, i2

    im1, im2 = pair()
    if semi:
        from flow_supervisor_amd.train import SemiTrainStep
        lr, lam, gam = (5e-6, 1.0, 0.8) if a.variant == "l2l" else (1e-6, 0.25, 0.85)      # train_semi.sh:3-11
        sstep = SemiTrainStep(model, lr=lr, wdecay=0.0, iters=a.iters, gamma=gam, unsup_lambda=lam, capturable=use_graph)
        ch, cw = a.crop_height, a.crop_width

        def sample(frame, oy, ox):
            f1, f2 = frame
            c1 = (f1[:, :, oy:oy + ch, ox:ox + cw] + 3.0 * torch.randn(B, 3, ch, cw, device=dev, generator=g)).clamp(0, 255).contiguous()
            c2 = (f2[:, :, oy:oy + ch, ox:ox + cw] + 3.0 * torch.randn(B, 3, ch, cw, device=dev, generator=g)).clamp(0, 255).contiguous()
            flow = torch.randn(B, 2, ch, cw, device=dev, generator=g) * 4.0
            valid = (torch.rand(B, ch, cw, device=dev, generator=g) > 0.1).float()
            return (c1, c2, f1, f2, ox, oy, flow, valid)        # offsets as python ints: no device sync in the step

        sup, unsup = sample((im1, im2), 40, 136), sample(pair(), 16, 200)

        def step(_a, _b):
            ls, lu = sstep(sup, unsup)
            return ls + lu
    else:
        tstep0 = TrainStep(model, lr=1.6e-5, iters=a.iters, capturable=use_graph)
        step = tstep0
        if a.flow_regime == "rough":
            # a fixed rough warm start: N(0, 8) cells at 1/8 resolution (64 px), independent per query
            finit = 8.0 * torch.randn(B, 2, a.height // 8, a.width // 8, device=dev, generator=g)

            class _Rough:          # (same call surface as TrainStep for the code below)
                def __call__(self, i1, i2):
                    return tstep0(i1, i2, flow_init=finit)

                def forward_backward(self, i1, i2):
                    return tstep0.forward_backward(i1, i2, flow_init=finit)

                exchange = staticmethod(tstep0.exchange)
                update = staticmethod(tstep0.update)
            step = _Rough()

    eager_step = step
    if split_graph:
        # At N > 1 every step of this run -- warm-up, the captured one, the eager kernel-timing pass -- exchanges gradients the
        # same way: ONE blocking all-reduce of the flat buffer between backward and the update, so the run issues a single
        # kind of collective on a single stream (the hook-issued bucket route is FSRAFT_BENCH_GRAPH_MULTI=0 / 2).
        tstep = sstep if semi else step

        def eager_step(_a, _b):
            l = tstep.forward_backward(sup, unsup) if semi else tstep.forward_backward(im1, im2)
            tstep.exchange()
            tstep.update()
            return l[0] + l[1] if semi else l

    graph = None
    loss = None
    graph_note = "eager"
    if use_graph:
        # Whole-step hipGraph: the step is shape-static, so forward + backward + optimizer are captured
        # once (after eager warm-up on a side stream) and each timed step is one graph launch.  This
        # removes ~25 ms/step of host-side launch latency (700+ kernel launches issued from Python).
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(a.warmup, 2)):
                eager_step(im1, im2)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        cap_kw = {}
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # The process group's watchdog thread polls the events of finished collectives (hipEventQuery).  Under the default
            # GLOBAL capture mode such a call from another thread while this one captures is an error -- it invalidates the
            # capture and the watchdog aborts the process (seen once in ~10 runs of tests/_rccl_worker.py: hipErrorStreamCapture-
            # Unsupported).  Thread-local mode confines the check to the capturing thread.  Before capturing, drain: every
            # work handle of the warm-up steps was waited for by the step itself (FlatGradients.finish), the barrier below is
            # a collective BEHIND them on every rank and the device sync retires its event, so the watchdog has nothing of
            # this process left to poll when the capture begins.
            barrier()
            torch.cuda.synchronize()
            cap_kw["capture_error_mode"] = "thread_local"
        try:
            if split_graph:
                g_fb, g_up = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_fb, stream=side, **cap_kw):   # same stream as the warm-up: autograd pins each AccumulateGrad node to the stream it was created on
                    loss = tstep.forward_backward(sup, unsup) if semi else tstep.forward_backward(im1, im2)
                    if semi:
                        loss = loss[0] + loss[1]
                tstep.exchange()                      # (between the captures as between the replays: the flat buffer holds the first capture's gradients)
                with torch.cuda.graph(g_up, stream=side, pool=g_fb.pool(), **cap_kw):
                    tstep.update()
                graph = (g_fb, g_up)

                part_events = []                      # per timed step: events around the three parts (self-diagnosing N > 1 line)

                def run():
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    ev[0].record()
                    g_fb.replay()
                    ev[1].record()
                    tstep.exchange()                  # eager: one all-reduce of the flat gradient buffer on the current stream
                    ev[2].record()
                    g_up.replay()
                    ev[3].record()
                    part_events.append(ev)
                graph_note = "two hipGraphs (forward + loss + backward | clip + AdamW + re-pack) with the all-reduce issued eagerly between them"
            else:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side, **cap_kw):
                    loss = step(im1, im2)
                run = graph.replay
                graph_note = "hipGraph replay of the whole step" + (" with the bucket all-reduces inside" if world > 1 else "")
        except Exception as e:                       # (never seen; the eager path below is the same step)
            print(f"bench: graph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graph, loss, graph_note = None, None, f"eager (graph capture failed: {type(e).__name__})"
            torch.cuda.synchronize()
        if world > 1:                                # consensus: replays and eager steps must not be mixed across ranks
            ok_all = -max_over_ranks(-(1.0 if graph is not None else 0.0), dev)
            if ok_all < 1.0 and graph is not None:
                graph, loss, graph_note = None, None, "eager (graph capture failed on another rank)"
    if graph is None:
        for _ in range(a.warmup if not use_graph else 0):
            eager_step(im1, im2)

        def run():
            nonlocal loss
            loss = eager_step(im1, im2)
    if graph is not None:
        for _ in range(a.warmup):
            run()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_local = dt
    dt = max_over_ranks(dt, dev)
    if rccl is not None:
        # per-rank view of the timed region: wall time, and for the two-graph route the mean time of each part (forward + backward
        # graph | the all-reduce | clip + AdamW graph) -- a slow rank, a slow link or a slow exchange shows here, not only in `value`
        import torch.distributed as dist
        mine = [1e3 * dt_local / a.steps, 0.0, 0.0, 0.0]
        if split_graph and graph is not None and 'part_events' in locals() and part_events:
            evs = part_events[-a.steps:]
            for k in range(3):
                mine[1 + k] = sum(e[k].elapsed_time(e[k + 1]) for e in evs) / len(evs)
        t_mine = torch.tensor(mine, device=dev, dtype=torch.float64)
        if shared:
            t_mine = t_mine.cpu()
        gathered = [torch.zeros_like(t_mine) for _ in range(world)]
        dist.all_gather(gathered, t_mine)
        rows = [g.tolist() for g in gathered]
        rccl["per_rank_ms_per_step"] = [round(r[0], 3) for r in rows]
        if any(r[1] > 0 for r in rows):
            rccl["per_rank_graph_fb_ms"] = [round(r[1], 3) for r in rows]
            rccl["per_rank_exchange_ms"] = [round(r[2], 3) for r in rows]
            rccl["per_rank_graph_up_ms"] = [round(r[3], 3) for r in rows]

    # Per-launch kernel timing with events on the launching stream.  Eager steps are timed directly
    # inside the region above when --graph 0; with the hipGraph the same step is re-run eagerly right
    # after the timed region (events cannot be read back from inside a captured graph).
    timer = None
    if not a.no_kernel_timing:
        if graph is not None:
            # graph replays update the parameters without touching their Python-side version counters;
            # bump them so the eager pass repacks the GEMM weights from the current values
            ops.parameters_updated(list(model.parameters()))
        timer = ops.KernelTimer()
        ops.TIMER = timer
        tsteps = min(a.steps, 3)
        # (per-kernel durations are taken with the branches of the forward pass on ONE stream: beside another stream's kernels a
        #  launch's events bracket the contention too, and the family sums would no longer add up to a step)
        was_overlap, streams.OVERLAP = streams.OVERLAP, False
        for _ in range(tsteps):
            loss_e = eager_step(im1, im2)
        torch.cuda.synchronize()
        streams.OVERLAP = was_overlap
        ops.TIMER = None
        timer.steps = tsteps
    if graph is not None:
        run()
        torch.cuda.synchronize()
    loss_v = float(loss if loss is not None else loss_e)

    split_mode = os.environ.get("FSRAFT_CONV_SPLIT", "1") != "0"
    extra = {}
    if semi and world == 1 and not a.no_extra:
        # the same optimisation step in the reference's order (two forward / backward passes of one pair each, eager)
        was = sstep.batched
        sstep.batched = not was
        for _ in range(3):
            step(im1, im2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            step(im1, im2)
        torch.cuda.synchronize()
        extra["value_sequential_passes" if was else "value_batched_passes"] = 2 * B * 5 / (time.perf_counter() - t1)
        sstep.batched = was
        extra["semi_step"] = ("labelled + unlabelled sample as ONE batch of two (per-sample crop offsets), one backward" if was else
                              "two forward / backward passes (the reference's order)")
    if world == 1 and a.variant == "raft" and not a.no_extra:
        # the same step with every GEMM on the exact-fp32 MFMA cores, and with the encoders as BASELINE.json's north_star has
        # them (PyTorch-ROCm / MIOpen convolutions): short runs, reported next to `value`
        def short_run(n=5):
            for _ in range(3):          # (MIOpen picks its kernels on the first calls of the north_star encoder configuration)
                step(im1, im2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                step(im1, im2)
            torch.cuda.synchronize()
            return B * n / (time.perf_counter() - t1)
        if split_mode:
            # The same step with every product an exact fp32 MFMA product (v_mfma_f32_32x32x2_f32), timed the way `value` is
            # (VERDICT r5 next #1a): captured as a hipGraph, warmed, a.steps replays between device syncs; then one eager step
            # under the kernel timer for its own roofline figure against the fp32 MFMA peak.
            set_arithmetic(False)
            ops.parameters_updated(list(model.parameters()))
            ex = {"arithmetic": "exact fp32 MFMA products (v_mfma_f32_32x32x2_f32), fp32 accumulation"}
            try:
                if graph is None or split_graph:
                    raise RuntimeError("the main region ran eagerly")
                side2 = torch.cuda.Stream()
                side2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side2):
                    for _ in range(2):
                        step(im1, im2)
                torch.cuda.current_stream().wait_stream(side2)
                torch.cuda.synchronize()
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, stream=side2):
                    step(im1, im2)
                for _ in range(max(a.warmup, 1)):
                    g2.replay()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    g2.replay()
                torch.cuda.synchronize()
                dte = time.perf_counter() - t1
                ex.update(value=B * a.steps / dte, ms_per_step=1e3 * dte / a.steps, steps=a.steps, launch="hipGraph replay of the whole step")
                del g2
            except Exception as e:       # noqa: BLE001
                ex.update(value=short_run(), launch=f"eager, 5 steps ({type(e).__name__}: {e})")
            if not a.no_kernel_timing:
                ops.parameters_updated(list(model.parameters()))
                tm = ops.KernelTimer()
                ops.TIMER = tm
                was_overlap, streams.OVERLAP = streams.OVERLAP, False
                step(im1, im2)
                torch.cuda.synchronize()
                streams.OVERLAP = was_overlap
                ops.TIMER = None
                sm = tm.summary()
                if "conv_igemm" in sm:
                    c = sm["conv_igemm"]
                    tf = c["flops"] / (c["ms_total"] * 1e-3) / 1e12
                    ex["roofline"] = {"kernel": "conv_igemm", "bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
                                      "frac": tf / PEAK_F32_MFMA_TF, "ms_per_step": c["ms_total"], "launches_per_step": c["launches"],
                                      "peak_basis": "algorithmic fp32 FLOPs vs dense fp32 MFMA peak (157.3 TFLOP/s)"}
            extra["exact_f32"] = ex
            extra["value_exact_f32"] = ex["value"]
            set_arithmetic(True)
            ops.parameters_updated(list(model.parameters()))
        if os.environ.get("FSRAFT_ENCODER_CL", "1") != "0":
            os.environ["FSRAFT_ENCODER_CL"] = "0"
            extra["value_north_star_encoders"] = short_run()
            os.environ["FSRAFT_ENCODER_CL"] = "1"
    if world == 1 and not a.no_extra:
        lc = (semi_loss_check(dev, a.variant, a.height, a.width, a.crop_height, a.crop_width, a.iters) if semi
              else loss_check(dev, a.variant, a.height, a.width, a.iters))
        if lc:
            extra["loss_check"] = lc
        if a.variant == "raft" and not semi:
            sc = step_check(dev, B, a.height, a.width, a.iters)
            if sc:
                extra["step_check"] = sc

    if rank != 0:
        return
    pairs = B * world * a.steps * (2 if semi else 1)     # flow-supervisor step: a labelled and an unlabelled pair
    shape_note = " (Sintel 436x1024 padded)" if (a.height, a.width) == (440, 1024) else (" (KITTI 375x1242 padded)" if (a.height, a.width) == (376, 1248) else "")
    out = {
        "metric": f"image-pairs/s fwd+bwd, {a.iters} GRU iters, " + ("436x1024" if (a.height, a.width) == (440, 1024) else f"{a.height}x{a.width}")
                  + (f" (flow-supervisor step: {a.iters}+{a.iters} iters, crop {a.crop_height}x{a.crop_width})" if semi else ""),
        "variant": a.variant,
        "flow_regime": a.flow_regime,
        "value": pairs / dt, "unit": "image-pairs/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (fp16x3 products, 2^-22)" if split_mode else "f32",
        "dtype_note": ("f32 storage and accumulation; every GEMM product evaluated as 3 fp16 MFMA products of operands scaled by their "
                       "tensor's power-of-two scale (hi = fp16(x s), lo = fp16(x s - hi); a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 "
                       "accumulation): <= 2^-22 relative per product, the accuracy class of an fp32 GEMM -- the parity suite holds this "
                       "mode and the exact-fp32 mode to ONE tolerance table, and a convolution's error against fp64 is at or below the "
                       "exact-fp32 MFMA kernel's (profiles/r06_conv_accuracy.txt).  Rounds 1-5 ran bf16x3 (2^-17).  exact_f32 = the same "
                       "step on the exact-fp32 MFMA cores, timed the same way")
                      if split_mode else "exact fp32 MFMA",
        "data": "synthetic",
        "config": {"workload": ({"raft": "RAFT full", "gma": "RAFT-GMA (config 5)", "alt": "RAFT full, AlternateCorrBlock (config 4)",
                                 "dropin": "RAFT full, the REFERENCE's model shell (NCHW tensors, per-iteration CorrBlock(coords) -> "
                                           "BasicUpdateBlock.forward -> upsample_flow, INTEGRATION.md section 1) over the swapped blocks"}[a.variant] +
                                f", {a.height}x{a.width}{shape_note}, {a.iters} GRU iters, "
                                f"{B} pairs/GPU, train step = fwd + sequence loss + bwd + RCCL all-reduce + clip + AdamW") if not semi else
                               (f"flow-supervisor step ({'L2L' if a.variant == 'l2l' else 'GMAL2L'}, pytorch/train.py:246-284): per GPU {B} labelled + {B} "
                                f"unlabelled pair(s), crop {a.crop_height}x{a.crop_width} inside a {a.height}x{a.width} frame, {a.iters} student + "
                                f"{a.iters} supervisor iterations each, sequence_loss / sequence_loss_unsup, two backward passes, RCCL "
                                f"all-reduce, clip, one AdamW step"),
                   "global_batch": B * world, "parallelism": f"dp{world}", "loss": loss_v,
                   "launch": (graph_note + " (exact_f32 likewise; the short run behind value_north_star_encoders and the per-kernel "
                              "timing are eager)") if graph is not None else graph_note,
                   "streams": ("one stream (--one-stream)" if not streams.OVERLAP else
                               "independent branches of the forward pass (context encoder | feature encoder + volume; the motion encoder's "
                               "flow | correlation branch; the flow-supervisor forward's uncropped-frame encodings | student iterations) on "
                               "two HIP streams, their backward likewise (core/streams.py); the per-kernel timing pass runs on one stream"),
                   "encoders": ("MIOpen NCHW convolutions (north_star configuration)" if os.environ.get("FSRAFT_ENCODER_CL", "1") == "0"
                                else "channels_last on the fsraft kernels (7x7 stem on csrc/stem.hip, stride-2 units via space-to-depth; no MIOpen call left); "
                                     "value_north_star_encoders = the same step with the encoders on MIOpen")},
    }
    if rccl is not None:
        rccl["graph"] = "captured" if graph is not None else "eager"
        rccl["reason"] = graph_note
        out["rccl"] = rccl
    out.update(extra)
    if "value_exact_f32" in out or "value_north_star_encoders" in out:
        # the two companions of `value`, spelled out next to the workload: the same step in the reference's own arithmetic
        # (every product an exact fp32 MFMA) and with the encoders as north_star leaves them (PyTorch-ROCm / MIOpen convolutions)
        out["config"]["companions"] = (
            (f"value_exact_f32 = {out['value_exact_f32']:.1f} pairs/s (exact fp32 MFMA products; `value` runs fp16x3 products of scaled "
             f"operands, <= 2^-22 per product, fp32 storage / accumulation; details under exact_f32)" if "value_exact_f32" in out else "") +
            (f"; value_north_star_encoders = {out['value_north_star_encoders']:.1f} pairs/s (encoders on MIOpen NCHW convolutions as north_star "
             f"scopes them, eager)" if "value_north_star_encoders" in out else ""))
    if timer is not None:
        kern = {}
        build_split = split_mode and os.environ.get("FSRAFT_BUILD_SPLIT", "1") != "0"
        split = {"conv_igemm": split_mode, "conv_wgrad": os.environ.get("FSRAFT_WGRAD_SPLIT", "2") != "0", "gemm_f32": True,
                 "altcorr_fwd": split_mode, "altcorr_bwd": split_mode}
        for fam, s in timer.summary().items():
            mfma = fam in ("conv_igemm", "conv_wgrad", "gemm_f32", "altcorr_fwd", "altcorr_bwd")
            sec = s["ms_total"] * 1e-3
            basis = None
            if mfma and split.get(fam):
                # split: every algorithmic fp32 product costs 3 fp16 MFMA products (fp16 and bf16 MFMA issue at the same rate), so the
                # ceiling for ALGORITHMIC flops is the dense 16-bit MFMA peak / 3
                ach, peak, unit = s["flops"] / sec / 1e12, PEAK_BF16_MFMA_TF / 3.0, "TFLOP/s"
                basis = "algorithmic fp32 FLOPs vs dense fp16 / bf16 MFMA peak (2500 TFLOP/s) / 3 MFMA products per fp32 product"
                if fam == "altcorr_fwd":
                    basis += ("; algorithmic = the (2r+2)^2 window products per query and level (alt_cuda_corr's count) -- the tile GEMM "
                              "multiplies whole regions, ~2.6x that")
                if fam == "altcorr_bwd":
                    basis += "; algorithmic = alt_cuda_corr.backward's window products; computed as chunks of the gradient volume + record GEMMs"
            elif mfma:
                ach, peak, unit = s["flops"] / sec / 1e12, PEAK_F32_MFMA_TF, "TFLOP/s"
                basis = ("algorithmic fp32 FLOPs vs the fp32 vector / MFMA peak (157.3 TFLOP/s); the kernel is a vector-ALU dot-product "
                         "kernel fed from L2" if fam == "altcorr_fwd" else
                         "FLOPs of alt_cuda_corr.backward's window products vs the fp32 peak; computed here as chunks of the gradient "
                         "volume + fp16x3 GEMMs" if fam == "altcorr_bwd" else "algorithmic fp32 FLOPs vs dense fp32 MFMA peak")
            else:
                ach, peak, unit = s["bytes"] / sec / 1e9, PEAK_HBM_GBS, "GB/s"
            kern[fam] = {"bound": "mfma" if mfma else "hbm", "achieved": ach, "peak": peak, "unit": unit,
                         "frac": ach / peak, "traffic": None, "launches_per_step": s["launches"] / timer.steps,
                         "ms_per_step": s["ms_total"] / timer.steps, "avg_launch_us": 1e3 * s["ms_avg"]}
            if basis:
                kern[fam]["peak_basis"] = basis
            if mfma and s.get("flops_done", s["flops"]) < 0.999 * s["flops"]:
                # the volume-backward GEMMs visit only the k-tiles the step's lookups reached: `frac` prices what was really
                # multiplied, `frac_dense` the dense contraction the launch replaces (a model number, not a hardware one)
                kern[fam]["frac_dense"] = kern[fam]["frac"]
                kern[fam]["achieved_dense"] = ach
                kern[fam]["achieved"] = s["flops_done"] / sec / 1e12
                kern[fam]["frac"] = kern[fam]["achieved"] / peak
                kern[fam]["k_tiles_visited"] = s["flops_done"] / s["flops"]
                kern[fam]["peak_basis"] = (basis or "") + "; frac = FLOPs of the k-tiles actually visited, frac_dense = the dense contraction's"
            if fam == "corr_build":     # both views: HBM (the north-star bound) and the matrix pipe in the arithmetic actually used
                mpeak = PEAK_BF16_MFMA_TF / 3.0 if build_split else PEAK_F32_MFMA_TF
                kern[fam]["mfma_tflops"] = s["flops"] / sec / 1e12
                kern[fam]["mfma_frac"] = s["flops"] / sec / 1e12 / mpeak
                kern[fam]["mfma_peak_basis"] = ("dense fp16 / bf16 MFMA peak (2500 TFLOP/s) / 3 products per fp32 product" if build_split
                                                else "dense fp32 MFMA peak (157.3 TFLOP/s)")
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
        out["roofline"] = dict(kern[dom], kernel=dom)
        out["kernels"] = kern
        # the north-star quantity: algorithmic bytes of the whole correlation path (build, lookups, their
        # backward incl. the zero fill of the gradient volume, build backward) over the time of every kernel that serves it
        fams = [f for f in ("corr_build", "corr_lookup_fwd", "corr_lookup_bwd", "corr_build_bwd") if f in kern]
        if len(fams) == 4:
            summ = timer.summary()
            byt = sum(summ[f]["bytes"] for f in fams) / timer.steps
            ms = sum(kern[f]["ms_per_step"] for f in fams)
            out["roofline_corr"] = {"bound": "hbm", "achieved": byt / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": byt / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes_per_step": byt, "ms_per_step": ms,
                                    "kernels": fams, "traffic": None,
                                    "note": "algorithmic bytes of SURVEY.md 8d (1.452 GB per pair at 55x128, 12 lookups) / summed "
                                            "kernel time of build + lookups + gradient volume + build backward"}
        # ... and its forward half alone, which is what BASELINE.json's north_star names ("4D corr build+lookup"): the in-step build
        # and the twelve lookups
        ff = [f for f in ("corr_build", "corr_lookup_fwd") if f in kern]
        if len(ff) == 2:
            summ = timer.summary()
            byt = sum(summ[f]["bytes"] for f in ff) / timer.steps
            ms = sum(kern[f]["ms_per_step"] for f in ff)
            out["roofline_corr_fwd"] = {"bound": "hbm", "achieved": byt / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                        "frac": byt / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes_per_step": byt, "ms_per_step": ms,
                                        "kernels": ff, "traffic": None,
                                        "note": "forward half of roofline_corr, timed inside the step: volume build + pyramid (275.7 MB per "
                                                "pair) and the lookups (20.4 MB per pair and iteration)"}
        if "corr_lookup_bwd" in kern:
            kern["corr_lookup_bwd"]["model_note"] = ("frac is against SURVEY.md 8d's byte model of the REFERENCE algorithm (per lookup: read dOut, "
                                                     "read-modify-write the window taps; plus one zero fill of the dense gradient): the kernel "
                                                     "reads dOut once and writes only the records the backward GEMMs read, so a model fraction "
                                                     "above 1 is possible; hbm_gbs_counters is its real rate from the PMC bytes")
        tr = os.path.join(ROOT, "profiles", "traffic.json")     # PMC-derived HBM bytes per launch, if profiled
        t = json.load(open(tr)) if os.path.exists(tr) else None
        this_key = {"variant": a.variant, "height": a.height, "width": a.width, "batch_per_gpu": B, "iters": a.iters}
        if t is not None and t.get("_meta", {}).get("key") != this_key:
            # the counters were collected on ANOTHER workload: a family's per-launch bytes there say nothing about this shape's launches
            # (VERDICT r4 weak #8a) -- `traffic` stays null and the line says which workload the committed file belongs to
            out["traffic_source"] = {"applied": False, "file_key": t.get("_meta", {}).get("key"), "this_run": this_key}
            t = None
        if t is not None:
            for fam, v in t.items():
                if fam in kern and not fam.startswith("_"):                   # PMC bytes per LAUNCH (family average), like `avg_launch_us`; x launches_per_step = per step
                    kern[fam]["traffic"] = v
                    kern[fam]["traffic_unit"] = "HBM bytes per launch (family average; FETCH_SIZE x 2 + WRITE_SIZE, profiles/README.md)"
                    kern[fam]["traffic_per_step"] = v * kern[fam]["launches_per_step"]
                    if not (fam in ("conv_igemm", "conv_wgrad", "gemm_f32", "altcorr_fwd", "altcorr_bwd")):
                        kern[fam]["hbm_gbs_counters"] = v / (kern[fam]["avg_launch_us"] * 1e-6) / 1e9
            for k in ("traffic", "traffic_unit", "traffic_per_step"):
                if k in kern[dom]:
                    out["roofline"][k] = kern[dom][k]
            if "roofline_corr_fwd" in out and all(kern[f].get("traffic_per_step") for f in out["roofline_corr_fwd"]["kernels"]):
                out["roofline_corr_fwd"]["traffic"] = sum(kern[f]["traffic_per_step"] for f in out["roofline_corr_fwd"]["kernels"])
            if "_meta" in t:
                out["traffic_source"] = dict(t["_meta"], applied=True)
            if "roofline_corr" in out and all(kern[f].get("traffic_per_step") for f in out["roofline_corr"]["kernels"] if f != "corr_build_bwd"):
                out["roofline_corr"]["traffic"] = sum(kern[f].get("traffic_per_step") or 0.0 for f in out["roofline_corr"]["kernels"])
                out["roofline_corr"]["traffic_unit"] = "HBM bytes per step over the families that were profiled (corr_build_bwd: its two GEMMs are counted under gemm_f32)"
        # A roofline fraction above 1 is a statement about the byte MODEL, not about the hardware (VERDICT r4 weak #3 / #8b): flag
        # every such component, and give the whole path a second time on the bytes the counters saw.
        for fam, kv in kern.items():
            if kv["frac"] > 1.0:
                kv["frac_above_one"] = ("model bytes / FLOPs exceed what the kernel moves: this fraction is NOT a hardware roofline "
                                        "fraction; see hbm_gbs_counters / traffic")
        for name in ("roofline_corr", "roofline_corr_fwd"):
            rc = out.get(name)
            if rc is None:
                continue
            over = [f for f in rc["kernels"] if kern[f]["frac"] > 1.0]
            if over:
                rc["components_above_one"] = over
                capped = sum(min(timer.summary()[f]["bytes"] / timer.steps, kern[f]["ms_per_step"] * 1e-3 * PEAK_HBM_GBS * 1e9) for f in rc["kernels"])
                rc["frac_capped"] = capped / (rc["ms_per_step"] * 1e-3) / 1e9 / PEAK_HBM_GBS
                rc["frac_capped_note"] = "every component's model bytes capped at what 8 TB/s could move in its own time"
            if rc.get("traffic"):
                rc["frac_counters"] = rc["traffic"] / (rc["ms_per_step"] * 1e-3) / 1e9 / PEAK_HBM_GBS
                rc["frac_counters_note"] = ("the same kernels' HBM bytes from the PMC counters (profiles/traffic.json, same workload) over "
                                            "the same time: real traffic, wasted re-reads included" +
                                            ("; corr_build_bwd's GEMM bytes are not in it (counted under gemm_f32)" if name == "roofline_corr" else ""))
    if world == 1 and not a.no_extra and a.variant == "raft" and timer is not None and "roofline_corr" in out:
        out["roofline_corr_isolated"] = corr_isolated(dev, B, a.height // 8, a.width // 8, a.iters)
    if world == 1 and not a.no_cpu_baseline and a.variant == "raft":
        out["cpu_baseline"] = cpu_baseline(a.height, a.width, a.iters, B)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
