"""GPU parity tests, rows d / e and the boundary: optimizer, hipGraph replays, bench launch, data parallelism, host threads, devices.
(Split out of the former tests/test_gpu_parity.py in round 6; shared helpers, fixtures and the ONE tolerance table live in
tests/_gpu_common.py.)"""
import pytest

from _gpu_common import *      # noqa: F401,F403  (helpers, fixtures, tolerance table)

pytestmark = pytest.mark.gpu


def test_amax_words_of_external_tensors():
    """fsraft_amax_jobs / fsraft_amax_scaled (include/fsraft.h "amax words", csrc/amax.hip) on tensors no kernel of the library produced:
    the word w must satisfy max|x| / 2 <= w <= 4 max|x| (what a power-of-two scale needs, csrc/split_arith.hpp), 0 for an all-zero
    tensor, NaN for a tensor holding one; rows with a pitch (ld > C), channel counts that are not multiples of four (the scalar path),
    a pointer that is not 16-byte aligned, more than 32 jobs in one call (two launches), several jobs raising ONE word."""
    from flow_supervisor_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)

    def word_of(jobs_of, n=1):
        ws = [ops.new_amax(DEV) for _ in range(n)]
        ops.amax_jobs(jobs_of(ws))
        torch.cuda.synchronize()
        return [float(w.item()) for w in ws]

    def check(w, t, what):
        m = float(t.abs().max().item())
        assert (m == 0.0 and w == 0.0) or (m / 2 <= w <= 4 * m), (what, w, m)

    for n, scale in ((1, 1.0), (7, 1e-6), (4096, 3e4), (1 << 20, 1.0), ((1 << 22) + 3, 1e-3)):
        t = (torch.randn(n, generator=g) * scale).to(DEV)
        check(word_of(lambda ws: [(t.data_ptr(), 1, n, n, ws[0])])[0], t, f"contiguous n={n}")
    z = torch.zeros(1000, device=DEV)
    assert word_of(lambda ws: [(z.data_ptr(), 1, 1000, 1000, ws[0])])[0] == 0.0
    bad = torch.randn(5000, generator=g).to(DEV)
    bad[1234] = float("nan")
    assert math.isnan(word_of(lambda ws: [(bad.data_ptr(), 1, 5000, 5000, ws[0])])[0])
    # rows with a pitch: only the first C columns of every row count (the pad columns hold something larger)
    for rows, C, ld in ((300, 98, 100), (64, 324, 324), (1000, 2, 4), (17, 126, 128)):
        t = torch.randn(rows, ld, generator=g).to(DEV)
        t[:, C:] = 1e6
        check(word_of(lambda ws: [(t.data_ptr(), rows, C, ld, ws[0])])[0], t[:, :C], f"rows={rows} C={C} ld={ld}")
    # a pointer that is only 4-byte aligned
    base = torch.randn(4099, generator=g).to(DEV)
    t = base[3:]
    check(word_of(lambda ws: [(t.data_ptr(), 1, t.numel(), t.numel(), ws[0])])[0], t, "unaligned")
    # 40 jobs in one call, each with its own word; then all of them into one word
    ts = [(torch.randn(100 + 37 * i, generator=g) * (2.0 ** (i - 20))).to(DEV) for i in range(40)]
    ws = word_of(lambda ws: [(t.data_ptr(), 1, t.numel(), t.numel(), w) for t, w in zip(ts, ws)], n=40)
    for i, (w, t) in enumerate(zip(ws, ts)):
        check(w, t, f"job {i} of 40")
    w1 = word_of(lambda ws: [(t.data_ptr(), 1, t.numel(), t.numel(), ws[0]) for t in ts])[0]
    check(w1, torch.cat(ts), "40 jobs, one word")
    # a bound derived from another word: dst = max(dst, factor * src)
    src = ops.amax_tensor(ts[30])
    out = ops.amax_scaled(src, 12.0)
    torch.cuda.synchronize()
    assert float(out.item()) == 12.0 * float(src.item())


def test_flat_adamw_matches_torch_adamw_with_clipping():
    """parallel.FlatAdamW (csrc/optim.hip: clip_grad_norm_ + AdamW as one kernel over flat buffers, pytorch/train.py:137, 280-282)
    against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW over six steps: parameters, clipped gradients, returned norm.
    Tensor sizes that are not multiples of four (the flat layout pads every tensor to 64 floats), a learning-rate change."""
    from flow_supervisor_amd.parallel import FlatAdamW, FlatGradients
    torch.manual_seed(3)
    shapes = [(64, 3, 7, 7), (2,), (17, 5), (1,), (256, 128, 3, 3), (96,), (33,)]
    ours = [torch.nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    grads = FlatGradients(ours)
    opt = FlatAdamW(grads, lr=3e-3, weight_decay=1e-2, eps=1e-8)
    topt = torch.optim.AdamW(ref, lr=3e-3, weight_decay=1e-2, eps=1e-8)
    for p, sh in zip(ours, shapes):
        assert tuple(p.shape) == sh and p.data_ptr() % 256 == 0
    for it in range(6):
        gs = [torch.randn(*sh, device=DEV) * (10.0 if it % 2 else 0.01) for sh in shapes]     # clipped / not clipped
        for p, q, g in zip(ours, ref, gs):
            grads.views[p].copy_(g)
            q.grad = g.clone()
        if it == 3:
            opt.set_lr(1e-3)
            for grp in topt.param_groups:
                grp["lr"] = 1e-3
        tn = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        topt.step()
        n = opt.step(clip=1.0)
        close(n, tn, 0.0, rtol=1e-6, what="gradient norm")
        for p, q in zip(ours, ref):
            close(p, q, 1e-7, rtol=1e-6, what=f"parameters after step {it}")
            close(grads.views[p], q.grad, 1e-9, rtol=1e-6, what="clipped gradient")


def test_flat_adamw_is_a_torch_optimizer():
    """the reference drives its optimizer with StepLR(optimizer, num_steps // 5, 0.5) + scheduler.step() and
    checkpoints it (pytorch/train.py:134-141, 283).  FlatAdamW must take a torch lr_scheduler, round-trip its state through
    state_dict() / load_state_dict(), leave parameters without a gradient (and their moments) alone like torch's AdamW
    skips `grad is None`, and refuse to step once a parameter was re-bound away from its flat buffer."""
    from flow_supervisor_amd.parallel import FlatAdamW, FlatGradients
    torch.manual_seed(4)
    shapes = [(32, 16, 3, 3), (32,), (7, 5), (130,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
    ours = mk()
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    grads = FlatGradients(ours)
    opt = FlatAdamW(grads, lr=2e-3, weight_decay=1e-2)
    topt = torch.optim.AdamW(ref, lr=2e-3, weight_decay=1e-2)
    assert isinstance(opt, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.StepLR(opt, 2, gamma=0.5)
    tsched = torch.optim.lr_scheduler.StepLR(topt, 2, gamma=0.5)

    def one_step(o, ps, gr, sc, skip=()):
        gs = [torch.randn(*sh, device=DEV) for sh in shapes]
        if o is opt:
            gr.begin()
            for i, (p, g) in enumerate(zip(ps, gs)):
                if i not in skip:
                    p.grad = g.clone()
            gr.finish()
            o.step()
        else:
            for i, (p, g) in enumerate(zip(ps, gs)):
                p.grad = None if i in skip else g.clone()
            o.step()
        sc.step()

    for it in range(5):
        skip = (1, 3) if it in (1, 2) else ()
        torch.manual_seed(100 + it); one_step(opt, ours, grads, sched, skip)
        torch.manual_seed(100 + it); one_step(topt, ref, None, tsched, skip)
        assert abs(sched.get_last_lr()[0] - tsched.get_last_lr()[0]) < 1e-12
        if it == 2:
            # torch counts steps per parameter; ours has one counter: parameters that skipped steps 1 and 2 differ from
            # torch's in their bias correction afterwards, so the comparison of THOSE stops here (untouched while skipped)
            for i in (1, 3):
                close(ours[i], ref[i], 1e-7, rtol=1e-6, what="a parameter without gradient is left alone")
        for i, (p, q) in enumerate(zip(ours, ref)):
            if i in (1, 3) and it >= 3:
                continue
            close(p, q, 1e-7, rtol=2e-6, what=f"parameter {i} after step {it} (StepLR lr {sched.get_last_lr()[0]:g})")

    # checkpoint / resume: a fresh optimizer loaded from state_dict() continues identically
    sd = opt.state_dict()
    twins = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    g2 = FlatGradients(twins)
    opt2 = FlatAdamW(g2, lr=1.0)
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"] and float(opt2.lr) == float(opt.lr)
    gs = [torch.randn(*sh, device=DEV) for sh in shapes]
    for o, gr, ps in ((opt, grads, ours), (opt2, g2, twins)):
        gr.begin()
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        gr.finish()
        o.step(clip=1.0)
    for p, q in zip(ours, twins):
        assert torch.equal(p, q), "resumed optimizer diverged"
    with pytest.raises(ValueError):
        FlatAdamW(FlatGradients(mk()[:2])).load_state_dict(sd)

    # a re-bound parameter (model.float() / load_state_dict(assign=True) style) must not be trained silently
    ours[0].data = ours[0].data.clone()
    with pytest.raises(RuntimeError, match="no longer lives"):
        opt.step()


def test_hipgraph_replays_of_the_train_step_follow_the_eager_steps():
    """bench.py times hipGraph replays of the whole train step (one rank).  Replays reuse every buffer of the capture, so
    anything zeroed "once" or by a node the graph drops shows up from the second replay on: ops._ZeroPool (chunks filled once
    per capture, not once per process) and the split-K record GEMMs (their zero fill is a kernel, the hipMemsetAsync node was
    not replayed).  Six replays against six eager steps from the same start: the losses must follow each other."""
    import argparse
    import copy
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.train import TrainStep
    torch.manual_seed(0)
    model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(DEV).train()
    model.freeze_bn()
    twin = copy.deepcopy(model)
    g = torch.Generator(device=DEV).manual_seed(1)
    im1 = torch.rand(2, 3, 184, 320, device=DEV, generator=g) * 255
    im2 = torch.rand(2, 3, 184, 320, device=DEV, generator=g) * 255
    eager = TrainStep(twin, lr=1e-4, iters=4, capturable=True)
    le = [float(eager(im1, im2)) for _ in range(8)]
    step = TrainStep(model, lr=1e-4, iters=4, capturable=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step(im1, im2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        loss = step(im1, im2)
    lg = []
    for _ in range(6):
        graph.replay()
        torch.cuda.synchronize()
        lg.append(float(loss))
    del graph
    # (the first steps of a random-init model on random images move the loss by factors -- 1.2, 12.7, 9.4, 5.1, 1.5, 2.8 ... --
    #  so a wrong update shows as a different sequence, while summation-order noise stays below 1e-3 relative)
    for a, b in zip(le[2:], lg):
        assert abs(a - b) <= 2e-3 * abs(a), (le, lg)


def test_step_replayed_as_two_hipgraphs_with_the_exchange_between_them():
    """bench.py at N > 1: forward + loss + backward as one hipGraph, the all-reduce of the flat gradient buffer issued eagerly,
    clip + AdamW + re-pack as a second hipGraph (TrainStep.forward_backward / exchange / update).  At world size 1 the exchange
    is the identity, so six replayed steps must follow six eager steps of the plain __call__ from the same start."""
    import argparse
    import copy
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.train import TrainStep
    torch.manual_seed(0)
    model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(DEV).train()
    model.freeze_bn()
    twin = copy.deepcopy(model)
    g = torch.Generator(device=DEV).manual_seed(1)
    im1 = torch.rand(2, 3, 184, 320, device=DEV, generator=g) * 255
    im2 = torch.rand(2, 3, 184, 320, device=DEV, generator=g) * 255
    eager = TrainStep(twin, lr=1e-4, iters=4, capturable=True)
    le = [float(eager(im1, im2)) for _ in range(8)]
    step = TrainStep(model, lr=1e-4, iters=4, capturable=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step(im1, im2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g_fb, g_up = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_fb, stream=side):
        loss = step.forward_backward(im1, im2)
    step.exchange()
    with torch.cuda.graph(g_up, stream=side, pool=g_fb.pool()):
        step.update()
    lg = []
    for _ in range(6):
        g_fb.replay()
        step.exchange()
        g_up.replay()
        torch.cuda.synchronize()
        lg.append(float(loss))
    del g_fb, g_up
    for a, b in zip(le[2:], lg):
        assert abs(a - b) <= 2e-3 * abs(a), (le, lg)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_starts_its_own_ranks(tmp_path, world):
    """`python3 bench.py --gpus 2 ...` typed as is, no torchrun and no WORLD_SIZE around it, must start its
    two ranks itself (children, before the parent touches the GPU), run the data-parallel step on both and print ONE JSON line
    with n_gpus = 2.  On a one-GPU box the ranks share cuda:0 and exchange over gloo through host memory; on a box with two
    devices the same command runs on RCCL.  The line must prove that the collective saw both ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # world = 8: the driver's `--gpus 8` command line, here with eight ranks sharing the box's device(s) --
    # control flow, consensus and the per-rank report of the line at the world size the scaling bench runs at
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--height", "128",
                        "--width", "192", "--iters", "3", "--batch-per-gpu", "1", "--no-cpu-baseline", "--no-extra"],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    print(f"bench --gpus {world}:", {k: out[k] for k in ("value", "n_gpus", "ms_per_step", "rccl")})
    assert out["n_gpus"] == world and out["config"]["global_batch"] == world
    assert out["rccl"]["world_size"] == world and out["rccl"]["ranks_seen_by_all_reduce"] == world
    assert out["rccl"]["graph"] == "captured", out["rccl"]
    assert out["value"] > 0 and out["config"]["loss"] == out["config"]["loss"]     # finite loss on the replayed steps
    # the self-diagnosing part of the line: one entry per rank for the wall time and for each of the three parts
    # of the two-graph route, and they add up to the step
    for k in ("per_rank_ms_per_step", "per_rank_graph_fb_ms", "per_rank_exchange_ms", "per_rank_graph_up_ms"):
        assert len(out["rccl"][k]) == world and all(v > 0 for v in out["rccl"][k]), (k, out["rccl"])
    parts = [sum(out["rccl"][k][r] for k in ("per_rank_graph_fb_ms", "per_rank_exchange_ms", "per_rank_graph_up_ms")) for r in range(world)]
    assert all(p <= 1.15 * max(out["rccl"]["per_rank_ms_per_step"]) for p in parts), (parts, out["rccl"])


def test_bench_rank_that_cannot_rendezvous_exits_with_a_message(tmp_path):
    """the first multi-GPU run must not hang.  A rank whose peers never arrive leaves the rendezvous after
    FSRAFT_DIST_TIMEOUT_S with exit code 3 and one line on stderr that names the rank and the rendezvous address."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               FSRAFT_DIST_TIMEOUT_S="8", FSRAFT_BENCH_SHARED_GPUS="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--height", "128",
                        "--width", "192", "--iters", "2", "--batch-per-gpu", "1", "--no-cpu-baseline", "--no-extra"],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert "could not join the process group" in r.stderr and "rank 1" in r.stderr, r.stderr[-1500:]


@pytest.mark.parametrize("global_batch,H,W,iters", [(4, 128, 192, 3), (3, 128, 192, 3), (2, 440, 1024, 12)])
def test_two_process_train_step_matches_single_process(global_batch, H, W, iters, tmp_path):
    """Two fresh processes (tests/_dp_worker.py), each running the real TrainStep on its shard of `global_batch` pairs at
    128x192 x 3 iterations and exchanging the flat gradient (gloo staged through the host: both ranks sit on cuda:0),
    against one process on the whole batch: reduced + clipped flat gradient and post-AdamW weights.  global_batch = 3
    gives shards of 2 and 1 (gradients weighted by local / global batch); the third case is the benchmark's own shape and
    iteration count with one pair per rank."""
    import os
    import socket
    import subprocess
    import sys
    from flow_supervisor_amd.train import TrainStep
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    out = str(tmp_path / "r0.pt")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(global_batch), out, str(H), str(W), str(iters)])
             for r in range(2)]
    rcs = [p.wait(timeout=900) for p in procs]
    assert rcs == [0, 0], rcs
    got = torch.load(out)
    m = _model(False, 650).train()
    m.freeze_bn()
    im1, im2 = (t.to(DEV) for t in synthetic_pair(global_batch, H, W, 651))
    step = TrainStep(m, lr=1e-4, iters=iters)
    loss = step(im1, im2)
    ref_g = step.grads.flat.cpu()
    ref_p = torch.cat([p.detach().reshape(-1).cpu() for p in step.grads.params])
    rel_g = float((got["flat"] - ref_g).norm() / ref_g.norm())
    rel_p = float((got["params"] - ref_p).norm() / ref_p.norm())
    print("dp2 vs single: grad rel", rel_g, "param rel", rel_p, "loss(rank 0 shard)", got["loss"], "loss(all)", float(loss))
    assert rel_g <= 2e-3, rel_g          # a different summation order over the batch
    assert rel_p <= 2e-4, rel_p          # one AdamW step of lr 1e-4: where a gradient is ~0 its sign, hence the update, can differ


def test_rccl_exchange_at_world_size_one(tmp_path):
    """VERDICT r2 next #4 / ADVICE r2: the asynchronous bucket all-reduces issued from the backward hooks had only ever run
    through gloo.  A fresh process (tests/_rccl_worker.py) initialises the nccl (= RCCL) backend with one rank, forces the
    collectives on (FSRAFT_DP_FORCE_COLLECTIVE=1) and runs three real TrainSteps: RCCL's stream ordering against the
    hook-time copies and against clip + AdamW is what N > 1 ranks execute, and a sum over one rank is the identity, so
    gradients and weights must follow the no-collective run (to the run-to-run noise of the atomics in the weight
    gradients).  What one rank cannot show is a data race that only corrupts values when a peer contributes; what it does
    show is that the API sequence (async work handles from autograd's hook thread, wait() before the optimizer, capture)
    runs on RCCL.  The same worker captures a step WITH the
    collectives in a hipGraph and replays it: bench.py enables graphs at N > 1 only because this passes (if RCCL refuses
    capture on some stack the worker records the failure mode and the test reports it)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    out = str(tmp_path / "rccl.json")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    rc = subprocess.run([sys.executable, worker, str(port), out], env=env, timeout=900).returncode
    assert rc == 0, rc
    res = json.load(open(out))
    print("rccl world-1:", json.dumps(res))
    assert res["backend"] == "nccl"
    # (not bit-equal even without a collective: the weight gradients add with fp32 atomics, so two runs of the same three
    #  steps differ in the last bits and AdamW amplifies sign flips of ~0 gradients)
    assert res["eager"]["grad_rel"] <= 5e-3 and res["eager"]["param_rel"] <= 5e-4, res["eager"]
    assert all(abs(a - b) <= 1e-3 * abs(a) for a, b in zip(res["eager"]["losses_plain"], res["eager"]["losses_rccl"])), res["eager"]
    assert res["eager_unbucketed"]["grad_rel"] <= 5e-3 and res["eager_unbucketed"]["param_rel"] <= 5e-4, res["eager_unbucketed"]
    log = os.environ.get("FSRAFT_RCCL_LOG")
    if log:
        with open(log, "w") as f:
            json.dump(res, f, indent=1)
    if res["graph"]["ok"]:
        # replays re-run the same kernels on the same buffers; atomics in the weight gradients reorder sums
        assert res["graph"]["param_rel_vs_eager"] <= 5e-4, res["graph"]
        assert all(r <= 3e-3 for r in res["graph"]["loss_rel_vs_eager"]), res["graph"]
    else:
        pytest.xfail("hipGraph capture of a step containing RCCL all-reduces failed: " + res["graph"]["error"])


def test_two_host_threads_on_two_streams_use_their_own_split_k_scratch():
    """VERDICT r4 next #5 / SURVEY 8b "Threading": the reference's multi-GPU caller is nn.DataParallel (pytorch/train.py:192) --
    one host thread per replica, all inside one process -- and ctypes releases the GIL around every libfsraft call, so two threads'
    calls interleave freely.  Round 4 registered ONE process-wide split-K scratch pointer (fsraft_conv_workspace) and switched it
    on the host: thread A's launch could pick up thread B's buffer.  Now the buffer travels in each call's descriptor.  Two threads,
    each on its own stream, run different small-grid convolutions (the route that parks partial tiles in the scratch) 40 times
    each, concurrently; every result must be bit-equal to the same convolution run alone."""
    import threading
    from flow_supervisor_amd import ops
    runs = [_small_grid_conv(11), _small_grid_conv(12, H=47, W=156, cs=(128, 128), N=128, kh=1, kw=5)]
    alone = [r() for r in runs]
    torch.cuda.synchronize()
    errors, results = [], [None, None]
    barrier = threading.Barrier(2)

    def worker(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                barrier.wait()
                outs = [runs[i]() for _ in range(40)]
                s.synchronize()
            results[i] = outs
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    keys = [k for k in ops._CONV_WS if k[0] == torch.cuda.current_device()]
    assert len(keys) >= 3, keys                  # the main thread's stream and one per worker: nothing shared
    for i in range(2):
        for k, o in enumerate(results[i]):
            assert torch.equal(o, alone[i]), (i, k, float((o - alone[i]).abs().max()))


def test_legacy_workspace_registration_is_per_thread():
    """fsraft_conv_workspace (kept for bindings written against the round-3 header) registers a buffer for the CALLING THREAD only:
    a descriptor without `ws` enqueued by another thread must not touch it (that thread simply gets no split-K route), while the
    registering thread's own call does use it.  Checked through a sentinel pattern in the buffer."""
    import ctypes
    import threading
    from flow_supervisor_amd import _lib, ops
    lib = _lib.load()
    run = _small_grid_conv(13)
    ref = run()
    ws = torch.full((ops.CONV_WS_FLOATS,), -7.0, device=DEV)
    saved = dict(ops._CONV_WS)
    real = ops._conv_workspace
    ops._conv_workspace = lambda device, pixels: None          # descriptors without ws: the registration decides
    try:
        assert lib.fsraft_conv_workspace(ctypes.c_void_p(ws.data_ptr()), ws.numel()) == 0       # this (main) thread
        box = {}

        def other():
            box["out"] = run()
            torch.cuda.synchronize()
        t = threading.Thread(target=other)
        t.start(); t.join()
        assert bool((ws == -7.0).all()), "another thread's launch wrote into this thread's registered scratch"
        close(box["out"], ref, 2e-5, what="convolution without a scratch buffer (no split-K route)")
        mine = run()
        torch.cuda.synchronize()
        assert not bool((ws == -7.0).all()), "the registering thread's own call did not use its scratch"
        assert torch.equal(mine, ref)
    finally:
        lib.fsraft_conv_workspace(ctypes.c_void_p(0), 0)
        ops._conv_workspace = real
        ops._CONV_WS.clear(); ops._CONV_WS.update(saved)


def test_two_host_threads_train_two_models_concurrently():
    """The whole path from two host threads at once (one model replica and one stream per thread, as a DataParallel-style caller
    would drive two devices; here both on the box's one GPU): forward + loss + backward of a small RAFT, three steps each, must give
    the losses and gradients of the same steps run one thread after the other."""
    import threading
    from flow_supervisor_amd.train import raft_sequence_loss

    def steps(seed, out):
        m = _model(False, seed).train()
        m.freeze_bn()
        im1, im2 = (t.to(DEV) for t in synthetic_pair(1, 128, 192, seed + 1))
        for _ in range(3):
            for p in m.parameters():
                p.grad = None
            loss = raft_sequence_loss(m(im1, im2, iters=3))
            loss.backward()
        torch.cuda.current_stream().synchronize()
        out.append((float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))

    serial = [[], []]
    for i in range(2):
        steps(50 + 10 * i, serial[i])
    errors, conc = [], [[], []]

    def worker(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                steps(50 + 10 * i, conc[i])
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for i in range(2):
        (l0, g0), (l1, g1) = serial[i][0], conc[i][0]
        assert abs(l0 - l1) <= 2e-5 * abs(l0), (l0, l1)
        for k in g0:
            if k.startswith("fnet."):
                continue            # (atomically accumulated InstanceNorm statistics: run-to-run noise of its own, TRAIN_TOL)
            close(g1[k], g0[k], 1e-6, rtol=2e-3, what=f"thread {i} {k}")


def test_ops_refuse_tensors_of_another_device():
    """"use the current device + current stream, re-entrant across devices".  The kernels are enqueued on the current
    device's current stream, so tensors living on another device are refused with a RuntimeError (require_cuda_f32) instead of
    being dereferenced by the wrong GPU; the module-level entry points make their input's device current themselves
    (_lib.on_tensor_device).  Needs two devices for the cross-device half."""
    from flow_supervisor_amd import _lib, ops
    from flow_supervisor_amd.core.corr import CorrBlock
    a = torch.randn(1, 64, 8, 12, device=DEV)
    _lib.require_cuda_f32(a, None, a)
    with pytest.raises(RuntimeError):
        _lib.require_cuda_f32(a, a.cpu())
    if torch.cuda.device_count() < 2:
        pytest.skip("one device: the cross-device refusal needs two")
    b = a.to("cuda:1")
    with pytest.raises(RuntimeError, match="one device"):
        _lib.require_cuda_f32(a, b)
    with pytest.raises(RuntimeError, match="current device"):
        ops.to_records(b.permute(0, 2, 3, 1).contiguous())          # current device is cuda:0
    blk = CorrBlock(b, b)                                             # the entry point switches to the tensors' device
    out = blk(torch.zeros(1, 2, 8, 12, device="cuda:1"))
    ref = CorrBlock(a, a)(torch.zeros(1, 2, 8, 12, device=DEV))
    assert out.device == b.device and torch.cuda.current_device() == 0
    close(out.cpu(), ref.cpu(), 1e-6, what="CorrBlock on cuda:1 from a thread whose current device is cuda:0")
