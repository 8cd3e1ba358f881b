"""world_size-2 checks of the data-parallel layer on CPU (gloo): flat-gradient all-reduce,
batch sharding, parameter broadcast and the max-over-ranks timing reduction bench.py uses."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_loss(preds, gamma=0.8):
    """Charbonnier sequence objective in plain torch (the product's fused loss kernel needs a GPU; this test is about the
    gradient exchange, on CPU)."""
    n = len(preds)
    return sum((gamma ** (n - i - 1)) * torch.sqrt(p * p + 1e-6).mean() for i, p in enumerate(preds))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from flow_supervisor_amd.parallel import (FlatGradients, barrier, broadcast_parameters, init_distributed,
                                              max_over_ranks, shard_batch)
    r, w, _ = init_distributed("cpu")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # deliberately different init per rank
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))
    broadcast_parameters(model)
    grads = FlatGradients([p for _, p in model.named_parameters()], [n for n, _ in model.named_parameters()])
    assert len(grads.buckets) == 2                     # one bucket per top-level module ("0", "2")
    torch.manual_seed(7)
    x = torch.randn(6, 3, 10, 12)                      # the GLOBAL batch, identical on every rank
    s, n = shard_batch(6, rank, world)
    grads.begin()                                      # arms the per-bucket hooks: exchanges start during backward
    # per-rank mean loss over its shard; mean over ranks of per-rank means == global mean (equal shards)
    loss = _toy_loss([model(x[s:s + n]), 0.5 * model(x[s:s + n])])
    loss.backward()
    grads.all_reduce_mean_()
    total = grads.clip_norm_(1.0)
    barrier()
    t = max_over_ranks(1.0 + rank, torch.device("cpu"))
    if rank == 0:
        torch.save({"flat": grads.flat.clone(), "w0": model[0].weight.detach().clone(), "t": t, "norm": total}, out)


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    # single-process reference on the whole batch with rank 0's initial weights
    sys.path.insert(0, ROOT)
    from flow_supervisor_amd.parallel import FlatGradients, shard_batch
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))
    assert torch.equal(model[0].weight, got["w0"])     # broadcast from rank 0 took effect
    grads = FlatGradients(model.parameters())
    torch.manual_seed(7)
    x = torch.randn(6, 3, 10, 12)
    _toy_loss([model(x), 0.5 * model(x)]).backward()
    ref_norm = grads.flat.norm()
    grads.clip_norm_(1.0)
    assert torch.allclose(got["flat"], grads.flat, atol=1e-6, rtol=1e-5)
    assert abs(float(got["norm"]) - float(ref_norm)) < 1e-5
    assert got["t"] == 2.0                               # max over ranks of (1, 2)
    assert [shard_batch(7, r, 3) for r in range(3)] == [(0, 3), (3, 2), (5, 2)]


def _worker_unequal(rank, world, port, out):
    """global batch 5 on 2 ranks -> shards of 3 and 2; broadcast AFTER a first use of the weights (resume / re-sync)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from flow_supervisor_amd.parallel import FlatGradients, broadcast_parameters, init_distributed, shard_batch
    init_distributed("cpu")
    torch.manual_seed(200 + rank)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))
    versions = [p._version for p in model.parameters()]
    broadcast_parameters(model)
    # the packed-weight caches of the product key on (data_ptr, _version): a broadcast must be visible there
    assert all(p._version > v for p, v in zip(model.parameters(), versions)), "broadcast did not bump the version counters"
    grads = FlatGradients(model.parameters())
    torch.manual_seed(8)
    x = torch.randn(5, 3, 10, 12)
    s, n = shard_batch(5, rank, world)
    grads.begin(n, 5)
    _toy_loss([model(x[s:s + n]), 0.5 * model(x[s:s + n])]).backward()
    grads.all_reduce_mean_()
    if rank == 0:
        torch.save({"flat": grads.flat.clone()}, out)


def test_unequal_shards_are_weighted_by_local_batch(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_unequal, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    sys.path.insert(0, ROOT)
    from flow_supervisor_amd.parallel import FlatGradients
    torch.manual_seed(200)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))
    grads = FlatGradients(model.parameters())
    torch.manual_seed(8)
    x = torch.randn(5, 3, 10, 12)
    _toy_loss([model(x), 0.5 * model(x)]).backward()
    assert torch.allclose(got["flat"], grads.flat, atol=1e-6, rtol=1e-5)


def _worker_two_passes(rank, world, port, out, bucketed):
    """The flow-supervisor step: a labelled and an unlabelled forward/backward feed ONE optimizer step
    (pytorch/train.py:270-277).  begin(backward_passes=2) holds every bucket back until both have arrived; a third
    backward must raise instead of adding unreduced gradients into the already-exchanged views."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      FSRAFT_DP_BUCKETS="1" if bucketed else "0")
    torch.set_num_threads(1)
    from flow_supervisor_amd.parallel import FlatGradients, broadcast_parameters, init_distributed, shard_batch
    init_distributed("cpu")
    torch.manual_seed(300 + rank)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))
    broadcast_parameters(model)
    grads = FlatGradients([p for _, p in model.named_parameters()], [n for n, _ in model.named_parameters()])
    assert grads.bucketed == bucketed
    torch.manual_seed(9)
    x, xu = torch.randn(4, 3, 10, 12), torch.randn(4, 3, 10, 12)
    s, n = shard_batch(4, rank, world)
    grads.begin(backward_passes=2)
    _toy_loss([model(x[s:s + n])]).backward()
    assert not any(grads._done), "a bucket was exchanged after the first of two backward passes"
    _toy_loss([0.25 * model(xu[s:s + n])]).backward()
    assert all(grads._done) == bucketed                # bucketed: both buckets left from the hooks of the second pass
    grads.all_reduce_mean_()
    flat = grads.flat.clone()
    raised = False
    if bucketed:                                       # next step: one pass announced, two made
        grads.begin()
        _toy_loss([model(x[s:s + n])]).backward()
        try:
            _toy_loss([model(x[s:s + n])]).backward()
        except RuntimeError as e:
            raised = "already exchanged" in str(e)
        grads.all_reduce_mean_()
    if rank == 0:
        torch.save({"flat": flat, "raised": raised}, out)


def _two_pass_reference():
    sys.path.insert(0, ROOT)
    from flow_supervisor_amd.parallel import FlatGradients
    torch.manual_seed(300)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))
    grads = FlatGradients(model.parameters())
    torch.manual_seed(9)
    x, xu = torch.randn(4, 3, 10, 12), torch.randn(4, 3, 10, 12)
    _toy_loss([model(x)]).backward()
    _toy_loss([0.25 * model(xu)]).backward()
    return grads.flat


def test_two_backward_passes_per_step_and_unannounced_third(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_two_passes, args=(2, _free_port(), out, True), nprocs=2, join=True)
    got = torch.load(out)
    assert got["raised"], "a gradient for an already-exchanged bucket must raise"
    assert torch.allclose(got["flat"], _two_pass_reference(), atol=1e-6, rtol=1e-5)


def test_single_flat_allreduce_switch(tmp_path):
    """FSRAFT_DP_BUCKETS=0: nothing is exchanged from the hooks, one all-reduce of the flat buffer in finish()."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_two_passes, args=(2, _free_port(), out, False), nprocs=2, join=True)
    got = torch.load(out)
    assert torch.allclose(got["flat"], _two_pass_reference(), atol=1e-6, rtol=1e-5)


def _worker_deferred(rank, world, port, out):
    """The route of a step replayed as two hipGraphs (bench.py at N > 1): backward only gathers (begin(exchange=False)), the ONE
    all-reduce of the whole flat buffer is issued afterwards (exchange_all); unequal shards keep their weights."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from flow_supervisor_amd.parallel import FlatGradients, broadcast_parameters, init_distributed, shard_batch
    init_distributed("cpu")
    torch.manual_seed(100 + rank)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))
    broadcast_parameters(model)
    grads = FlatGradients([p for _, p in model.named_parameters()], [n for n, _ in model.named_parameters()])
    torch.manual_seed(7)
    x = torch.randn(5, 3, 10, 12)
    s, n = shard_batch(5, rank, world)
    grads.begin(n, 5, exchange=False)
    _toy_loss([model(x[s:s + n]), 0.5 * model(x[s:s + n])]).backward()
    grads.finish()
    local = grads.flat.clone()                         # nothing was exchanged yet: this rank's own gradient
    grads.exchange_all()
    if rank == 0:
        torch.save({"flat": grads.flat.clone(), "local": local}, out)


def test_deferred_exchange_is_one_allreduce_after_backward(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_deferred, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    sys.path.insert(0, ROOT)
    from flow_supervisor_amd.parallel import FlatGradients
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))
    grads = FlatGradients(model.parameters())
    torch.manual_seed(7)
    x = torch.randn(5, 3, 10, 12)
    _toy_loss([model(x), 0.5 * model(x)]).backward()
    assert torch.allclose(got["flat"], grads.flat, atol=1e-6, rtol=1e-5)
    assert not torch.allclose(got["local"], grads.flat, atol=1e-6, rtol=1e-3)      # rank 0's shard alone is not the global gradient
