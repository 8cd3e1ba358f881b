"""Child process of tests/test_gpu_runtime.py::test_two_process_train_step_matches_single_process: one data-parallel rank
running the REAL TrainStep (HIP path) on its shard.  Ranks share cuda:0, so the exchange goes over gloo staged through
host memory (RCCL refuses two ranks on one device); everything else is the code path bench.py runs at N > 1.
usage: _dp_worker.py rank world port global_batch out.pt [H W iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, gb, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    H, W, iters = (int(v) for v in sys.argv[6:9]) if len(sys.argv) >= 9 else (128, 192, 3)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    import argparse
    import torch
    from _util import shapes
    from oracle.weights import procedural_state_dict, synthetic_pair
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.parallel import barrier, broadcast_parameters, init_distributed, shard_batch
    from flow_supervisor_amd.train import TrainStep
    init_distributed("cuda", backend="gloo")
    m = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False))
    # rank 0 holds the reference weights, the others start elsewhere: the broadcast has to bring them over
    m.load_state_dict(procedural_state_dict(shapes("raft_basic"), 650 + 7 * rank))
    m = m.to("cuda").train()
    m.freeze_bn()
    im1, im2 = (t.to("cuda") for t in synthetic_pair(gb, H, W, 651))
    step = TrainStep(m, lr=1e-4, iters=iters)
    # a forward BEFORE the broadcast fills the packed-weight caches with the pre-broadcast values (a broadcast
    # through .data would leave them stale)
    with torch.no_grad():
        m(im1[:1], im2[:1], iters=1)
    broadcast_parameters(m)
    s, n = shard_batch(gb, rank, world)
    loss = step(im1[s:s + n], im2[s:s + n], global_batch=gb)
    torch.cuda.synchronize()
    barrier()
    if rank == 0:
        torch.save({"flat": step.grads.flat.cpu(), "loss": float(loss),
                    "params": torch.cat([p.detach().reshape(-1).cpu() for p in step.grads.params])}, out)


if __name__ == "__main__":
    main()
