"""Child process of tests/test_gpu_runtime.py::test_rccl_exchange_at_world_size_one: a fresh process that initialises the
`nccl` (= RCCL) backend with ONE rank and forces FlatGradients to issue its asynchronous bucket all-reduces from the backward
hooks anyway (FSRAFT_DP_FORCE_COLLECTIVE=1; parallel.py skips them at world size 1).  This runs, on the one GPU a test box
has, the part of the N > 1 step no other test reaches: RCCL's own stream against the compute stream around the hook-time
`_foreach_copy_` / `mul_`, `Work.wait()` before clip + AdamW, and a hipGraph capture of a step that contains the collectives.
The sum over one rank is the identity, so every result must equal the no-collective run of the same steps.
usage: _rccl_worker.py port out.json [H W iters batch]"""
import json
import os
import time
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    port, out = sys.argv[1], sys.argv[2]
    H, W, iters, B = (int(v) for v in sys.argv[3:7]) if len(sys.argv) >= 7 else (128, 192, 3, 2)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import argparse
    import copy
    import torch
    import torch.distributed as dist
    from _util import shapes
    from oracle.weights import procedural_state_dict, synthetic_pair
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.train import TrainStep

    torch.cuda.set_device(0)
    res = {"backend": None}
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    res["backend"] = dist.get_backend()

    def fresh():
        m = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False))
        m.load_state_dict(procedural_state_dict(shapes("raft_basic"), 660))
        m = m.to("cuda").train()
        m.freeze_bn()
        return m

    im1, im2 = (t.to("cuda") for t in synthetic_pair(B, H, W, 661))

    def run(force, nsteps=3, graph=False):
        os.environ["FSRAFT_DP_FORCE_COLLECTIVE"] = "1" if force else "0"
        m = fresh()
        step = TrainStep(m, lr=1e-4, iters=iters, capturable=graph)
        assert step.grads.force_collective == force
        losses = []
        if not graph:
            for _ in range(nsteps):
                losses.append(float(step(im1, im2)))
        else:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                losses.append(float(step(im1, im2)))            # eager warm-up step on the capture stream
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            # (as bench.py: the watchdog thread's event queries must not meet a GLOBAL-mode capture)
            time.sleep(0.5)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                loss = step(im1, im2)
            for _ in range(nsteps - 1):
                g.replay()
                torch.cuda.synchronize()
                losses.append(float(loss))
        torch.cuda.synchronize()
        flat = step.grads.flat.detach().clone()
        params = torch.cat([p.detach().reshape(-1) for p in step.grads.params]).clone()
        return losses, flat, params

    l0, g0, p0 = run(False)
    l1, g1, p1 = run(True)
    res["eager"] = {"losses_plain": l0, "losses_rccl": l1, "grad_equal": bool(torch.equal(g0, g1)), "param_equal": bool(torch.equal(p0, p1)),
                    "grad_rel": float((g0 - g1).norm() / g0.norm()), "param_rel": float((p0 - p1).norm() / p0.norm())}
    # the single flat all-reduce in finish() (FSRAFT_DP_BUCKETS=0) through RCCL as well
    os.environ["FSRAFT_DP_BUCKETS"] = "0"
    l2, g2, p2 = run(True)
    os.environ["FSRAFT_DP_BUCKETS"] = "1"
    res["eager_unbucketed"] = {"grad_rel": float((g0 - g2).norm() / g0.norm()), "param_rel": float((p0 - p2).norm() / p0.norm())}
    # whole-step hipGraph capture WITH the collectives inside: what bench.py would need to replay graphs at N > 1
    try:
        l3, g3, p3 = run(True, graph=True)
        res["graph"] = {"ok": True, "losses": l3, "param_rel_vs_eager": float((p0 - p3).norm() / p0.norm()),
                        "loss_rel_vs_eager": [abs(a - b) / abs(a) for a, b in zip(l0, l3)]}
    except Exception as e:                                   # the failure mode is the result
        res["graph"] = {"ok": False, "error": f"{type(e).__name__}: {str(e)[:600]}"}
    with open(out, "w") as f:
        json.dump(res, f)
    try:
        dist.destroy_process_group()
    except Exception:
        pass


if __name__ == "__main__":
    main()
