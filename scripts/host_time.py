#!/usr/bin/env python3
"""Host-side cost of one train step: CPU time to ENQUEUE a step (process CPU time, no synchronisation inside) for eager steps and
for hipGraph replays, alone and with N processes doing the same at once on one host (all on GPU 0: the wall times then say
nothing, the CPU times per step do -- eight ranks of a node share the host's cores the same way).
usage: python scripts/host_time.py [--procs 8]        (GPU box)"""
import argparse
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")


def worker():
    import torch
    from flow_supervisor_amd.core.raft import RAFT
    from flow_supervisor_amd.train import TrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).to(dev).train()
    model.freeze_bn()
    step = TrainStep(model, lr=1.6e-5, iters=12, capturable=True)
    im1 = torch.rand(4, 3, 440, 1024, device=dev) * 255
    im2 = torch.rand(4, 3, 440, 1024, device=dev) * 255
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            step(im1, im2)
    torch.cuda.synchronize()
    n = 6
    with torch.cuda.stream(side):
        c0, w0 = time.process_time(), time.perf_counter()
        for _ in range(n):
            step(im1, im2)
        c1, w1 = time.process_time(), time.perf_counter()
        torch.cuda.synchronize()
        w2 = time.perf_counter()
    eager_cpu, eager_enq, eager_wall = (c1 - c0) / n, (w1 - w0) / n, (w2 - w0) / n
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step(im1, im2)
    torch.cuda.synchronize()
    c0, w0 = time.process_time(), time.perf_counter()
    for _ in range(n):
        g.replay()
    c1 = time.process_time()
    torch.cuda.synchronize()
    w2 = time.perf_counter()
    print(f"pid {os.getpid()}: eager cpu {1e3 * eager_cpu:.1f} ms/step (enqueue wall {1e3 * eager_enq:.1f}, step wall {1e3 * eager_wall:.1f});  "
          f"graph replay cpu {1e3 * (c1 - c0) / n:.2f} ms/step (step wall {1e3 * (w2 - w0) / n:.1f})", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        worker()
    else:
        print(f"host: {os.cpu_count()} cores; {a.procs} process(es) at once", flush=True)
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"]) for _ in range(a.procs)]
        rc = max(p.wait() for p in ps)
        sys.exit(rc)
