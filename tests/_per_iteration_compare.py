"""Helper of tests/test_gpu_end_to_end.py::test_per_step_batches_against_the_per_iteration_path_on_odd_shapes (not collected by
pytest): the per-step batches (update.HeadBatch / MotionBatch / batched heads backward) against the per-iteration path on odd
shapes and iteration counts -- RAFT, RAFT with alt-corr, L2L.  Run directly it prints the worst relative difference of the
parameter gradients per case: python tests/_per_iteration_compare.py (GPU box)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flow_supervisor_amd.core import update as U  # noqa: E402
from flow_supervisor_amd.core.l2l import L2L  # noqa: E402
from flow_supervisor_amd.core.raft import RAFT  # noqa: E402
from flow_supervisor_amd.train import raft_sequence_loss  # noqa: E402

dev = "cuda"


def grads(model_fn, call, on, sup_k=None):
    U.HEAD_BATCH = U.MOTION_BATCH = U.HEADS_BWD_BATCH = on
    torch.manual_seed(7)
    m = model_fn().to(dev).train()
    m.freeze_bn()
    preds = call(m)
    if sup_k is None:
        raft_sequence_loss(preds).backward()
    else:        # the promise behind sup_grad_samples: the supervisor's predictions of the samples behind k get no gradient
        half = len(preds) // 2
        (raft_sequence_loss(preds[:half]) + raft_sequence_loss([p[:sup_k] for p in preds[half:]])).backward()
    return [p.detach().clone() for p in preds], {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


def compare(name, model_fn, call, sup_k=None, ref_call=None):
    was = (U.HEAD_BATCH, U.MOTION_BATCH, U.HEADS_BWD_BATCH)
    try:
        pa, ga = grads(model_fn, call, True, sup_k)
        pb, gb = grads(model_fn, ref_call or call, False, sup_k)
    finally:
        U.HEAD_BATCH, U.MOTION_BATCH, U.HEADS_BWD_BATCH = was
    dp = max((a - b).abs().max().item() for a, b in zip(pa, pb))
    worst = max(((ga[n] - gb[n]).norm().item() / (gb[n].norm().item() + 1e-12), n) for n in gb if not n.startswith("fnet."))
    worst_f = max(((ga[n] - gb[n]).norm().item() / (gb[n].norm().item() + 1e-12), n) for n in gb if n.startswith("fnet."))
    ok = dp < 1e-4 and worst[0] < 5e-3 and set(ga) == set(gb)
    print(f"{'ok  ' if ok else 'FAIL'} {name:40s} pred diff {dp:.2e}  grads (rest) {worst[0]:.2e} {worst[1]}  (fnet) {worst_f[0]:.2e}")
    return ok


ns = lambda alt=False: argparse.Namespace(small=False, mixed_precision=False, alternate_corr=alt)


def l2l_inputs(B, offs, H=128, W=192, h=96, w=128):
    c1, c2 = torch.rand(B, 3, H, W, device=dev) * 255, torch.rand(B, 3, H, W, device=dev) * 255
    ox, oy = offs
    if isinstance(ox, int):
        i1, i2 = c1[:, :, oy:oy + h, ox:ox + w].contiguous(), c2[:, :, oy:oy + h, ox:ox + w].contiguous()
    else:
        i1 = torch.cat([c1[i:i + 1, :, oy[i]:oy[i] + h, ox[i]:ox[i] + w] for i in range(B)]).contiguous()
        i2 = torch.cat([c2[i:i + 1, :, oy[i]:oy[i] + h, ox[i]:ox[i] + w] for i in range(B)]).contiguous()
    return i1, i2, c1, c2, ox, oy


if __name__ == "__main__":
    ok = True
    for B, H, W, it in ((1, 64, 96, 1), (3, 72, 104, 2), (2, 128, 192, 3), (1, 136, 200, 5)):
        im1, im2 = torch.rand(B, 3, H, W, device=dev) * 255, torch.rand(B, 3, H, W, device=dev) * 255
        ok &= compare(f"raft B={B} {H}x{W} iters={it}", lambda: RAFT(ns()), lambda m: m(im1, im2, iters=it))
        if H >= 128:      # (AlternateCorrBlock pools once more than it has levels, like the reference's: 16 x 16 feature maps at least)
            ok &= compare(f"raft alt B={B} {H}x{W} iters={it}", lambda: RAFT(ns(True)), lambda m: m(im1, im2, iters=it))
    for B, it, offs in ((1, 2, (8, 16)), (2, 4, ([8, 24], [16, 0])), (2, 5, ([0, 24], [16, 8]))):
        i1, i2, c1, c2, ox, oy = l2l_inputs(B, offs)
        ok &= compare(f"l2l B={B} iters={it} offsets={offs}", lambda: L2L(ns()), lambda m: m(i1, i2, c1, c2, ox, oy, iters=it))
        if B == 2:
            # batches + sup_grad_samples against the plain per-iteration forward (no promise used) under a loss that keeps the promise
            ok &= compare(f"l2l B={B} iters={it} sup_grad_samples=1", lambda: L2L(ns()),
                          lambda m: m(i1, i2, c1, c2, ox, oy, iters=it, sup_grad_samples=1), sup_k=1,
                          ref_call=lambda m: m(i1, i2, c1, c2, ox, oy, iters=it))
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)
