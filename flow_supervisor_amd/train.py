"""The training step the benchmark times: forward (12 GRU iterations), sequence loss, backward,
gradient all-reduce, clip, AdamW.  Callers of the hot path, written with framework ops."""
import torch

from .parallel import FlatGradients


def sequence_loss(flow_preds, flow_gt=None, gamma=0.8):
    """sum_i gamma^(n-1-i) * mean(sqrt((pred_i - gt)^2 + 1e-6)): the Charbonnier sequence loss of
    pytorch/train.py:60-96 with every pixel valid (gt defaults to zero flow)."""
    n = len(flow_preds)
    loss = 0.0
    for i, p in enumerate(flow_preds):
        d = p if flow_gt is None else p - flow_gt
        loss = loss + (gamma ** (n - i - 1)) * torch.sqrt(d * d + 1e-6).mean()
    return loss


class TrainStep:
    def __init__(self, model, lr=4e-4, wdecay=1e-5, eps=1e-8, clip=1.0, iters=12, capturable=False):
        self.model = model
        self.iters = iters
        self.clip = clip
        self.grads = FlatGradients(model.parameters())
        fused = self.grads.flat.is_cuda
        self.opt = torch.optim.AdamW(self.grads.params, lr=lr, weight_decay=wdecay, eps=eps, fused=fused,
                                     capturable=bool(capturable and fused))

    def __call__(self, image1, image2, flow_gt=None):
        self.grads.zero_()
        preds = self.model(image1, image2, iters=self.iters)
        loss = sequence_loss(preds, flow_gt)
        loss.backward()
        self.grads.all_reduce_mean_()
        self.grads.clip_norm_(self.clip)
        self.opt.step()
        return loss.detach()
