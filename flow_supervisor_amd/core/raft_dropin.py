"""The INTEGRATION.md section 1 route as a runnable model: the REFERENCE's model shell -- its loop, its tensor layouts, its calls --
around this package's swapped blocks, nothing else of this package's own shell (`core/raft.py::RAFT.forward`: the flow-carrying,
channels-last loop with the once-per-step head / motion-encoder batches and the second stream).

What a maintainer of iwbn/flow-supervisor gets by changing three imports in `pytorch/core/raft.py` (`from corr import CorrBlock,
AlternateCorrBlock`, `from update import BasicUpdateBlock, SmallUpdateBlock`, `from extractor import BasicEncoder, SmallEncoder`)
and nothing else: per iteration `corr_fn(coords1)` returns NCHW `[B, 324, H/8, W/8]`, `update_block(net, inp, corr, flow)` takes
and returns NCHW tensors (`net`, `up_mask [B, 576, H/8, W/8]`, `delta_flow`), `upsample_flow(coords1 - coords0, up_mask)` runs
every iteration, `coords1` is carried and detached exactly as in pytorch/core/raft.py:99-144.  The loop below restates those
lines; `bench.py --variant dropin` measures it so that the drop-in route has a throughput number next to the headline's
(VERDICT r4 next #6); `tests/test_gpu_end_to_end.py::test_reference_shaped_shell_matches_the_package_shell` holds its outputs and
gradients to the package shell's.
"""
import torch

from .._lib import on_tensor_device
from .corr import AlternateCorrBlock, CorrBlock
from .raft import RAFT
from .utils.utils import coords_grid, upflow8


class ReferenceShapedRAFT(RAFT):
    """Same parameters, same state_dict keys as `RAFT` (and as the reference's); only `forward` differs."""

    def initialize_flow(self, img):
        """coords0 = coords1 = the pixel grid at 1/8 resolution (pytorch/core/raft.py:63-70)."""
        N, _, H, W = img.shape
        return coords_grid(N, H // 8, W // 8, device=img.device), coords_grid(N, H // 8, W // 8, device=img.device)

    @on_tensor_device
    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False):
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        fmap1, fmap2 = self.fnet([image1, image2])                       # raft.py:98-100
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        block = AlternateCorrBlock if self.args.alternate_corr else CorrBlock
        corr_fn = block(fmap1, fmap2, radius=self.args.corr_radius)     # raft.py:104-107
        cnet = self.cnet(image1)                                         # raft.py:110-114
        net, inp = torch.split(cnet, [self.hidden_dim, self.context_dim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        coords0, coords1 = self.initialize_flow(image1)
        if flow_init is not None:
            coords1 = coords1 + flow_init
        predictions = []
        flow_up = None
        for _ in range(iters):                                           # raft.py:121-139
            coords1 = coords1.detach()
            corr = corr_fn(coords1)                                      # NCHW [B, 324, H/8, W/8], contiguous
            net, up_mask, delta_flow = self.update_block(net, inp, corr, coords1 - coords0)
            coords1 = coords1 + delta_flow
            flow_up = upflow8(coords1 - coords0) if up_mask is None else self.upsample_flow(coords1 - coords0, up_mask)
            predictions.append(flow_up)
        if test_mode:
            return coords1 - coords0, flow_up
        return predictions
