"""Helpers with the reference's names and semantics (pytorch/core/utils/utils.py)."""
import torch
import torch.nn.functional as F

from ... import ops
from ..._lib import on_tensor_device


class InputPadder:
    """Replicate-pads images so both sides divide by 8 (utils.py:7-24)."""

    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = dims[-2:]
        ph = (((self.ht // 8) + 1) * 8 - self.ht) % 8
        pw = (((self.wd // 8) + 1) * 8 - self.wd) % 8
        top = ph // 2 if mode == "sintel" else 0
        self._pad = [pw // 2, pw - pw // 2, top, ph - top]

    def pad(self, *inputs):
        return [F.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        l, r, t, b = self._pad
        return x[..., t:ht - b, l:wd - r]


def forward_interpolate(flow):
    """Warm start for the next frame (utils.py:26-54; evaluate.py:43 `forward_interpolate(flow_low[0])[None].cuda()`):
    flow [2,H,W] -> [2,H,W], every vector carried to where it points and the grid filled from the nearest landed point.
    The reference round-trips through the host (scipy griddata); this runs fsraft_forward_interpolate on the device and
    returns a device tensor (the caller's `.cuda()` is then a no-op).  There is no host implementation here."""
    from ... import _lib as L
    if flow.dim() != 3 or flow.shape[0] != 2:
        raise ValueError(f"forward_interpolate expects [2,H,W], got {tuple(flow.shape)}")
    if not flow.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("forward_interpolate runs on the HIP device only (no GPU visible)")
        flow = flow.cuda()
    flow = flow.detach().float().contiguous()
    out = torch.empty_like(flow)
    L.check(L.load().fsraft_forward_interpolate(L.ptr(flow), L.ptr(out), flow.shape[1], flow.shape[2], L.stream()),
            "forward_interpolate")
    return out


def coords_grid(batch, ht, wd, device=None):
    """[B,2,ht,wd] float, channel 0 = x, channel 1 = y (utils.py:74-77)."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device, dtype=torch.float32),
                            torch.arange(wd, device=device, dtype=torch.float32), indexing="ij")
    return torch.stack([xs, ys], dim=0)[None].repeat(batch, 1, 1, 1)


@on_tensor_device
def bilinear_sampler(img, coords, mode="bilinear", mask=False):
    """grid_sample in pixel coordinates, align_corners=True, zero padding (utils.py:57-71).
    Kept as a framework op: the hot loop does not call it (CorrBlock uses the fused HIP lookup)."""
    H, W = img.shape[-2:]
    xgrid, ygrid = coords.split([1, 1], dim=-1)
    grid = torch.cat([2 * xgrid / (W - 1) - 1, 2 * ygrid / (H - 1) - 1], dim=-1)
    out = F.grid_sample(img, grid, align_corners=True)
    if mask:
        valid = (grid[..., :1] > -1) & (grid[..., 1:] > -1) & (grid[..., :1] < 1) & (grid[..., 1:] < 1)
        return out, valid.float()
    return out


class _UpFlow8(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow):
        ctx.hw = flow.shape[-2:]
        return ops.upflow8_fwd(flow)

    @staticmethod
    def backward(ctx, g):
        return ops.upflow8_bwd(g, *ctx.hw)


@on_tensor_device
def upflow8(flow, mode="bilinear"):
    """8 * bilinear x8 upsampling with align_corners=True (utils.py:80-82) on the HIP kernel."""
    if mode != "bilinear":
        return 8 * F.interpolate(flow, size=(8 * flow.shape[2], 8 * flow.shape[3]), mode=mode, align_corners=True)
    return _UpFlow8.apply(flow)


_MIXED_WARNED = [False]


def warn_mixed_precision(args):
    """`args.mixed_precision` (pytorch/core/raft.py:99-127, train.py:232: autocast around the encoders and the update block) is
    accepted for signature compatibility and has NO effect here: the models do not enter autocast -- under it the encoders would
    leave the fsraft kernels for the framework's half-precision convolutions, which are SLOWER on this stack than the fp32 path
    (bench.py's value_north_star_encoders) -- and every kernel of the path stores and accumulates in fp32 (GEMM products as three fp16 products of
    scaled operands, <= 2^-22 each, or exact fp32: fsraft_set_arithmetic): at least the precision the flag would give, at full speed.  Said once, loudly."""
    if getattr(args, "mixed_precision", False) and not _MIXED_WARNED[0]:
        import warnings
        _MIXED_WARNED[0] = True
        warnings.warn("flow_supervisor_amd: args.mixed_precision=True has no effect -- the models do not enter autocast and the "
                      "fsraft kernels compute in fp32 (fp16x3 products of scaled operands, <= 2^-22 each, or exact-fp32 products); results are those of mixed_precision=False",
                      stacklevel=3)
