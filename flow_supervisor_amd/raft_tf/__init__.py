"""NHWC-shaped twins of the reference's TensorFlow hot-path API (API parity only):

    raft.corr.CorrBlock(num_levels=4, radius=4, is_max_disp=False).__call__(corr_pyramid, coords)   raft/corr.py:5-22
    raft.allfield.calc_all_field(a, b, num_pool)                                                     raft/allfield.py:61-92
    raft.upsample.UpsampleConvexWithMask(scale=8).call([x, mask, ref])                               raft/upsample.py:4-41
    raft.smurf_models.raft_update.BasicUpdateBlock(args, hidden_dim).call([net, inp, corr, flow])    raft_update.py:180-212

TensorFlow is not installed here, so these take and return torch tensors laid out the way the TF code lays
them out (channels last, coords as [B,H,W,2] with (x, y) in the last dim) and run the same HIP kernels as the
PyTorch-shaped API.  PyTorch pooling semantics are normative (floor halving); TF's SAME/ceil pooling differs
only when H/8 or W/8 is odd and is unpinned (no TF oracle in this environment).
"""
from .api import BasicUpdateBlock, CorrBlock, UpsampleConvexWithMask, calc_all_field  # noqa: F401
