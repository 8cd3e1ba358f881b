"""NHWC-shaped twins of the reference's TensorFlow hot-path API (API parity only):

    raft.corr.CorrBlock(num_levels=4, radius=4, is_max_disp=False).__call__(corr_pyramid, coords)   raft/corr.py:5-22
    raft.allfield.calc_all_field(a, b, num_pool)                                                     raft/allfield.py:61-92
    raft.allfield.build_pyramid(c_volume, num_pool)                                                  raft/allfield.py:94-106
    transpose_volume(c_volume)  == tf.transpose(c_volume, [0, 3, 4, 1, 2])                           raft/semi.py:250, 257
    raft.upsample.UpsampleConvexWithMask(scale=8).call([x, mask, ref])                               raft/upsample.py:4-41
    raft.smurf_models.raft_update.BasicUpdateBlock(args, hidden_dim).call([net, inp, corr, flow])    raft_update.py:180-212

TensorFlow is not installed here, so these take and return torch tensors laid out the way the TF code lays
them out (channels last, coords as [B,H,W,2] with (x, y) in the last dim) and run the same HIP kernels as the
PyTorch-shaped API.  Pooling: the fused build pools with PyTorch's floor halving; TF pools level 0 with padding='SAME'
(ceil sizes, partial edge windows averaged over their in-range elements: e.g. 55 rows at 1/8 of Sintel give 28 / 14 / 7 rows,
not 27 / 13 / 6).  When a pooled dimension is odd, calc_all_field and build_pyramid therefore build the pyramid with
fsraft_corr_pool_pyramid_same and CorrBlock looks it up with fsraft_corr_lookup_fwd_same.  No TF oracle exists in this
environment: the SAME path is tested against a restatement of TF's documented semantics (parity unpinned).
"""
from .api import (BasicUpdateBlock, CorrBlock, UpsampleConvexWithMask, build_pyramid, calc_all_field,  # noqa: F401
                  transpose_volume)
