"""Module-level drop-in for the reference's pybind extension ``alt_cuda_corr``
(pytorch/alt_cuda_corr/correlation.cpp:51-54): ``forward`` and ``backward`` with the same
arguments, list-of-tensors return values and RuntimeError on non-CUDA / non-contiguous inputs.

    import flow_supervisor_amd.alt_cuda_corr as alt_cuda_corr
    corr, = alt_cuda_corr.forward(fmap1, fmap2, coords, radius)
"""
from . import ops
from ._lib import on_tensor_device


@on_tensor_device
def forward(fmap1, fmap2, coords, radius):
    return [ops.altcorr_fwd(fmap1, fmap2, coords, int(radius))]


@on_tensor_device
def backward(fmap1, fmap2, coords, corr_grad, radius):
    return list(ops.altcorr_bwd(fmap1, fmap2, coords, corr_grad, int(radius)))
