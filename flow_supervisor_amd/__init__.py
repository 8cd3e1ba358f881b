"""flow_supervisor_amd: the RAFT hot path of iwbn/flow-supervisor rebuilt for MI355X (gfx950).

Layout mirrors the reference's PyTorch tree so it drops in:
    flow_supervisor_amd.core.corr      CorrBlock, AlternateCorrBlock     (pytorch/core/corr.py)
    flow_supervisor_amd.core.update    BasicUpdateBlock, SmallUpdateBlock (pytorch/core/update.py)
    flow_supervisor_amd.core.raft      RAFT (+ upsample_flow)             (pytorch/core/raft.py)
    flow_supervisor_amd.core.utils.utils  coords_grid, bilinear_sampler, upflow8, InputPadder
    flow_supervisor_amd.alt_cuda_corr  forward / backward                 (pytorch/alt_cuda_corr)
The compute lives in csrc/*.hip behind the C ABI of include/fsraft.h (libfsraft.so).
"""
