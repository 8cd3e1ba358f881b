"""pytorch/core/gma_corr.py surface.  gma_corr.CorrBlock (:15-63) is the same all-pairs volume + pyramid + lookup as
core/corr.py; the reference's CorrBlockSingleScale (:66-103) is never constructed anywhere in the reference and
is the one-level special case of the same kernels."""
from .corr import AlternateCorrBlock, CorrBlock  # noqa: F401


class CorrBlockSingleScale(CorrBlock):
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        super().__init__(fmap1, fmap2, num_levels=1, radius=radius)
