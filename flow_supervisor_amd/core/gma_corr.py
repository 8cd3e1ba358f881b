"""pytorch/core/gma_corr.py surface.  gma_corr.CorrBlock (:15-63) is the same all-pairs volume + pyramid + lookup as
core/corr.py; CorrBlockSingleScale (:66-103, never constructed anywhere in the reference) is the one-level case of the
same kernels: one (2r+1)^2 lookup of the full-resolution volume."""
from .corr import AlternateCorrBlock, CorrBlock  # noqa: F401


class CorrBlockSingleScale(CorrBlock):
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        super().__init__(fmap1, fmap2, num_levels=1, radius=radius)
