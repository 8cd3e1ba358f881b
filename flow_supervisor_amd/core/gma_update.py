"""pytorch/core/gma_update.py surface: the GMA update block and the (shared) RAFT sub-module containers."""
from .update import (BasicMotionEncoder, BasicUpdateBlock, ConvGRU, FlowHead, GMAUpdateBlock,  # noqa: F401
                     SepConvGRU, SmallMotionEncoder, SmallUpdateBlock)
