"""Feed side of the hot path.  Round 1 ships the rank-sharding sampler; the paired
dataset / pinned double-buffer loader are the next rows."""
from .data_sampler import EnlargedSampler

__all__ = ["EnlargedSampler"]
