"""Feed side of the hot path: rank-sharding sampler, paired dataset, pinned double-buffer feed."""
from .data_sampler import EnlargedSampler
from .paired_image_dataset import PairedImageDataset
from .prefetch import DevicePrefetcher, build_dataloader
from .realesrgan_dataset import RealESRGANDataset

__all__ = ["EnlargedSampler", "PairedImageDataset", "RealESRGANDataset", "DevicePrefetcher", "build_dataloader"]
