"""Rank sharding of the training set: how the data-parallel path partitions its units.

Behaviour of /root/reference/traiNNer/data/data_sampler.py:9-55, pinned by tests/golden/sampler.pt:
the epoch seeds a permutation of `ceil(len * ratio / world) * world` slots, slots map to samples modulo the
dataset length, and rank r takes slots r, r + world, r + 2*world, ...  No collective is involved: every rank derives
its own index stream from (epoch, rank, world).
"""
from __future__ import annotations

import math
from collections.abc import Iterator, Sized

import torch
from torch.utils.data.sampler import Sampler


class EnlargedSampler(Sampler):
    def __init__(self, dataset: Sized, num_replicas: int, rank: int | None, ratio: int = 1) -> None:
        if not isinstance(dataset, Sized):
            raise AssertionError("Unable to determine length of dataset")
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.epoch = 0
        self.num_samples = math.ceil(len(dataset) * ratio / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self) -> Iterator[int]:
        gen = torch.Generator()
        gen.manual_seed(self.epoch)
        slots = torch.randperm(self.total_size, generator=gen)
        mine = slots[self.rank:self.total_size:self.num_replicas] % len(self.dataset)
        assert mine.numel() == self.num_samples
        return iter(mine.tolist())
