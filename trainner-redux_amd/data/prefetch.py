"""Pinned-host double-buffered feed (reference: CUDAPrefetcher, traiNNer/data/prefetch_dataloader.py:418-499).

The loader side produces uint8 HWC batches in pinned memory; `DevicePrefetcher` keeps ONE batch in flight on a side stream
(hipMemcpyAsync H2D, then `tr_u8hwc_to_nchw` on the same stream -> fp32 NCHW [0,1]), so the copy and the conversion of batch i+1 overlap
the training step of batch i.  `next()` makes the compute stream wait on the batch's event and hands out device tensors that satisfy
the reference's DataFeed contract (what `SRModel.feed_data` expects).  fp32 CHW batches (the reference's own datasets) pass through
the same double buffer without the conversion.
"""
from __future__ import annotations

import torch

from .. import ops


class DevicePrefetcher:
    def __init__(self, loader, device, mean=None, std=None) -> None:
        self.loader, self.device, self.mean, self.std = loader, torch.device(device), mean, std
        self.stream = torch.cuda.Stream(device=self.device)
        self._it = None
        self._next = None
        self.reset()

    def reset(self) -> None:
        self._it = iter(self.loader)
        self._preload()

    def _stage(self, t: torch.Tensor) -> torch.Tensor:
        if not t.is_pinned():
            t = t.pin_memory()
        d = t.to(self.device, non_blocking=True)
        if d.dtype == torch.uint8:                          # (N, H, W, C) patches -> the trainer's fp32 NCHW
            out = torch.empty(d.shape[0], d.shape[3], d.shape[1], d.shape[2], dtype=torch.float32, device=self.device)
            ops.u8hwc_to_nchw(d.contiguous(), out, self.mean, self.std)
            d.record_stream(self.stream)
            return out
        return d

    def _preload(self) -> None:
        try:
            batch = next(self._it)
        except StopIteration:
            self._next = None
            return
        with torch.cuda.stream(self.stream):
            staged = {k: (self._stage(v) if torch.is_tensor(v) else v) for k, v in batch.items()}
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._next = (staged, ev)

    def next(self):
        """-> the next DataFeed on the device (None at the end of the epoch); starts staging the one after."""
        if self._next is None:
            return None
        staged, ev = self._next
        torch.cuda.current_stream(self.device).wait_event(ev)
        for v in staged.values():
            if torch.is_tensor(v):
                v.record_stream(torch.cuda.current_stream(self.device))
        self._preload()
        return staged


def worker_seed(seed: int, num_workers: int, rank: int, worker_id: int) -> int:
    """data/__init__.py:172-178 (`worker_init_fn`): `num_workers * rank + worker_id + seed`, so that worker k of every rank draws its own
    crop / flip / rotation stream."""
    return num_workers * rank + worker_id + seed


def build_dataloader(dataset, batch_size: int, num_workers: int = 0, sampler=None, seed: int | None = None, prefetch_factor: int = 2,
                     rank: int | None = None):
    """data/__init__.py:61-178 in one place: pinned batches, the reference's worker seeding (`worker_seed`), drop_last for training.
    With `num_workers == 0` the loading happens in this process, which is seeded as worker 0 of its rank."""
    import random

    import numpy as np

    from ..utils.dist_util import get_dist_info

    if rank is None:
        rank, _ = get_dist_info()

    def init(worker_id):
        if seed is not None:
            s = worker_seed(seed, max(num_workers, 1), rank, worker_id)
            random.seed(s)
            np.random.seed(s)

    if not num_workers:
        init(0)
    kw = dict(batch_size=batch_size, shuffle=sampler is None, sampler=sampler, num_workers=num_workers, drop_last=True, pin_memory=True,
              worker_init_fn=init if num_workers else None)
    if num_workers:
        kw.update(prefetch_factor=prefetch_factor, persistent_workers=True)
    return torch.utils.data.DataLoader(dataset, **kw)
