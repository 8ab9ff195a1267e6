"""RealESRGANModel -- SRModel whose training `feed_data` synthesises the LQ batch on the device from GT + blur kernels
(/root/reference/traiNNer/models/realesrgan_model.py:52-93,456-664; selected by `high_order_degradation: true`, models/__init__.py:16).

DataFeed contract (utils/types.py:10-17, data/realesrgan_dataset.py:213-219): training batches carry `gt`, `kernel1`, `kernel2`,
`sinc_kernel`; validation batches carry `lq` (and `gt`) like the paired model.  The stage list, its order and every RNG draw live in
`data/otf.OTFDegrader`; everything after `feed_data` (optimize_parameters, EMA, checkpoints, validation) is inherited unchanged.
Unlike the reference (hard-coded `.cuda()` on a DiffJPEG module it never calls, :81-83) nothing here is constructed that is not used.

Overlap.  The compression stage is a host codec round trip (as in the reference), i.e. a device -> host -> device hop in the middle of
the feed.  On the training stream that hop waits for everything queued before it -- the whole previous step -- and the GPU then idles while
the host runs the codec and re-fills the launch queue: configs[2] (ESRGAN + VGG + UNet-D, batch 8, 64 -> 256; 52.8 ms of GPU work per
step) measured 81 ms/step that way (62 ms/step now, 55 with `use_compile`: side stream + batch-parallel codec + no hidden syncs in the step).  The feed therefore runs on its own high-priority HIP stream: its kernels and copies interleave with
the previous step still executing on the training stream, the host waits only for that stream's few small kernels, and the training stream
just waits on an event before it reads the new pair.  `prefetch(data)` additionally splits the feed so that the device half of the NEXT batch
is issued before the current step (the codec then finds its pixels already in pinned memory).  Same thread, same RNG draw order.
(A worker thread was measured slower: it fights the launch-bound training thread for the interpreter lock.)
"""
from __future__ import annotations

import torch

from ..data.otf import OTFDegrader
from .sr_model import SRModel


class RealESRGANModel(SRModel):
    def __init__(self, opt) -> None:
        super().__init__(opt)
        self.otf = OTFDegrader(opt)
        self.queue_size = self.otf.pool.queue_size
        self.kernel1 = self.kernel2 = self.sinc_kernel = None
        self._otf_pending = None        # (id(data), state of OTFDegrader.start, kernels)
        self._otf_stream = None

    def _side(self):
        if self._otf_stream is None:
            self._otf_stream = torch.cuda.Stream(device=self.device, priority=-1)
        return torch.cuda.stream(self._otf_stream)

    def _start(self, data: dict):
        with self._side():
            gt = self._to_device_f32(data["gt"])
            ks = tuple(data[k].to(self.device, non_blocking=True) for k in ("kernel1", "kernel2", "sinc_kernel"))
            return self.otf.start(gt, *ks), ks

    def prefetch(self, data: dict) -> None:
        """Run the device half of a coming training batch's feed now; `feed_data(data)` (same object) finishes it."""
        if not (self.is_train and "kernel1" in data) or self._otf_pending is not None:
            return
        self._otf_pending = (id(data), *self._start(data))

    def feed_data(self, data: dict) -> None:
        if self.is_train and "kernel1" in data:
            assert "gt" in data and "kernel1" in data and "kernel2" in data and "sinc_kernel" in data
            pend, self._otf_pending = self._otf_pending, None
            if pend is not None and pend[0] != id(data):
                raise RuntimeError("feed_data received a different batch than the one handed to prefetch()")
            st, ks = pend[1:] if pend is not None else self._start(data)
            with self._side():
                lq, gt = self.otf.finish(st)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self._otf_stream)
            for t in (lq, gt, *ks):
                t.record_stream(cur)          # allocated on the feed stream, read on the training stream
            self.lq, self.gt = lq, gt
            self.kernel1, self.kernel2, self.sinc_kernel = ks
        else:
            super().feed_data(data)     # paired training or validation
