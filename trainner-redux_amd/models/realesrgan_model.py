"""RealESRGANModel -- SRModel whose training `feed_data` synthesises the LQ batch on the device from GT + blur kernels
(/root/reference/traiNNer/models/realesrgan_model.py:52-93,456-664; selected by `high_order_degradation: true`, models/__init__.py:16).

DataFeed contract (utils/types.py:10-17, data/realesrgan_dataset.py:213-219): training batches carry `gt`, `kernel1`, `kernel2`,
`sinc_kernel`; validation batches carry `lq` (and `gt`) like the paired model.  The stage list, its order and every RNG draw live in
`data/otf.OTFDegrader`; everything after `feed_data` (optimize_parameters, EMA, checkpoints, validation) is inherited unchanged.
Unlike the reference (hard-coded `.cuda()` on a DiffJPEG module it never calls, :81-83) nothing here is constructed that is not used.

Overlap.  The compression stage is a host codec round trip (as in the reference), i.e. a device -> host -> device hop in the middle of
the feed.  `prefetch(data)` runs only the FIRST half of a coming batch's feed (device stages + an async copy to pinned memory + an event) and
the following `feed_data(data)` with the same batch object the second half (codec on the already-copied pixels, 8-bit rounding, crop,
pool), so the hop waits on an event recorded a whole training step earlier.  Same thread, same stream, same RNG draw order.
Measured (tools/time_otf_step.py, configs[2]: ESRGAN + VGG + UNet-D at batch 8 / 64 -> 256): that configuration is bound by the HOST --
52.8 ms of GPU work per step against 58 ms to enqueue its ~1,600 launches from Python -- so the feed's 7-14 ms of codec time adds to the
step whichever way it is scheduled (68 ms unsplit, 73 ms split; a worker thread on a side stream: 86 ms, it fights the training thread
for the interpreter lock).  The split pays only where the step is GPU-bound; the cure for the launch-bound tiles is a recorded launch plan /
hipGraph of forward_hip + backward_hip, not scheduling of the feed.
"""
from __future__ import annotations

from ..data.otf import OTFDegrader
from .sr_model import SRModel


class RealESRGANModel(SRModel):
    def __init__(self, opt) -> None:
        super().__init__(opt)
        self.otf = OTFDegrader(opt)
        self.queue_size = self.otf.pool.queue_size
        self.kernel1 = self.kernel2 = self.sinc_kernel = None
        self._otf_pending = None        # (id(data), state of OTFDegrader.start, kernels)

    def _start(self, data: dict):
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
            self.lq, self.gt = self.otf.finish(st)
            self.kernel1, self.kernel2, self.sinc_kernel = ks
        else:
            super().feed_data(data)     # paired training or validation
