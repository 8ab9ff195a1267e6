"""RealESRGANModel -- SRModel whose training `feed_data` synthesises the LQ batch on the device from GT + blur kernels
(/root/reference/traiNNer/models/realesrgan_model.py:52-93,456-664; selected by `high_order_degradation: true`, models/__init__.py:16).

DataFeed contract (utils/types.py:10-17, data/realesrgan_dataset.py:213-219): training batches carry `gt`, `kernel1`, `kernel2`,
`sinc_kernel`; validation batches carry `lq` (and `gt`) like the paired model.  The stage list, its order and every RNG draw live in
`data/otf.OTFDegrader`; everything after `feed_data` (optimize_parameters, EMA, checkpoints, validation) is inherited unchanged.
Unlike the reference (hard-coded `.cuda()` on a DiffJPEG module it never calls, :81-83) nothing here is constructed that is not used.
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

    def feed_data(self, data: dict) -> None:
        if self.is_train and "kernel1" in data:
            assert "gt" in data and "kernel1" in data and "kernel2" in data and "sinc_kernel" in data
            gt = self._to_device_f32(data["gt"])
            self.kernel1 = data["kernel1"].to(self.device, non_blocking=True)
            self.kernel2 = data["kernel2"].to(self.device, non_blocking=True)
            self.sinc_kernel = data["sinc_kernel"].to(self.device, non_blocking=True)
            self.lq, self.gt = self.otf(gt, self.kernel1, self.kernel2, self.sinc_kernel)
        else:
            super().feed_data(data)     # paired training or validation
