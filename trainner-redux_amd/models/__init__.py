"""`build_model(opt)` -- /root/reference/traiNNer/models/__init__.py:16: OTF configs (`high_order_degradation: true`) select
RealESRGANModel, whose `feed_data` runs the degradation stages on the device; everything else is `SRModel`."""
from __future__ import annotations

from .realesrgan_model import RealESRGANModel
from .sr_model import SRModel

__all__ = ["SRModel", "RealESRGANModel", "build_model"]


def build_model(opt) -> SRModel:
    if getattr(opt, "high_order_degradation", None):
        return RealESRGANModel(opt)
    return SRModel(opt)
