"""`build_model(opt)` -- /root/reference/traiNNer/models/__init__.py:16: OTF configs select RealESRGANModel
(its `feed_data` is SURVEY.md section 8f-1, a later row); everything else is `SRModel`."""
from __future__ import annotations

from .sr_model import SRModel

__all__ = ["SRModel", "build_model"]


def build_model(opt) -> SRModel:
    if getattr(opt, "high_order_degradation", None):
        raise NotImplementedError("RealESRGANModel (OTF degradation feed) is not built yet; use paired data")
    return SRModel(opt)
