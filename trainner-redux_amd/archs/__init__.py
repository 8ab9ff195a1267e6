"""`build_network` -- same lookup order and call convention as the reference
(/root/reference/traiNNer/archs/__init__.py:17-64): every `*_arch.py` in this folder is imported at
package import so its `@SPANDREL_REGISTRY.register()` / `@ARCH_REGISTRY.register()` decorators run,
and `build_network({"type": name, **kwargs})` tries SPANDREL -> ARCH -> TESTARCH.
"""
from __future__ import annotations

import importlib
import os
from copy import deepcopy
from typing import Any

from torch import nn

from ..utils.registry import ARCH_REGISTRY, SPANDREL_REGISTRY, TESTARCH_REGISTRY

__all__ = ["build_network"]

_here = os.path.dirname(os.path.abspath(__file__))
_arch_modules = [
    importlib.import_module(f"{__name__}.{os.path.splitext(f)[0]}") for f in sorted(os.listdir(_here)) if f.endswith("_arch.py")
]


def build_network(opt: dict[str, Any]) -> nn.Module:
    opt = deepcopy(opt)
    network_type = opt.pop("type")
    for reg in (SPANDREL_REGISTRY, ARCH_REGISTRY):
        if network_type in reg:
            return reg.get(network_type)(**opt)
    return TESTARCH_REGISTRY.get(network_type)(**opt)
