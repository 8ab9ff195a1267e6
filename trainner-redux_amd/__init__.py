"""MI355X-native (gfx950) engine for the traiNNer-redux `SRModel.optimize_parameters` hot path.

Host side mirrors the reference's plugin surface (`ARCH_REGISTRY` / `LOSS_REGISTRY` /
`OPTIMIZER_REGISTRY`, `build_network` / `build_loss` / `build_optimizer`, `SRModel`) --
reference: traiNNer/utils/registry.py, traiNNer/archs/__init__.py:29-64,
traiNNer/losses/__init__.py:55-119, traiNNer/models/sr_model.py.  All per-pixel work is in
`csrc/` (hand-written HIP for CDNA4) behind the C ABI declared in include/trainner_hip.h and
loaded with ctypes (`_lib.py`).  There is no CPU fallback: ops raise if the library is missing.
"""
__version__ = "0.1.0"
