"""Import shim: the package directory is `trainner-redux_amd/` (a hyphen is not importable),
so `import trainner_redux_amd` loads it from there."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trainner-redux_amd")
_spec = importlib.util.spec_from_file_location(
    "trainner_redux_amd", os.path.join(_root, "__init__.py"), submodule_search_locations=[_root]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["trainner_redux_amd"] = _mod
_spec.loader.exec_module(_mod)
