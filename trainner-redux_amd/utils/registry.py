"""Name -> factory registries: the plugin surface of the reference, kept call-compatible.

Behaviour mirrored from /root/reference/traiNNer/utils/registry.py:
  * `@REG.register()` / `REG.register(obj)`; key = `obj.__name__.lower()` or an explicit `name`,
    optionally with `_<suffix>` appended (:42-84); a duplicate key is an AssertionError (:46-49);
  * `REG.get(name)` lower-cases the name, falls back to `name + "_traiNNer"`, raises KeyError (:86-95);
  * `name in REG` is case-insensitive; iteration yields (key, obj) pairs; `.keys()`.
Singletons: the same eight registries the reference defines (:107-114).
"""
from __future__ import annotations

from collections.abc import Callable, Iterator


class Registry:
    def __init__(self, name: str) -> None:
        self._name = name
        self._table: dict[str, Callable] = {}

    @property
    def name(self) -> str:
        return self._name

    def _insert(self, key: str, obj: Callable, suffix: str | None) -> None:
        if isinstance(suffix, str):
            key = f"{key}_{suffix}"
        assert key not in self._table, f"An object named '{key}' was already registered in '{self._name}' registry!"
        self._table[key] = obj

    def register(self, obj: Callable | None = None, suffix: str | None = None, name: str | None = None):
        if obj is not None:  # plain call form
            self._insert(name or obj.__name__.lower(), obj, suffix)
            return None

        def decorator(target: Callable) -> Callable:
            self._insert(name or target.__name__.lower(), target, suffix)
            return target

        return decorator

    def get(self, name: str, suffix: str = "traiNNer") -> Callable:
        key = name.lower()
        for candidate in (key, f"{key}_{suffix}"):
            found = self._table.get(candidate)
            if found is not None:
                return found
        raise KeyError(f"No object named '{key}' found in '{self._name}' registry!")

    def __contains__(self, name: str) -> bool:
        return name.lower() in self._table

    def __iter__(self) -> Iterator[tuple[str, Callable]]:
        return iter(self._table.items())

    def keys(self):
        return self._table.keys()


DATASET_REGISTRY = Registry("dataset")
ARCH_REGISTRY = Registry("arch")
SPANDREL_REGISTRY = Registry("spandrel")
TESTARCH_REGISTRY = Registry("testarch")
MODEL_REGISTRY = Registry("model")
LOSS_REGISTRY = Registry("loss")
METRIC_REGISTRY = Registry("metric")
OPTIMIZER_REGISTRY = Registry("optimizer")
