"""Registries with the reference's names (utils/registry.py:13-94, utils/enums.py:33-79) so that
`ModelRegistry.get("video_encoder")`, `LossRegistry.get("contrastive")`, `RunnerRegistry.get("DeepCORO_clip")`
resolve to the HIP-backed classes of this package."""
from __future__ import annotations

import importlib
import pkgutil
from typing import Any, Dict, Optional, Type


class BaseRegistry:
    _registry: Dict[str, Type[Any]] = {}
    _registry_type: str = "base"

    @classmethod
    def register(cls, name: str):
        def decorator(class_type: Type[Any]) -> Type[Any]:
            cls._registry[str(name)] = class_type
            return class_type

        return decorator

    @classmethod
    def get(cls, name: str) -> Optional[Type[Any]]:
        name = str(name)
        if name not in cls._registry:
            raise ValueError(
                f"{cls._registry_type} {name} not found in registry {cls._registry_type}. "
                f"Available {cls._registry_type}s: {list(cls._registry.keys())}"
            )
        return cls._registry[name]

    @classmethod
    def list_registered(cls) -> Dict[str, Type[Any]]:
        return cls._registry.copy()

    @classmethod
    def create(cls, name: str, **kwargs):
        return cls.get(name)(**kwargs)


class RunnerRegistry(BaseRegistry):
    _registry: Dict[str, Type[Any]] = {}
    _registry_type = "runner"


class ModelRegistry(BaseRegistry):
    _registry: Dict[str, Type[Any]] = {}
    _registry_type = "model"


class LossRegistry(BaseRegistry):
    _registry: Dict[str, Type[Any]] = {}
    _registry_type = "loss"


def _adopt_reference_registries() -> None:
    """Inside the reference tree (its `utils.registry` importable), register into the reference's OWN singletons so
    that `projects/contrastive_pretraining_project.py` resolves the HIP classes without any other change."""
    global RunnerRegistry, ModelRegistry, LossRegistry
    try:
        ref = importlib.import_module("utils.registry")
        trio = (ref.RunnerRegistry, ref.ModelRegistry, ref.LossRegistry)
        if all(hasattr(r, "register") and hasattr(r, "get") and isinstance(getattr(r, "_registry", None), dict) for r in trio):
            RunnerRegistry, ModelRegistry, LossRegistry = trio
    except Exception:  # not inside the reference tree: keep the identical local registries
        pass


_adopt_reference_registries()


def register_submodules(package_name: str = "deepcoro_clip_amd", recursive: bool = True) -> None:
    """Import every submodule so that the @register decorators run (utils/registry.py:82-94)."""
    package = importlib.import_module(package_name)
    for _, name, is_pkg in pkgutil.walk_packages(package.__path__, package.__name__ + "."):
        if name.endswith(".build") or ".libdcc_hip" in name or ".csrc" in name:
            continue  # the build script and the C-ABI shared object are not Python modules to register
        importlib.import_module(name)
