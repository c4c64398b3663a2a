"""Component registry with the reference's surface (mindpose/register.py:12-59).

``register(module_name, extra_name="")`` stores a callable under ``fn.__name__`` and, when given,
under ``extra_name``; ``entrypoint(module_name, component_name)`` resolves it and raises
``ValueError`` listing what is supported on a miss; a duplicate registration only warns.
"""
import logging
from typing import Any, Callable, Dict, List

_REGISTRY: Dict[str, Dict[str, Callable[..., Any]]] = {}


def _add(module_name: str, name: str, fn: Callable[..., Any]) -> None:
    table = _REGISTRY.setdefault(module_name, {})
    if name in table:
        logging.warning(f"`{name}` is already registered")
    table[name] = fn


def register(module_name: str, extra_name: str = "") -> Callable[..., Any]:
    def wrapper(fn: Callable[..., Any]) -> Callable[..., Any]:
        _add(module_name, fn.__name__, fn)
        if extra_name:
            _add(module_name, extra_name, fn)
        return fn

    return wrapper


def list_components(module: str) -> List[str]:
    return sorted(_REGISTRY.get(module, {}))


def list_modules() -> List[str]:
    return sorted(_REGISTRY)


def entrypoint(module_name: str, component_name: str) -> Callable[..., Any]:
    if module_name not in _REGISTRY:
        raise ValueError(f"Unkown module `{module_name}`. Supported modules: {list_modules()}")
    if component_name not in _REGISTRY[module_name]:
        raise ValueError(
            f"Unkown components `{component_name}`. "
            f"Supported componetns in `{module_name}`: {list_components(module_name)}")
    return _REGISTRY[module_name][component_name]
