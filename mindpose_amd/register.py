"""Component registry with the reference's surface (mindpose/register.py:12-59).

``register(module_name, extra_name="")`` decorates a class / function and files it under its ``__name__`` and, when given,
under ``extra_name``; ``entrypoint(module_name, component_name)`` resolves it and raises ``ValueError`` listing what is
supported on a miss (messages kept, typos included: callers may match on them); a duplicate registration only warns.
"""
import logging
from typing import Any, Callable, Dict, List


class _Registry:
    """section -> {component name -> callable}"""

    def __init__(self) -> None:
        self.sections: Dict[str, Dict[str, Callable[..., Any]]] = {}

    def put(self, section: str, names, fn: Callable[..., Any]) -> None:
        table = self.sections.setdefault(section, {})
        for name in names:
            if name in table:
                logging.warning(f"`{name}` is already registered")
            table[name] = fn

    def get(self, section: str, name: str) -> Callable[..., Any]:
        table = self.sections.get(section)
        if table is None:
            raise ValueError(f"Unkown module `{section}`. Supported modules: {sorted(self.sections)}")
        try:
            return table[name]
        except KeyError:
            raise ValueError(f"Unkown components `{name}`. Supported componetns in `{section}`: {sorted(table)}") from None


_REG = _Registry()


def register(module_name: str, extra_name: str = "") -> Callable[..., Any]:
    def decorate(fn: Callable[..., Any]) -> Callable[..., Any]:
        _REG.put(module_name, [fn.__name__] + ([extra_name] if extra_name else []), fn)
        return fn

    return decorate


def entrypoint(module_name: str, component_name: str) -> Callable[..., Any]:
    return _REG.get(module_name, component_name)


def list_modules() -> List[str]:
    return sorted(_REG.sections)


def list_components(module: str) -> List[str]:
    return sorted(_REG.sections.get(module, {}))
