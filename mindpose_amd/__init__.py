"""mindpose_amd - MI355X-native top-down heat-map pose hot path behind mindpose's plugin surface.

Mirrors ``mindpose/__init__.py``: the registry, the model factories and the hot-path data/engine
components are importable from the package root.
"""
from . import register as _register  # noqa: F401
from .data import *  # noqa: F401, F403
from .engine import *  # noqa: F401, F403
from .models import *  # noqa: F401, F403
from .register import entrypoint, list_components, list_modules, register  # noqa: F401

__version__ = "0.1.0"
