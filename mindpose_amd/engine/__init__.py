from .inferencer import *  # noqa: F401, F403
