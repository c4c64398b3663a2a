from .transform import *  # noqa: F401, F403
