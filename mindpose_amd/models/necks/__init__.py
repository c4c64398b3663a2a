from .neck import Neck  # noqa: F401
