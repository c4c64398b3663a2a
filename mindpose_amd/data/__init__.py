from .data_factory import *  # noqa: F401, F403
from .dataset import *  # noqa: F401, F403
from .transform import *  # noqa: F401, F403
