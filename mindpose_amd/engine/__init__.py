from .evaluator import *  # noqa: F401, F403
from .inferencer import *  # noqa: F401, F403
