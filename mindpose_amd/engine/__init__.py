from .evaluator import *  # noqa: F401, F403
from .factory import create_evaluator, create_inferencer  # noqa: F401
from .inferencer import *  # noqa: F401, F403
