from .backbone import Backbone  # noqa: F401
from .hrnet import *  # noqa: F401, F403
from .resnet import *  # noqa: F401, F403
