from .loss import Loss  # noqa: F401
from .mse import JointsMSELoss  # noqa: F401
