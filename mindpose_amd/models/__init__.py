from . import backbones, decoders, heads, loss, necks  # noqa: F401
from .layers import auto_mixed_precision, share_tuner_choices, tune_on_rank0_first  # noqa: F401
from .model_factory import *  # noqa: F401, F403
from .networks import EvalNet, Net, NetWithLoss  # noqa: F401
from .synthetic import init_synthetic  # noqa: F401
