from .adamw import SGD, Adagrad, Adam, AdamWeightDecay, Momentum, split_decay  # noqa: F401
from .grad_allreduce import GradientAverager  # noqa: F401
from .ckpt import load_checkpoint, load_param_into_net, save_checkpoint  # noqa: F401
from .graph_step import GraphedTrainStep  # noqa: F401
from .loss_scale import DynamicLossScaleManager  # noqa: F401
from .lr import WarmupCosineDecayLR, WarmupMultiStepDecayLR, create_lr_scheduler  # noqa: F401
from .optim_factory import create_optimizer  # noqa: F401
from .sharding import shard_range  # noqa: F401
