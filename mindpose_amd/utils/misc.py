"""Small training helpers with the reference's names (mindpose/utils/misc.py:7-36): the 1-scalar all-reduce used for the epoch
loss and the running-average meter.  torch tensors (CPU or CUDA) instead of MindSpore ones; the collective is
``torch.distributed`` (RCCL on GPUs, gloo on CPU)."""
from typing import Union

import torch
import torch.distributed as dist


class Allreduce:
    """Sum of a tensor over all ranks, every rank gets the result (misc.py:7-16); the identity without a process group."""

    def __init__(self, process_group=None) -> None:
        self.group = process_group

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return x
        out = x.detach().clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out

    construct = __call__


class AverageMeter:
    """val / sum / count / avg of a stream of values (misc.py:19-36).  Values may be scalars or small tensors (a loss with
    several items).  Like the reference's meter, the running sum lives where the values live: a CUDA loss is accumulated ON THE
    DEVICE in fp64 (two tiny asynchronous kernels per update, no read-back), so a per-step `update` never stalls the host behind
    the graphed / overlapped training step; the one synchronising read happens when somebody looks at `avg` (once per epoch)."""

    def __init__(self) -> None:
        self.reset()

    def reset(self) -> None:
        self.val = torch.zeros((), dtype=torch.float64)
        self.avg = torch.zeros((), dtype=torch.float64)
        self.sum = torch.zeros((), dtype=torch.float64)
        self.count = 0.0

    def update(self, val: Union[float, torch.Tensor], n: int = 1) -> None:
        v = torch.as_tensor(val).detach().to(torch.float64)  # stays on the value's device
        self.val = v
        self.sum = (self.sum.to(v.device) if self.sum.device != v.device else self.sum) + v * n
        self.count += n
        self.avg = self.sum / self.count
