"""Data-parallel gradient mean over RCCL/xGMI (reference: ``set_auto_parallel_context(parallel_mode="data_parallel",
gradients_mean=True)``, tools/train.py:43-49 - MindSpore all-reduces every gradient each step).

MI355X-first shape of the same exchange: all gradients live in ONE flat fp32 arena (``p.grad`` are views into it), the
arena is cut into a few large buckets in reverse parameter order (the order backward produces them), and each bucket's
all-reduce is launched asynchronously the moment its last gradient has been accumulated, so communication overlaps the
rest of backward.  HRNet-W32 = 114 MB of fp32 gradients per step = 4 buckets of 32 MB.  ``torch.distributed`` backend
"nccl" is RCCL on ROCm; "gloo" runs the identical logic on CPU (tests).  BatchNorm statistics stay per device
(no SyncBN in the reference).
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradientAverager:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 32.0, process_group=None,
                 overlap: bool = True, arena_order: str = "reverse") -> None:
        """``arena_order``: "reverse" lays the arena out last-parameter-first (buckets fill front to back during
        backward); "given" keeps the caller's order, e.g. to alias an optimizer's flat parameter layout (buckets then
        complete back to front - the overlap is the same, every bucket still launches when its last gradient lands)."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dtype = self.params[0].device, torch.float32
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        total = sum(p.numel() for p in self.params)
        self.arena = torch.zeros(total, device=dev, dtype=dtype)
        if arena_order not in ("reverse", "given"):
            raise ValueError("arena_order must be 'reverse' or 'given'")
        order = list(reversed(self.params)) if arena_order == "reverse" else list(self.params)
        limit = max(1, int(bucket_mb * 1024 * 1024 / 4))
        self.buckets: List[dict] = []
        off = 0
        cur = dict(start=0, end=0, pending=0, count=0)
        self._bucket_of = {}
        for p in order:
            n = p.numel()
            if cur["count"] > 0 and (cur["end"] - cur["start"]) + n > limit:
                self.buckets.append(cur)
                cur = dict(start=off, end=off, pending=0, count=0)
            p.grad = self.arena[off:off + n].view_as(p)
            self._bucket_of[id(p)] = len(self.buckets)
            cur["end"] = off + n
            cur["count"] += 1
            off += n
        self.buckets.append(cur)
        self._handles: List = []
        self._hooks = []
        self.overlap = overlap and self.world > 1
        if self.overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
        for p in self.params:
            # without overlap hooks nothing needs to observe the accumulation: backward kernels may add into the arena slot
            # themselves (train_ops._direct_grad) instead of returning a tensor for autograd's AccumulateGrad to add
            p._mp_grad_direct = not self.overlap
        self.begin_step()

    # -- per step ---------------------------------------------------------------------------------
    def begin_step(self) -> None:
        """Zero the arena and re-arm the buckets (call before forward/backward)."""
        self.arena.zero_()
        for b in self.buckets:
            b["pending"] = b["count"]
        self._handles = []

    def _launch(self, b: dict) -> None:
        view = self.arena[b["start"]:b["end"]]
        self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad_ready(self, p: torch.nn.Parameter) -> None:
        if p.grad is None or p.grad.data_ptr() < self.arena.data_ptr():
            raise RuntimeError("gradient left the arena (p.grad was replaced); keep set_to_none=False semantics")
        b = self.buckets[self._bucket_of[id(p)]]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def finish(self) -> None:
        """Wait for the bucket all-reduces (launching any that did not overlap) and turn sums into means."""
        if self.world > 1:
            if not self.overlap:
                for b in self.buckets:
                    self._launch(b)
            else:
                for b in self.buckets:
                    if b["pending"] > 0:  # a parameter received no gradient this step
                        self._launch(b)
                        b["pending"] = 0
            for h in self._handles:
                h.wait()
            self.arena.mul_(1.0 / self.world)
        self._handles = []

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
