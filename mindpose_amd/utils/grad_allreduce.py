"""Data-parallel gradient mean over RCCL/xGMI (reference: ``set_auto_parallel_context(parallel_mode="data_parallel",
gradients_mean=True)``, tools/train.py:43-49 - MindSpore all-reduces every gradient each step).

MI355X-first shape of the same exchange: all gradients live in ONE flat fp32 arena (``p.grad`` are views into it), the
arena is cut into a few large buckets in reverse parameter order (the order backward produces them), and each bucket's
all-reduce is launched asynchronously the moment its last gradient has been accumulated, so communication overlaps the
rest of backward.  HRNet-W32 = 114 MB of fp32 gradients per step = 4 buckets of 32 MB.  BatchNorm statistics stay per device
(no SyncBN in the reference).

Where the mean's 1/world goes (``mean``):
  "collective"  ``ReduceOp.AVG`` inside RCCL (backend "nccl"): no extra pass over the arena
  "consumer"    the collective SUMs and the consumer folds ``mean_scale`` (= 1/world) into its own read of the arena - what the
                arena optimizers do (``mp_adamw_step_scaled`` multiplies it into the loss-scale factor)
  "pass"        SUM, then one ``arena *= 1/world`` pass (backend "gloo" has no AVG; CPU tests)

Transport (``transport``): "torch" = ``torch.distributed`` (backend "nccl" IS RCCL on ROCm; "gloo" runs the identical logic on
CPU), "native" = the library's own communicator (``mp_comm_init_rank`` / ``mp_allreduce_grads``, include/mindpose_hip.h) on a
dedicated HIP stream; its unique id travels through the ``torch.distributed`` store.
"""
import ctypes
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class NativeComm:
    """``mp_comm_*``: an RCCL communicator owned by libmindpose_hip.so, one per process group.

    Ordering against ``torch.distributed``'s own communicator (EvalCallback's loss all-reduce, ``broadcast_object_list``): a native
    collective is enqueued on this object's stream AFTER everything on the caller's current stream (``wait_stream``), and its
    consumer waits for the returned event on the current stream; torch's collectives order themselves against the current stream the
    same way.  Two collectives of the two communicators are therefore never in flight together as long as each is waited for
    (synchronous torch calls, ``GradientAverager.finish()``) before the next is issued - what every caller in this package does."""

    def __init__(self, device: torch.device, process_group=None) -> None:
        from .. import _lib
        self.lib = _lib.load()
        if not self.lib.mp_comm_available():
            raise _lib.MindposeHipError("no RCCL found in this process (mp_comm_available() == 0)")
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        ident = (ctypes.c_char * 128)()
        if self.rank == 0:
            _lib.check(self.lib.mp_comm_get_unique_id(ident), "mp_comm_get_unique_id")
        if self.world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                       group=process_group)
            ident = (ctypes.c_char * 128).from_buffer_copy(box[0])
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self.lib.mp_comm_init_rank(ctypes.byref(self.handle), self.world, ident, self.rank), "mp_comm_init_rank")
        self.stream = torch.cuda.Stream(device=device)
        self._check = _lib.check

    def all_reduce(self, view: torch.Tensor, average: bool, split: bool = False) -> torch.cuda.Event:
        """In-place all-reduce of ``view`` on the communicator's stream, ordered after everything already enqueued on the
        caller's current stream; returns the event the consumer waits for.  ``split``: reduce-scatter + all-gather."""
        self.stream.wait_stream(torch.cuda.current_stream(view.device))
        if split and view.numel() % self.world == 0:
            rc = self.lib.mp_reduce_scatter_allgather_grads(self.handle, view.data_ptr(), view.numel(), self.world, self.rank,
                                                            int(average), self.stream.cuda_stream)
        else:
            rc = self.lib.mp_allreduce_grads(self.handle, view.data_ptr(), view.numel(), int(average), self.stream.cuda_stream)
        self._check(rc, "mp_allreduce_grads")
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def count(self) -> int:
        """ncclCommCount of the library's communicator."""
        return int(self.lib.mp_comm_count(self.handle))

    def close(self) -> None:
        """ncclCommDestroy (after the communicator's stream has drained); idempotent."""
        if self.handle:
            try:
                self.stream.synchronize()
            except Exception:
                pass
            self.lib.mp_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _EventHandle:
    def __init__(self, ev: torch.cuda.Event) -> None:
        self.ev = ev

    def wait(self) -> None:
        torch.cuda.current_stream().wait_event(self.ev)


class GradientAverager:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 32.0, process_group=None,
                 overlap: bool = True, arena_order: str = "reverse", mean: Optional[str] = None,
                 transport: Optional[str] = None, force_collectives: bool = False) -> None:
        """``arena_order``: "reverse" lays the arena out last-parameter-first (buckets fill front to back during
        backward); "given" keeps the caller's order, e.g. to alias an optimizer's flat parameter layout (buckets then
        complete back to front - the overlap is the same, every bucket still launches when its last gradient lands).
        ``force_collectives``: issue the collectives on a one-rank group too (exercises the RCCL path on a one-GPU box)."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dtype = self.params[0].device, torch.float32
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        backend = dist.get_backend(process_group) if dist.is_initialized() else None
        transport = transport or os.environ.get("MINDPOSE_DP_TRANSPORT", "torch")
        if transport not in ("torch", "native"):
            raise ValueError("transport must be 'torch' or 'native'")
        if transport == "native" and dev.type != "cuda":
            raise ValueError("the native RCCL transport needs the arena on a GPU")
        self.native = NativeComm(dev, process_group) if (transport == "native" and dist.is_initialized()) else None
        has_avg = self.native is not None or backend == "nccl"
        if mean is None:
            mean = "collective" if has_avg else "pass"
        if mean not in ("collective", "consumer", "pass"):
            raise ValueError("mean must be 'collective', 'consumer' or 'pass'")
        if mean == "collective" and not has_avg:
            raise ValueError(f"backend {backend!r} has no averaging all-reduce: use mean='pass' or 'consumer'")
        self.mean = mean
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
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        self.overlap = overlap and self.active
        if self.overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
        for p in self.params:
            # without overlap hooks nothing needs to observe the accumulation: backward kernels may add into the arena slot
            # themselves (train_ops._direct_grad) instead of returning a tensor for autograd's AccumulateGrad to add
            p._mp_grad_direct = not self.overlap
        self.begin_step()

    def comm_ranks(self) -> int:
        """Ranks of the communicator the collectives run on - asked of the communicator (ncclCommCount for the native transport,
        the process group's size for torch.distributed), not of the launcher's arguments."""
        if self.native is not None:
            return self.native.count()
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def mean_scale(self) -> float:
        """Factor the consumer of the arena still has to apply after ``finish()`` (1 unless mean == "consumer")."""
        return 1.0 / self.world if (self.mean == "consumer" and self.world > 1) else 1.0

    # -- per step ---------------------------------------------------------------------------------
    def begin_step(self) -> None:
        """Zero the arena and re-arm the buckets (call before forward/backward)."""
        if self.arena.is_cuda:
            from ..models.train_ops import drop_wgrad_jobs
            # leftovers of a backward pass that raised must not reach this step's arena (THIS arena's slots only)
            drop_wgrad_jobs(self.arena.data_ptr(), self.arena.data_ptr() + self.arena.numel() * self.arena.element_size())
        self.arena.zero_()
        self.rearm()

    def rearm(self) -> None:
        """Re-arm the buckets without touching the arena (a captured step zeroes it inside its hipGraph)."""
        for b in self.buckets:
            b["pending"] = b["count"]
            b["launched"] = False
        self._handles = []

    def launch_bucket(self, index: int) -> None:
        """Start the all-reduce of bucket ``index`` now (a segmented step does this for the buckets whose gradients are complete
        while the rest of the backward pass still runs); ``finish()`` launches whatever was not."""
        b = self.buckets[index]
        if self.active and not b.get("launched"):
            self._launch(b)

    def _launch(self, b: dict) -> None:
        b["launched"] = True
        view = self.arena[b["start"]:b["end"]]
        if self.native is not None:
            self._handles.append(_EventHandle(self.native.all_reduce(view, average=self.mean == "collective")))
            return
        op = dist.ReduceOp.AVG if self.mean == "collective" else dist.ReduceOp.SUM
        self._handles.append(dist.all_reduce(view, op=op, group=self.group, async_op=True))

    def _on_grad_ready(self, p: torch.nn.Parameter) -> None:
        if p.grad is None or p.grad.data_ptr() < self.arena.data_ptr():
            raise RuntimeError("gradient left the arena (p.grad was replaced); keep set_to_none=False semantics")
        b = self.buckets[self._bucket_of[id(p)]]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def finish(self) -> None:
        """Wait for the bucket all-reduces (launching any that did not overlap); afterwards the arena holds the mean over ranks,
        up to ``mean_scale``."""
        if self.active:
            if not self.overlap:
                for b in self.buckets:
                    if not b.get("launched"):
                        self._launch(b)
            else:
                for b in self.buckets:
                    if b["pending"] > 0:  # a parameter received no gradient this step
                        self._launch(b)
                        b["pending"] = 0
            for h in self._handles:
                h.wait()
            if self.mean == "pass" and self.world > 1:
                self.arena.mul_(1.0 / self.world)
        self._handles = []

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def close(self) -> None:
        """Tear-down: pending collectives are waited for, hooks removed, the library's own communicator destroyed
        (``ncclCommDestroy``) - the owner (the arena optimizers' ``close()``) calls this before the process group goes away."""
        for h in self._handles:
            h.wait()
        self._handles = []
        self.remove_hooks()
        if self.native is not None:
            self.native.close()
            self.native = None
            self.active = False

    def __del__(self):
        try:
            if getattr(self, "native", None) is not None:
                self.native.close()
        except Exception:
            pass
