"""The whole training step - forward, loss, backward - as ONE hipGraph (MI355X-first: "capture launch-bound inner loops in
hipGraphs").  An HRNet-W32 step is ~3000 kernel launches issued from Python autograd; under amp O2 the kernels are so short
that the host cannot keep the GPU fed (the eager step takes ~64 ms at ANY batch size up to 192).  Shapes are static, so the
launch sequence is captured once and replayed with one call; the optimizer update (and, under data parallelism, the bucket
all-reduces) run after the replay on the gradient arena the graph filled.

Reference: the loop that ``mindspore.Model.train(..., dataset_sink_mode=True)`` runs on the device (tools/train.py:233) -
MindSpore's graph mode compiles the step once as well.
"""
from typing import Optional, Sequence

import torch


class GraphedTrainStep:
    def __init__(self, net_with_loss: torch.nn.Module, optimizer, example_inputs: Sequence[torch.Tensor],
                 loss_scale_manager=None, warmup: int = 3) -> None:
        """``optimizer``: an ``AdamWeightDecay`` built with ``overlap=False`` (its gradient arena is what the graph writes);
        ``example_inputs``: CUDA tensors with the step's static shapes (data, label, extra inputs of ``NetWithLoss``).
        The ``warmup`` eager forward/backward passes that precede the capture (autotuner, kernel attributes, allocator) do
        not update parameters, only the BatchNorm moving statistics."""
        if optimizer.grads.overlap:
            raise ValueError("build the optimizer with overlap=False for a graphed step: the bucket all-reduces run after the replay")
        self.nwl, self.opt, self.mgr = net_with_loss, optimizer, loss_scale_manager
        self.static_in = [t.detach().clone() for t in example_inputs]
        dev = self.static_in[0].device
        self.scale_t = torch.ones((), device=dev, dtype=torch.float32)
        if self.mgr is not None:
            self.scale_t.fill_(self.mgr.loss_scale)
        # inside the graph the branches of an HRModule run on side streams (fork / join = graph dependencies): +5 % on the step
        from ..models.backbones.hrnet import set_branch_streams
        prev_branch_streams = set_branch_streams(True)
        # ... and the weight gradients (leaves of the backward pass) on a side stream per branch stream, joined after backward
        from ..models.train_ops import flush_wgrad_jobs, join_wgrad_lanes, set_wgrad_lanes
        prev_wgrad_lanes = set_wgrad_lanes(True)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                optimizer.grads.begin_step()
                loss = net_with_loss(*self.static_in)
                (loss * self.scale_t).backward()
                flush_wgrad_jobs()
                join_wgrad_lanes(dev)
                del loss  # drop the autograd graph before the next pass / the capture
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            optimizer.grads.arena.zero_()
            self.static_loss = net_with_loss(*self.static_in)
            (self.static_loss * self.scale_t).backward()
            flush_wgrad_jobs()  # the remainders of the grouped weight gradients belong to the captured step
            join_wgrad_lanes(dev)
        set_branch_streams(prev_branch_streams)
        set_wgrad_lanes(prev_wgrad_lanes)
        self._planned = [m for m in net_with_loss.modules() if hasattr(m, "_plans")]  # walked once, not per step

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        """Copy the batch into the static buffers, replay the graph, run the optimizer; returns the (static) loss tensor."""
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        if self.mgr is not None:
            self.scale_t.fill_(self.mgr.loss_scale)
        self.opt.grads.rearm()  # the graph zeroes and fills the arena; under DP opt.step() launches the bucket all-reduces after it
        self.graph.replay()
        self.updated = self.opt.step(loss_scale_manager=self.mgr)
        if self.updated:
            # a replay never passes through PlannedModule.forward, which is what drops recorded inference plans in training
            # mode: without this an evaluation between graphed steps would replay packed weights / folded BatchNorm of an
            # earlier parameter state
            for m in self._planned:
                m._plans.clear()
        return self.static_loss
