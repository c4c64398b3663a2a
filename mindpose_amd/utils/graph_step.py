"""The whole training step - forward, loss, backward - as ONE hipGraph (MI355X-first: "capture launch-bound inner loops in
hipGraphs").  An HRNet-W32 step is ~3000 kernel launches issued from Python autograd; under amp O2 the kernels are so short
that the host cannot keep the GPU fed (the eager step takes ~64 ms at ANY batch size up to 192).  Shapes are static, so the
launch sequence is captured once and replayed with one call; the optimizer update (and, under data parallelism, the bucket
all-reduces) run after the replay on the gradient arena the graph filled.

Reference: the loop that ``mindspore.Model.train(..., dataset_sink_mode=True)`` runs on the device (tools/train.py:233) -
MindSpore's graph mode compiles the step once as well.

Data parallelism (tools/train.py:43-49, SURVEY 7 item 10 "bucketed all-reduce overlapped with backward"): with ``segments > 1`` the
backward pass is captured as SEVERAL hipGraphs cut at the backbone's stage boundaries (HRNet: stage 4 -> 3 -> 2 -> stem).  After
each replay the gradient buckets that are complete - every parameter of the bucket belongs to a finished segment and has no
grouped weight gradient still queued - are handed to the all-reduce, which then runs on the communication stream under the next
segment's kernels; only the last buckets are exposed.  The segments are the single backward pass cut in pieces: same node order,
same weight-gradient groups (they are carried across the cuts), so the gradient arena equals the one-graph step's bit for bit.
"""
import os
import time
from typing import List, Optional, Sequence

import torch


def plan_bucket_schedule(grads, seg_params, queued_after=None) -> List[List[int]]:
    """Which gradient buckets may leave for the all-reduce after each segment of a segmented backward pass: a bucket is complete once
    every parameter in it belongs to a finished segment (``seg_params[i]`` = the parameters segment i completes, in run order) and
    none of its arena slots is still waited on (``queued_after(i)`` = data_ptr set of slots with deferred work after segment i, e.g.
    grouped weight gradients carried across the cut).  Every bucket appears exactly once; what no earlier segment released goes with
    the last."""
    members = {}
    for p in grads.params:
        members.setdefault(grads._bucket_of[id(p)], []).append(p)
    done, scheduled, schedule = set(), set(), []
    for i, ps in enumerate(seg_params):
        last = i == len(seg_params) - 1
        done |= {id(p) for p in ps}
        queued = set() if (last or queued_after is None) else set(queued_after(i))
        ready = []
        for bi in range(len(grads.buckets)):
            if bi in scheduled:
                continue
            if last or all(id(p) in done and p.grad.data_ptr() not in queued for p in members.get(bi, [])):
                ready.append(bi)
                scheduled.add(bi)
        schedule.append(ready)
    return schedule


def run_segments(grads, runners, schedule, exchange: bool = True) -> float:
    """One backward pass as its segments (``runners[i]()`` = a graph replay, or any callable that completes segment i's gradients in
    the arena), the finished buckets handed to the all-reduce in between; returns the host seconds spent issuing them.  The caller's
    ``grads.finish()`` (the optimizer's ``step``) waits for them and launches whatever was not released."""
    grads.rearm()
    t_issue = 0.0
    for run, ready in zip(runners, schedule):
        run()
        if ready and grads.active and exchange:
            t0 = time.perf_counter()
            for bi in ready:
                grads.launch_bucket(bi)
            t_issue += time.perf_counter() - t0
    return t_issue


class GraphedTrainStep:
    def __init__(self, net_with_loss: torch.nn.Module, optimizer, example_inputs: Sequence[torch.Tensor],
                 loss_scale_manager=None, warmup: int = 3, segments: Optional[int] = None) -> None:
        """``optimizer``: an ``AdamWeightDecay`` built with ``overlap=False`` (its gradient arena is what the graph writes);
        ``example_inputs``: CUDA tensors with the step's static shapes (data, label, extra inputs of ``NetWithLoss``).
        The ``warmup`` eager forward/backward passes that precede the capture (autotuner, kernel attributes, allocator) do
        not update parameters, only the BatchNorm moving statistics.
        ``segments``: None = 4 when the gradient all-reduce is active (several ranks) and the backbone offers ``train_segments()``,
        else 1 (one graph, every bucket reduced after it); an explicit value is honoured where the backbone allows it."""
        if optimizer.grads.overlap:
            raise ValueError("build the optimizer with overlap=False for a graphed step: the bucket all-reduces are issued between "
                             "the captured segments / after the replay")
        self.nwl, self.opt, self.mgr = net_with_loss, optimizer, loss_scale_manager
        self.static_in = [t.detach().clone() for t in example_inputs]
        dev = self.static_in[0].device
        self.scale_t = torch.ones((), device=dev, dtype=torch.float32)
        if self.mgr is not None:
            self.scale_t.fill_(self.mgr.loss_scale)
        backbone = getattr(getattr(net_with_loss, "net", None), "backbone", None)
        groups = backbone.train_segments() if hasattr(backbone, "train_segments") else None
        if segments is None:
            segments = len(groups) if (groups and optimizer.grads.active) else 1
        self.segments = max(1, min(int(segments), len(groups))) if groups else 1
        # inside the graph the branches of an HRModule run on side streams (fork / join = graph dependencies): +5 % on the step
        from ..models.backbones.hrnet import quiet_accumulate_grad_stream_warning, set_branch_streams
        prev_branch_streams = set_branch_streams(True)
        with quiet_accumulate_grad_stream_warning():  # (only the warm-up passes and the capture run the autograd engine)
            self._warm_up_and_capture(net_with_loss, optimizer, backbone, groups, dev, warmup)
        set_branch_streams(prev_branch_streams)
        self._planned = [m for m in net_with_loss.modules() if hasattr(m, "_plans")]  # walked once, not per step

    def _warm_up_and_capture(self, net_with_loss, optimizer, backbone, groups, dev, warmup) -> None:
        from ..models.train_ops import flush_wgrad_jobs
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                optimizer.grads.begin_step()
                loss = net_with_loss(*self.static_in)
                (loss * self.scale_t).backward()
                flush_wgrad_jobs()
                del loss  # drop the autograd graph before the next pass / the capture
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.issue_ms: List[float] = []  # host time spent handing buckets to the all-reduce between the segments, per step
        if self.segments == 1:
            self.graph = torch.cuda.CUDAGraph()
            dot = os.environ.get("MINDPOSE_GRAPH_DEBUG_DOT")  # kernel work: the captured graph's nodes and edges as a DOT file
            if dot:
                self.graph.enable_debug_mode()
            with torch.cuda.graph(self.graph):
                optimizer.grads.arena.zero_()
                self.static_loss = net_with_loss(*self.static_in)
                (self.static_loss * self.scale_t).backward()
                flush_wgrad_jobs()  # the remainders of the grouped weight gradients belong to the captured step
            self.graphs, self.bucket_schedule = [self.graph], [[]]
            if dot:
                self.graph.debug_dump(dot)
        else:
            self._capture_segments(backbone, groups, dev)

    def _capture_segments(self, backbone, groups, dev) -> None:
        """Forward + loss + the backward pass of the LAST module group in the first graph, one graph per earlier group after it.
        The backbone cuts its autograd graph at the stage boundaries while this forward records them (detached leaves): every
        segment is its own autograd graph, run to its leaves by one ``torch.autograd.grad`` call whose boundary gradients seed the
        next.  (``backward(inputs=[non-leaf])`` is no alternative: the engine EXECUTES the boundary tensor's own producer node to
        capture its gradient - that node's arena side effects would happen twice.)"""
        from ..models.backbones.hrnet import join_branch_streams
        from ..models.train_ops import flush_wgrad_jobs, pending_wgrad_slots, set_wgrad_autoflush
        grads = self.opt.grads
        n_groups = len(groups)
        # merge the trailing groups when fewer segments were asked for: segment i covers groups[lo_i : hi_i] (backward order)
        bounds = [round(i * n_groups / self.segments) for i in range(self.segments + 1)]
        head = getattr(self.nwl.net, "head", None)
        arena_params = {id(p) for p in grads.params}

        def params_of(mods):
            out = []
            for m in mods:
                out += [p for p in m.parameters() if id(p) in arena_params]
            return out

        seg_params = []
        for i in range(self.segments):
            mods = [m for g in groups[bounds[i]:bounds[i + 1]] for m in g]
            if i == 0 and head is not None:
                mods = [head] + mods
            seg_params.append(params_of(mods))
        covered = {id(p) for ps in seg_params for p in ps}
        seg_params[-1] += [p for p in grads.params if id(p) not in covered]  # anything the groups do not name completes last
        self.graphs = []
        queued_at = []  # arena slots with a grouped weight gradient still queued when segment i's capture ended
        prev_auto = set_wgrad_autoflush(False)
        pool = None
        try:
            cuts_all: List = []  # per stage boundary (forward order): (tensors the graph was cut at, the detached leaves behind them)
            seeds, seed_grads = None, None
            for i in range(self.segments):
                g = torch.cuda.CUDAGraph()
                last = i == self.segments - 1
                with torch.cuda.graph(g, pool=pool):
                    if i == 0:
                        grads.arena.zero_()
                        # the backbone cuts ONLY the boundaries between segments: groups merged into one segment stay one autograd
                        # graph (a cut inside a segment would end its autograd.grad call at the inner leaves)
                        backbone._train_cut_sink = cuts_all
                        backbone._train_cut_at = {n_groups - 1 - bounds[j + 1] for j in range(self.segments - 1)}
                        try:
                            self.static_loss = self.nwl(*self.static_in)
                        finally:
                            backbone._train_cut_sink = None
                            backbone._train_cut_at = None
                        assert len(cuts_all) == self.segments - 1, (len(cuts_all), self.segments)
                        seeds, seed_grads = [self.static_loss * self.scale_t], [None]
                    # boundary in front of this segment's modules: cuts_all holds the cut boundaries in FORWARD order, the segments run
                    # in backward order
                    roots, leaves = ([], []) if last else cuts_all[len(cuts_all) - 1 - i]
                    # autograd.grad (not backward): the boundary gradients come back as the very tensors the consumers' data-gradient
                    # launches wrote (AccumulateGrad would clone them - the BatchNorm hand-over checks that identity); weight
                    # gradients are side effects of the nodes (direct arena slots), anything returned instead is added here
                    got = torch.autograd.grad(seeds, list(leaves) + seg_params[i], grad_outputs=seed_grads, allow_unused=True)
                    for p_, g_ in zip(seg_params[i], got[len(leaves):]):
                        if g_ is not None:
                            p_.grad.add_(g_)
                    join_branch_streams(dev)  # boundary gradients are produced on the branch / row streams
                    if last:
                        flush_wgrad_jobs()  # the remainders of the grouped weight gradients belong to the captured step
                    if not last:
                        pairs = [(r, g_) for r, g_ in zip(roots, got[:len(leaves)]) if g_ is not None]
                        seeds, seed_grads = [r for r, _ in pairs], [g_ for _, g_ in pairs]
                pool = pool or g.pool()
                self.graphs.append(g)
                queued_at.append(set() if last else pending_wgrad_slots())
            self.bucket_schedule = plan_bucket_schedule(grads, seg_params, lambda i: queued_at[i])
        finally:
            set_wgrad_autoflush(prev_auto)
        self.graph = self.graphs[0]

    def replay(self, exchange: bool = True) -> None:
        """The captured step without the update: every segment, the finished buckets handed to the all-reduce in between
        (``exchange=False``: no bucket is launched - the arena holds this rank's own gradients afterwards)."""
        t_issue = run_segments(self.opt.grads, [g.replay for g in self.graphs], self.bucket_schedule, exchange)  # (the first graph zeroes the arena)
        self.issue_ms.append(t_issue * 1e3)
        if len(self.issue_ms) > 4096:
            del self.issue_ms[:2048]

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        """Copy the batch into the static buffers, replay the graph(s), run the optimizer; returns the (static) loss tensor."""
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        if self.mgr is not None:
            self.scale_t.fill_(self.mgr.loss_scale)
        self.replay()  # under DP opt.step() waits for the bucket all-reduces (and launches those no segment boundary released)
        self.updated = self.opt.step(loss_scale_manager=self.mgr)
        if self.updated:
            # a replay never passes through PlannedModule.forward, which is what drops recorded inference plans in training
            # mode: without this an evaluation between graphed steps would replay packed weights / folded BatchNorm of an
            # earlier parameter state
            for m in self._planned:
                m._plans.clear()
        return self.static_loss
