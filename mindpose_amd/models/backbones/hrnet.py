"""HRNet backbone on the MI355X HIP path.

Same graph, constructor arguments, registry names and parameter names as the reference
(mindpose/models/backbones/hrnet.py:348-718); every conv+BN(+add)(+ReLU) group is one launch of the
direct fp32-MFMA convolution, and the HRModule fuse rows (:318-344) accumulate in place through the
conv epilogue (nearest up-sampling happens while storing; nothing is materialised).
"""
import contextlib
import os
from typing import Dict, List, Tuple, Type, Union

import torch
import torch.nn as nn

from ...register import register
from .. import train_ops as T
from ..layers import BatchNorm2d, Conv2d, Plan
from .backbone import Backbone
from .utils import load_pretrained

__all__ = ["HRNet", "hrnet_w32", "hrnet_w48"]


class BasicBlock(nn.Module):
    """relu(bn2(conv2(relu(bn1(conv1 x)))) + identity) - hrnet.py:30-83 (2 launches)."""

    expansion: int = 1

    def __init__(self, in_channels: int, channels: int, stride: int = 1, down_sample: nn.Module = None) -> None:
        super().__init__()
        self.conv1 = Conv2d(in_channels, channels, 3, stride=stride, padding=1)
        self.bn1 = BatchNorm2d(channels)
        self.conv2 = Conv2d(channels, channels, 3, stride=1, padding=1)
        self.bn2 = BatchNorm2d(channels)
        self.down_sample = down_sample

    def emit(self, plan: Plan, x: torch.Tensor) -> torch.Tensor:
        identity = x
        if self.down_sample is not None:
            identity = plan.conv(x, self.down_sample[0], self.down_sample[1])
        elif plan.fuses_basic_block(x, self.conv1, self.conv2):
            return plan.basic_block(x, self.conv1, self.bn1, self.conv2, self.bn2)  # fp16, 32 channels: one launch
        out = plan.conv(x, self.conv1, self.bn1, relu=True)
        return plan.conv(out, self.conv2, self.bn2, relu=True, res1=identity)

    def train_forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.down_sample is None:
            return T.residual_block(x, [(self.conv1, self.bn1), (self.conv2, self.bn2)])
        xd, xc = T.fan_out(x, 2)  # two consumers: their gradients are summed (and reduced for the producer's BatchNorm) in one launch
        identity = T.conv_bn_act(xd, self.down_sample[0], self.down_sample[1], relu=False)
        out = T.conv_bn_act(xc, self.conv1, self.bn1, relu=True)
        return T.conv_bn_act(out, self.conv2, self.bn2, relu=True, res=identity)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (stride) -> 1x1, + identity / down_sample - hrnet.py:86-146 (3-4 launches)."""

    expansion: int = 4

    def __init__(self, in_channels: int, channels: int, stride: int = 1, down_sample: nn.Module = None) -> None:
        super().__init__()
        width = channels
        self.conv1 = Conv2d(in_channels, width, 1)
        self.bn1 = BatchNorm2d(width)
        self.conv2 = Conv2d(width, width, 3, stride=stride, padding=1)
        self.bn2 = BatchNorm2d(width)
        self.conv3 = Conv2d(width, channels * self.expansion, 1)
        self.bn3 = BatchNorm2d(channels * self.expansion)
        self.down_sample = down_sample

    def emit(self, plan: Plan, x: torch.Tensor, reduced: torch.Tensor = None, nxt: "Bottleneck" = None):
        """``reduced``: relu(bn1(conv1 x)) when the previous block's last launch already produced it; ``nxt``: the following
        Bottleneck - when its reduce conv can ride on this block's expand conv (``Plan.fuses_expand_reduce``, fp16 plans) the pair
        (output, reduced output for ``nxt``) is returned instead of the output alone."""
        identity = x
        ds_in_chain = None
        if (self.down_sample is not None and len(self.down_sample) == 2 and nxt is not None and self.conv2.stride == 1
                and self.conv2.out_channels == 64 and plan.fuses_ds_expand_reduce(x, self.down_sample[0], self.conv3, nxt.conv1)):
            # the down-sample conv is computed INSIDE the expand + reduce chain launch below - its 256-channel output is neither
            # written nor read back
            ds_in_chain = (x, self.down_sample[0], self.down_sample[1])
        elif (self.down_sample is not None and reduced is None and len(self.down_sample) == 2
              and plan.fuses_dual_pw(x, self.down_sample[0], self.conv1)):
            # the down-sample conv and the reduce conv read the same input: one launch (fp16 plans without the form above)
            identity, reduced = plan.dual_pw(x, self.down_sample[0], self.down_sample[1], False, self.conv1, self.bn1, True)
        elif self.down_sample is not None:
            identity = plan.conv(x, self.down_sample[0], self.down_sample[1])
        out = reduced if reduced is not None else plan.conv(x, self.conv1, self.bn1, relu=True)
        out = plan.conv(out, self.conv2, self.bn2, relu=True)
        if ds_in_chain is not None:
            return plan.expand_reduce(out, None, self.conv3, self.bn3, nxt.conv1, nxt.bn1, ds=ds_in_chain)
        if nxt is not None and self.conv2.stride == 1 and plan.fuses_expand_reduce(out, identity, self.conv3, nxt.conv1):
            return plan.expand_reduce(out, identity, self.conv3, self.bn3, nxt.conv1, nxt.bn1)
        if nxt is None and plan.fuses_expand_only(out, identity, self.conv3):
            return plan.expand_reduce(out, identity, self.conv3, self.bn3, None, None)
        y = plan.conv(out, self.conv3, self.bn3, relu=True, res1=identity)
        return y if nxt is None else (y, None)

    def train_forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.down_sample is None:
            return T.residual_block(x, [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)])
        xd, xc = T.fan_out(x, 2)  # two consumers: their gradients are summed (and reduced for the producer's BatchNorm) in one launch
        identity = T.conv_bn_act(xd, self.down_sample[0], self.down_sample[1], relu=False)
        out = T.conv_bn_act(xc, self.conv1, self.bn1, relu=True)
        out = T.conv_bn_act(out, self.conv2, self.bn2, relu=True)
        return T.conv_bn_act(out, self.conv3, self.bn3, relu=True, res=identity)


def emit_block_sequence(plan: Plan, blocks, x: torch.Tensor) -> torch.Tensor:
    """A run of residual blocks (HRNet's layer1, a ResNet layer): consecutive Bottlenecks hand the next block's reduced input over
    when the pair ran as one chain launch (``Bottleneck.emit``: (output, the next block's reduced input or None))."""
    reduced = None
    for i, blk in enumerate(blocks):
        if isinstance(blk, Bottleneck) and i + 1 < len(blocks) and isinstance(blocks[i + 1], Bottleneck):
            x, reduced = blk.emit(plan, x, reduced, blocks[i + 1])
        elif isinstance(blk, Bottleneck):
            x, reduced = blk.emit(plan, x, reduced), None
        else:
            x, reduced = blk.emit(plan, x), None
    return x


def _conv_bn(cin: int, cout: int, k: int, stride: int = 1, padding: int = 0, relu: bool = False) -> nn.Sequential:
    """SequentialCell(conv, bn[, relu]) of the reference; the ReLU is a marker only (fused in the epilogue)."""
    layers = [Conv2d(cin, cout, k, stride=stride, padding=padding), BatchNorm2d(cout)]
    if relu:
        layers.append(nn.ReLU())
    return nn.Sequential(*layers)


def _emit_conv_bn(plan: Plan, seq: nn.Sequential, x: torch.Tensor, **kw) -> torch.Tensor:
    return plan.conv(x, seq[0], seq[1], relu=kw.pop("relu", len(seq) > 2), **kw)


def _train_conv_bn(seq: nn.Sequential, x: torch.Tensor, relu=None) -> torch.Tensor:
    return T.conv_bn_act(x, seq[0], seq[1], relu=(len(seq) > 2) if relu is None else relu)


def _flush_queued_weight_gradients(grad):
    T.flush_wgrad_jobs_early()
    return None  # the gradient is not changed


_BRANCH_STREAMS = {}
_BRANCH_STREAMS_ON = [False]


def set_branch_streams(on: bool) -> bool:
    """Run the branches of every HRModule on side streams in ``train_forward`` (``GraphedTrainStep`` turns this on around its
    warm-up and capture: inside a hipGraph the fork / join become graph dependencies, so the replay overlaps the small launches
    of the deep branches with the large ones).  Returns the previous setting.  ``MINDPOSE_TRAIN_BRANCH_STREAMS=0/1`` overrides."""
    prev = _BRANCH_STREAMS_ON[0]
    _BRANCH_STREAMS_ON[0] = bool(on)
    return prev


@contextlib.contextmanager
def quiet_accumulate_grad_stream_warning():
    """Parameters whose gradient goes back through autograd (the head conv's bias, the few fp32 parameters outside the arena's direct
    path) meet an AccumulateGrad node created on the caller's stream while the node that produced the gradient ran on a branch
    stream.  Inside a graphed step that mismatch is the design (the branch streams are joined explicitly; graph == eager and
    segmented == one-graph are tested bit for bit), so torch's warning about it is off WHILE the step's warm-up passes and capture
    run the autograd engine - and on again afterwards: the user's own eager backward passes keep the diagnostic."""
    switch = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    if switch is not None:
        switch(False)
    try:
        yield
    finally:
        if switch is not None:
            switch(True)


def branch_streams_enabled() -> bool:
    env = os.environ.get("MINDPOSE_TRAIN_BRANCH_STREAMS")
    return _BRANCH_STREAMS_ON[0] if env is None else env == "1"



def join_branch_streams(device) -> None:
    """The current stream waits for every branch / row side stream of ``device`` (a segmented capture ends each backward segment
    with this: the gradients of a stage boundary are produced on the side streams and nothing else joins them before the capture
    of that segment ends)."""
    cur = torch.cuda.current_stream(device)
    for st in _BRANCH_STREAMS.get(device, []):
        cur.wait_stream(st)


def _branch_streams(device, n):
    have = _BRANCH_STREAMS.setdefault(device, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


class HRModule(nn.Module):
    """High-resolution module: 4 blocks per branch, then the exchange unit - hrnet.py:149-344."""

    def __init__(self, num_branches: int, block: Type[Union[BasicBlock, Bottleneck]], num_blocks: List[int],
                 num_inchannels: List[int], num_channels: List[int], multi_scale_output: bool = True) -> None:
        super().__init__()
        self._check_branches(num_branches, num_blocks, num_inchannels, num_channels)
        self.num_inchannels = num_inchannels
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.branches = nn.ModuleList(
            [self._make_one_branch(i, block, num_blocks, num_channels) for i in range(num_branches)])
        self.fuse_layers = self._make_fuse_layers()

    @staticmethod
    def _check_branches(num_branches, num_blocks, num_inchannels, num_channels) -> None:
        if num_branches != len(num_blocks):
            raise ValueError(f"NUM_BRANCHES({num_branches})!= NUM_BLOCKS({len(num_blocks)})")
        if num_branches != len(num_channels):
            raise ValueError(f"NUM_BRANCHES({num_branches})!= NUM_CHANNELS({len(num_channels)})")
        if num_branches != len(num_inchannels):
            raise ValueError(f"NUM_BRANCHES({num_branches}) != NUM_INCHANNELS({len(num_inchannels)})")

    def _make_one_branch(self, i, block, num_blocks, num_channels, stride: int = 1) -> nn.Sequential:
        down = None
        if stride != 1 or self.num_inchannels[i] != num_channels[i] * block.expansion:
            down = _conv_bn(self.num_inchannels[i], num_channels[i] * block.expansion, 1, stride=stride)
        layers = [block(self.num_inchannels[i], num_channels[i], stride, down_sample=down)]
        self.num_inchannels[i] = num_channels[i] * block.expansion
        for _ in range(1, num_blocks[i]):
            layers.append(block(self.num_inchannels[i], num_channels[i]))
        return nn.Sequential(*layers)

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        nb, ch = self.num_branches, self.num_inchannels
        rows = []
        for i in range(nb if self.multi_scale_output else 1):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(_conv_bn(ch[j], ch[i], 1))
                elif j == i:
                    row.append(nn.Identity())
                else:
                    chain = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        chain.append(_conv_bn(ch[j], ch[i] if last else ch[j], 3, stride=2, padding=1, relu=not last))
                    row.append(nn.Sequential(*chain))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def emit(self, plan: Plan, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        xs = list(xs)
        # branch i runs on execution lane i: the four resolutions are independent until the exchange unit, and the deep
        # branches (few workgroups per launch) fill the chip while the wide ones run
        for i in range(self.num_branches):
            plan.set_lane(i)
            for blk in self.branches[i]:
                xs[i] = blk.emit(plan, xs[i])
        if self.num_branches == 1:
            return xs
        plan.barrier()  # every row of the exchange unit reads every branch
        outs = []
        nb = self.num_branches
        for i in range(len(self.fuse_layers)):
            plan.set_lane(i)  # row i feeds branch i of the next module: no barrier needed after the exchange unit
            # reference order (hrnet.py:327-339): y = t_0; y = y + t_1; ... ; relu(y).  Every term except
            # the identity x_i is a conv launch whose epilogue adds the running sum (res1); the identity
            # rides as res2 on the launch that precedes it (or as res1 of the first launch when i == 0).
            down_terms = [j for j in range(nb) if j < i]
            up_terms = [j for j in range(nb) if j > i]
            vec = plan.half or xs[i].shape[3] % 4 == 0  # the streaming fuse-sum kernel moves 16 B per lane
            acc = xs[0] if i == 0 else None
            ybuf = None
            # (a) down-sampling terms j < i: 3x3 s2 chains whose last conv accumulates into ybuf in its epilogue;
            #     the identity x_i rides as res2 of the launch that precedes it in the reference order
            for j in down_terms:
                last = j == down_terms[-1]
                relu = last and not up_terms
                res2 = xs[i] if j == i - 1 else None
                chain = self.fuse_layers[i][j]
                t = xs[j]
                for k in range(len(chain) - 1):
                    t = _emit_conv_bn(plan, chain[k], t)
                if ybuf is None:
                    ybuf = plan.alloc(*xs[i].shape)
                _emit_conv_bn(plan, chain[-1], t, relu=relu, res1=acc, res2=res2, out=ybuf)
                acc = ybuf
            # (b) up-sampling terms j > i: BN(conv1x1(x_j)) stays at low resolution; ONE streaming pass adds all of
            #     them (nearest up-sampling on the fly, reference order) and applies the ReLU
            if up_terms:
                if ybuf is None:
                    ybuf = plan.alloc(*xs[i].shape)
                ups = []
                for j in up_terms:
                    seq = self.fuse_layers[i][j]
                    up = xs[i].shape[2] // xs[j].shape[2]
                    if xs[j].shape[2] * up != xs[i].shape[2] or xs[j].shape[3] * up != xs[i].shape[3]:
                        raise ValueError(
                            f"HRNet fuse needs an integer nearest-upsample factor, got {tuple(xs[j].shape[2:])} -> "
                            f"{tuple(xs[i].shape[2:])}; use an input whose height and width are multiples of 32")
                    if vec:
                        ups.append((plan.conv(xs[j], seq[0], seq[1]), up))
                    else:  # odd widths: accumulate through the conv epilogue's up-sample mapping
                        plan.conv(xs[j], seq[0], seq[1], relu=j == up_terms[-1], res1=acc, out=ybuf, upsample=up)
                        acc = ybuf
                if vec:
                    plan.fuse_sum(acc, ups, ybuf, relu=True)
            outs.append(ybuf)
        return outs

    def train_forward(self, xs: List[torch.Tensor], fork: bool = True, join: bool = True) -> List[torch.Tensor]:
        """Training form of hrnet.py:318-344: same term order; one exchange-unit sum kernel per row.  ``join`` = False: the rows stay on
        their side streams when this returns, ``fork`` = False: the branches start on theirs without waiting for the current stream -
        row i of one module feeds branch i of the next ON THE SAME STREAM, so consecutive modules of a stage need no join / fork star
        in between (MINDPOSE_TRAIN_CHAIN_MODULES=0: one after every module, as before)."""
        xs = list(xs)
        if self.num_branches > 1 and xs[0].is_cuda and branch_streams_enabled():
            # the branches are independent until the exchange unit: branch i > 0 on side stream i (forked from / joined to the
            # current stream), so the small launches of the deep branches overlap the large ones; autograd replays each node on
            # its forward stream, which parallelises the backward the same way
            cur = torch.cuda.current_stream(xs[0].device)
            side = _branch_streams(xs[0].device, self.num_branches - 1)
            for i in range(1, self.num_branches):
                if fork:
                    side[i - 1].wait_stream(cur)
                with torch.cuda.stream(side[i - 1]):
                    for blk in self.branches[i]:
                        xs[i] = blk.train_forward(xs[i])
            for blk in self.branches[0]:
                xs[0] = blk.train_forward(xs[0])
            for st in side:
                cur.wait_stream(st)
            # (the fan-out nodes below stay on the current stream: forked onto the branch streams - so that the backward pass has no
            # main-stream spine through the fan-ins - a side stream waits on several other side streams, and ROCm 7.2 faults in
            # hipStreamEndCapture on that shape; tried in round 5)
        else:
            for i in range(self.num_branches):
                for blk in self.branches[i]:
                    xs[i] = blk.train_forward(xs[i])
        if self.num_branches == 1:
            return xs
        # every branch output feeds every row of the exchange unit: one handle per row, so that the backward pass sums the rows'
        # gradients in one launch per branch (T.FanOutFn) instead of pairwise
        rows = len(self.fuse_layers)
        if rows == 1:
            # one row = one consumer per branch output, no fan-in node: the branch chains of the backward pass would each be released
            # behind "their" term's launches on the current stream - one after the other in a replayed graph.  One node in between.
            xs = T.grad_join(xs)
        handles = T.fan_out_many(xs, [rows] * len(xs))  # ONE autograd node: the backward pass forks the branch chains from a single point
        outs = []
        # The rows of the exchange unit (hrnet.py:318-344) are independent chains of small launches - 1x1 conv + BatchNorm per
        # up-sampled term, one to three stride-2 conv + BatchNorm groups per down-sampled one, the sum - ~17 (three branches) to ~36
        # (four) launches of ~10 us that ran one after the other: row i > 0 goes to side stream i, like the branches above
        # (MINDPOSE_TRAIN_FUSE_STREAMS=0: all rows on the current stream).
        row_streams = None
        if rows > 1 and xs[0].is_cuda and branch_streams_enabled() and os.environ.get("MINDPOSE_TRAIN_FUSE_STREAMS", "1") != "0":
            cur = torch.cuda.current_stream(xs[0].device)
            row_streams = _branch_streams(xs[0].device, rows - 1)
            for st in row_streams:
                st.wait_stream(cur)
        for i in range(rows):
            ctxm = torch.cuda.stream(row_streams[i - 1]) if (row_streams is not None and i > 0) else contextlib.nullcontext()
            with ctxm:
                outs.append(self._train_row(i, xs, handles))
        if row_streams is not None and join:
            for st in row_streams:
                cur.wait_stream(st)
        return outs

    def _train_row(self, i, xs, handles):
        """Row i of the exchange unit: sum over the branches j of (identity | up-sampled 1x1 conv | stride-2 conv chain)."""
        if True:
            terms = []
            for j in range(self.num_branches):
                xj = handles[j][i]
                if j == i:
                    terms.append((xj, 1))
                elif j > i:
                    seq = self.fuse_layers[i][j]
                    up = xs[i].shape[2] // xs[j].shape[2]
                    if xs[j].shape[2] * up != xs[i].shape[2] or xs[j].shape[3] * up != xs[i].shape[3]:
                        raise ValueError("HRNet fuse needs an integer nearest-upsample factor")
                    terms.append((_train_conv_bn(seq, xj, relu=False), up))
                else:
                    chain = self.fuse_layers[i][j]
                    t = xj
                    for k in range(len(chain)):
                        t = _train_conv_bn(chain[k], t)
                    terms.append((t, 1))
            return T.fuse_sum(terms[0][0], terms[1:])


@register("backbone")
class HRNet(Backbone):
    """HRNet backbone - hrnet.py:348-614.  ``stage_cfg`` has the reference's schema."""

    blocks_dict = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}

    def __init__(self, stage_cfg: Dict[str, Dict[str, int]], in_channels: int = 3) -> None:
        super().__init__()
        self.stage_cfg = stage_cfg
        self.conv1 = Conv2d(in_channels, 64, 3, stride=2, padding=1)
        self.bn1 = BatchNorm2d(64)
        self.conv2 = Conv2d(64, 64, 3, stride=2, padding=1)
        self.bn2 = BatchNorm2d(64)

        self.stage1_cfg = stage_cfg["stage1"]
        block = self.blocks_dict[self.stage1_cfg["block"]]
        self.layer1 = self._make_layer(block, 64, self.stage1_cfg["num_channels"][0], self.stage1_cfg["num_blocks"][0])
        pre = [self.stage1_cfg["num_channels"][0] * block.expansion]

        for idx in (2, 3, 4):
            cfg = stage_cfg[f"stage{idx}"]
            setattr(self, f"stage{idx}_cfg", cfg)
            block = self.blocks_dict[cfg["block"]]
            channels = [c * block.expansion for c in cfg["num_channels"]]
            trans, flags = self._make_transition_layer(pre, channels)
            setattr(self, f"transition{idx - 1}", trans)
            setattr(self, f"transition{idx - 1}_flags", flags)
            mso = True if idx < 4 else cfg.get("multiscale_output", False)
            stage, pre = self._make_stage(cfg, channels, multi_scale_output=mso)
            setattr(self, f"stage{idx}", stage)

    def _make_transition_layer(self, pre: List[int], cur: List[int]) -> Tuple[nn.ModuleList, List[bool]]:
        layers, flags = [], []
        for i in range(len(cur)):
            if i < len(pre):
                if cur[i] != pre[i]:
                    layers.append(_conv_bn(pre[i], cur[i], 3, padding=1, relu=True))
                    flags.append(True)
                else:
                    layers.append(nn.Identity())
                    flags.append(False)
            else:
                chain = []
                for j in range(i + 1 - len(pre)):
                    cout = cur[i] if j == i - len(pre) else pre[-1]
                    chain.append(_conv_bn(pre[-1], cout, 3, stride=2, padding=1, relu=True))
                layers.append(nn.Sequential(*chain))
                flags.append(True)
        return nn.ModuleList(layers), flags

    def _make_layer(self, block, in_channels: int, out_channels: int, blocks: int, stride: int = 1) -> nn.Sequential:
        down = None
        if stride != 1 or in_channels != out_channels * block.expansion:
            down = _conv_bn(in_channels, out_channels * block.expansion, 1, stride=stride)
        layers = [block(in_channels, out_channels, stride, down_sample=down)]
        for _ in range(1, blocks):
            layers.append(block(out_channels * block.expansion, out_channels))
        return nn.Sequential(*layers)

    def _make_stage(self, cfg, num_inchannels, multi_scale_output: bool = True):
        block = self.blocks_dict[cfg["block"]]
        modules = []
        for i in range(cfg["num_modules"]):
            mso = not (not multi_scale_output and i == cfg["num_modules"] - 1)
            modules.append(HRModule(cfg["num_branches"], block, cfg["num_blocks"], num_inchannels,
                                    cfg["num_channels"], mso))
            num_inchannels = modules[-1].num_inchannels
        return nn.Sequential(*modules), num_inchannels

    def emit(self, plan: Plan, x: torch.Tensor) -> torch.Tensor:
        """Recorded form of ``forward_feature`` hrnet.py:559-605."""
        if plan.fuses_stem(x, self.conv1):  # the dedicated first-conv kernels (fp16 plans: reads the fp32 image itself, no layout pass)
            x = plan.stem(x, self.conv1, self.bn1)
        else:
            x = plan.enter(x)
            x = plan.conv(x, self.conv1, self.bn1, relu=True)
        x = plan.conv(x, self.conv2, self.bn2, relu=True)
        x = emit_block_sequence(plan, list(self.layer1), x)
        ys = [x]
        for idx in (2, 3, 4):
            trans = getattr(self, f"transition{idx - 1}")
            flags = getattr(self, f"transition{idx - 1}_flags")
            cfg = getattr(self, f"stage{idx}_cfg")
            xs = []
            for i in range(cfg["num_branches"]):
                plan.set_lane(min(i, len(ys) - 1))  # a transition runs on the lane that produced its input
                if not flags[i]:
                    xs.append(ys[i])
                elif i < len(ys):
                    xs.append(_emit_conv_bn(plan, trans[i], ys[i]))
                else:
                    t = ys[-1]  # new branches are fed from the LAST previous output (hrnet.py:591,:600)
                    for seq in trans[i]:
                        t = _emit_conv_bn(plan, seq, t)
                    xs.append(t)
            plan.barrier()  # the new branch's lane consumes what another lane just produced
            for mod in getattr(self, f"stage{idx}"):
                xs = mod.emit(plan, xs)
            ys = xs
        plan.set_lane(0)  # the last exchange unit has a single row (lane 0); the head continues there
        return ys[0]

    def train_forward(self, x: torch.Tensor) -> torch.Tensor:
        """Training form of ``forward_feature`` hrnet.py:559-605 (batch-statistics BatchNorm, autograd)."""
        if self.amp_level in ("O2", "O3") and not T._is_c8(x):
            x = T.to_c8(x)
        x = T.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        x = T.conv_bn_act(x, self.conv2, self.bn2, relu=True)
        for blk in self.layer1:
            x = blk.train_forward(x)
        if x.requires_grad and x.is_cuda:
            # when the backward pass arrives here only the branch-less stage 1 and the stem are left: the weight gradients queued so
            # far go out on a side stream beside them (train_ops.flush_wgrad_jobs_early) instead of alone behind the pass
            x.register_hook(_flush_queued_weight_gradients)
        ys = [x]
        cuts = getattr(self, "_train_cut_sink", None)
        cut_at = getattr(self, "_train_cut_at", None)  # boundaries to cut (0 = in front of stage 2 ... 2 = stage 4); None = all three
        for idx in (2, 3, 4):
            if cuts is not None and (cut_at is None or idx - 2 in cut_at):
                # stage boundary of a segmented backward pass (utils/graph_step.py): the autograd graph is CUT here - the next stage
                # continues from detached leaves (no launch: a detach is a view), the step feeds their gradients to the tensors
                # they were cut from.  The BatchNorm hand-over link travels with the tensor.
                leaves = []
                for y in ys:
                    leaf = y.detach().requires_grad_()
                    link = getattr(y, "_mp_bn_link", None)
                    if link is not None:
                        leaf._mp_bn_link = link
                    leaves.append(leaf)
                cuts.append((list(ys), leaves))
                ys = leaves
            trans = getattr(self, f"transition{idx - 1}")
            flags = getattr(self, f"transition{idx - 1}_flags")
            cfg = getattr(self, f"stage{idx}_cfg")
            # the last previous output feeds its own branch AND every new branch (hrnet.py:591, :600): one handle per consumer, so
            # that the gradients meet in one fan-in launch instead of autograd's adds
            uses = [1 if i < len(ys) else 0 for i in range(len(ys))]
            uses[-1] += sum(1 for i in range(cfg["num_branches"]) if i >= len(ys))
            handles = [list(hs) for hs in T.fan_out_many(ys, uses)]
            xs = []
            for i in range(cfg["num_branches"]):
                if not flags[i]:
                    xs.append(handles[i].pop())
                elif i < len(ys):
                    xs.append(_train_conv_bn(trans[i], handles[i].pop()))
                else:
                    t = handles[-1].pop()
                    for seq in trans[i]:
                        t = _train_conv_bn(seq, t)
                    xs.append(t)
            mods = list(getattr(self, f"stage{idx}"))
            chain = os.environ.get("MINDPOSE_TRAIN_CHAIN_MODULES", "1") != "0"
            for k, mod in enumerate(mods):
                first, last = k == 0, k == len(mods) - 1
                xs = mod.train_forward(xs, fork=first or not chain, join=last or not chain)
            ys = xs
        return ys[0]

    def train_segments(self):
        """Module groups of the training graph in BACKWARD order, cut at the stage boundaries ``train_forward`` reports through
        ``_train_cut_sink`` (last boundary first): [stage 4 + transition 3], [stage 3 + transition 2], [stage 2 + transition 1],
        [stage 1 + stem].  A segmented step (utils/graph_step.py) runs the backward pass group by group and hands each group's
        finished gradient buckets to the all-reduce while the next group computes."""
        return [[self.stage4, self.transition3], [self.stage3, self.transition2], [self.stage2, self.transition1],
                [self.layer1, self.conv1, self.bn1, self.conv2, self.bn2]]

    @property
    def out_channels(self) -> int:
        return self.stage4_cfg["num_channels"][0]


def _hrnet_cfg(c: int) -> Dict[str, Dict]:
    return dict(
        stage1=dict(num_modules=1, num_branches=1, block="BOTTLENECK", num_blocks=[4], num_channels=[64]),
        stage2=dict(num_modules=1, num_branches=2, block="BASIC", num_blocks=[4, 4], num_channels=[c, 2 * c]),
        stage3=dict(num_modules=4, num_branches=3, block="BASIC", num_blocks=[4, 4, 4], num_channels=[c, 2 * c, 4 * c]),
        stage4=dict(num_modules=3, num_branches=4, block="BASIC", num_blocks=[4, 4, 4, 4],
                    num_channels=[c, 2 * c, 4 * c, 8 * c], multiscale_output=False),
    )


@register("backbone")
def hrnet_w32(pretrained: bool = False, ckpt_url: str = "", in_channels: int = 3) -> HRNet:
    """HRNet with width 32 - hrnet.py:618-666."""
    model = HRNet(_hrnet_cfg(32), in_channels=in_channels)
    if pretrained:
        load_pretrained(model, ckpt_url=ckpt_url)
    return model


@register("backbone")
def hrnet_w48(pretrained: bool = False, ckpt_url: str = "", in_channels: int = 3) -> HRNet:
    """HRNet with width 48 - hrnet.py:670-718."""
    model = HRNet(_hrnet_cfg(48), in_channels=in_channels)
    if pretrained:
        load_pretrained(model, ckpt_url=ckpt_url)
    return model
