"""CPU oracle: HRNet / ResNet backbones and heads (TEST INFRASTRUCTURE, see oracle/__init__.py).

torch-CPU fp32 functional restatement of
  * ``HRNet``  /root/reference/mindpose/models/backbones/hrnet.py:348-614 (W32 cfg :618-666, W48 :670-718)
  * ``ResNet`` /root/reference/mindpose/models/backbones/resnet.py:142-273
  * ``HRNetHead`` heads/hrnet_head.py:14-49, ``SimpleBaselineHead`` heads/simple_baseline_head.py:17-98
driven by a flat ``{parameter name: tensor}`` dict that uses the reference's parameter names
(``conv1.weight``, ``bn1.gamma`` / ``beta`` / ``moving_mean`` / ``moving_variance``, ...), so the same
dict feeds the HIP product modules and this oracle.

PARITY UNPINNED by the reference (arithmetic lives in MindSpore, tests are shape-only).
MindSpore layer semantics assumed [MS-knowledge, SURVEY.md 8c]:
  Conv2d weight (Cout,Cin,kh,kw), no bias unless has_bias; pad_mode="pad" padding=p is symmetric;
  default pad_mode="same" is a no-op for the 1x1 convs used here; BatchNorm2d eps=1e-5 (eval:
  moving statistics); MaxPool2d(pad_mode="same") pads bottom/right with -inf;
  Conv2dTranspose(k=4,s=2,pad_mode="pad",padding=1) doubles H,W, weight (Cin,Cout,kh,kw);
  ResizeNearestNeighbor align_corners=False -> src = floor(dst*in/out).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5

HRNET_CFG = {
    "hrnet_w32": dict(
        stage1=dict(num_modules=1, num_branches=1, block="BOTTLENECK", num_blocks=[4], num_channels=[64]),
        stage2=dict(num_modules=1, num_branches=2, block="BASIC", num_blocks=[4, 4], num_channels=[32, 64]),
        stage3=dict(num_modules=4, num_branches=3, block="BASIC", num_blocks=[4, 4, 4], num_channels=[32, 64, 128]),
        stage4=dict(num_modules=3, num_branches=4, block="BASIC", num_blocks=[4, 4, 4, 4],
                    num_channels=[32, 64, 128, 256], multiscale_output=False),
    ),
    "hrnet_w48": dict(
        stage1=dict(num_modules=1, num_branches=1, block="BOTTLENECK", num_blocks=[4], num_channels=[64]),
        stage2=dict(num_modules=1, num_branches=2, block="BASIC", num_blocks=[4, 4], num_channels=[48, 96]),
        stage3=dict(num_modules=4, num_branches=3, block="BASIC", num_blocks=[4, 4, 4], num_channels=[48, 96, 192]),
        stage4=dict(num_modules=3, num_branches=4, block="BASIC", num_blocks=[4, 4, 4, 4],
                    num_channels=[48, 96, 192, 384], multiscale_output=False),
    ),
}
RESNET_LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3], "resnet152": [3, 8, 36, 3]}


class _P:
    """Prefix view over the flat parameter dict.  ``train=True`` switches BatchNorm to batch statistics (moving
    statistics updated in place with momentum 0.9, mindspore.nn.BatchNorm2d training) and keeps the autograd graph
    of the parameter tensors (used to check the HIP backward kernels)."""

    def __init__(self, params, prefix="", train=False, amp=False):
        self.params, self.prefix, self.train, self.amp = params, prefix, train, amp

    def sub(self, name):
        return _P(self.params, f"{self.prefix}{name}.", self.train, self.amp)

    def __getitem__(self, name):
        t = self.params[self.prefix + name]
        if not torch.is_tensor(t):
            t = torch.as_tensor(t)
        if self.train:
            return t
        return t.detach().to(torch.float32).cpu()

    def has(self, name):
        return (self.prefix + name) in self.params


def _r(p, x):
    """amp O2 emulation: every cell output is an fp16 tensor (mindspore.amp.auto_mixed_precision casts the cells to
    fp16 and keeps BatchNorm in fp32 with an fp16 output) - round to nearest-even fp16, keep computing in fp32."""
    return x.half().float() if p.amp else x


def _conv(p, x, stride=1, padding=0):
    bias = p["bias"] if p.has("bias") else None
    if p.amp:  # fp16 operands, fp32 accumulation (cube unit), fp16 output
        return _r(p, F.conv2d(x, _r(p, p["weight"]), None if bias is None else _r(p, bias), stride=stride, padding=padding))
    return F.conv2d(x, p["weight"], bias, stride=stride, padding=padding)


def _bn(p, x):
    if p.amp and not p.train:
        return _r(p, F.batch_norm(x, p["moving_mean"], p["moving_variance"], p["gamma"], p["beta"], training=False, eps=BN_EPS))
    if p.train:  # torch momentum 0.1 == MindSpore momentum 0.9 (weight of the OLD moving value)
        # amp O2 keeps BatchNorm in fp32 (statistics, gamma / beta) and hands an fp16 tensor on: one rounding of the output
        return _r(p, F.batch_norm(x, p["moving_mean"], p["moving_variance"], p["gamma"], p["beta"],
                                  training=True, momentum=0.1, eps=BN_EPS))
    return F.batch_norm(x, p["moving_mean"], p["moving_variance"], p["gamma"], p["beta"],
                        training=False, eps=BN_EPS)


def _basic_block(p, x):
    """hrnet.py:66-83: relu(bn2(conv2(relu(bn1(conv1 x)))) + identity)."""
    out = F.relu(_bn(p.sub("bn1"), _conv(p.sub("conv1"), x, 1, 1)))
    out = _bn(p.sub("bn2"), _conv(p.sub("conv2"), out, 1, 1))
    return F.relu(_r(p, out + x))


def _bottleneck(p, x, stride=1):
    """hrnet.py:126-146 / resnet.py:118-138: 1x1 -> 3x3 (stride here) -> 1x1, + (down_sample) identity."""
    out = F.relu(_bn(p.sub("bn1"), _conv(p.sub("conv1"), x)))
    out = F.relu(_bn(p.sub("bn2"), _conv(p.sub("conv2"), out, stride, 1)))
    out = _bn(p.sub("bn3"), _conv(p.sub("conv3"), out))
    identity = x
    if p.has("down_sample.0.weight"):
        ds = p.sub("down_sample")
        identity = _bn(ds.sub("1"), _conv(ds.sub("0"), x, stride, 0))
    return F.relu(_r(p, out + identity))


def _hr_module(p, xs, num_branches, num_blocks, multi_scale_output):
    """HRModule.construct hrnet.py:318-344."""
    xs = list(xs)
    for i in range(num_branches):
        for b in range(num_blocks[i]):
            xs[i] = _basic_block(p.sub(f"branches.{i}.{b}"), xs[i])
    if num_branches == 1:
        return xs
    outs = []
    for i in range(num_branches if multi_scale_output else 1):
        fl = p.sub(f"fuse_layers.{i}")
        y = None
        for j in range(num_branches):
            if j == i:
                t = xs[j]
            elif j > i:
                t = _bn(fl.sub(f"{j}.1"), _conv(fl.sub(f"{j}.0"), xs[j]))
                t = F.interpolate(t, size=xs[i].shape[2:], mode="nearest")
            else:
                t = xs[j]
                for k in range(i - j):
                    s = fl.sub(f"{j}.{k}")
                    t = _bn(s.sub("1"), _conv(s.sub("0"), t, 2, 1))
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else _r(p, y + t)
        outs.append(F.relu(y))
    return outs


def hrnet_forward(params, x, name="hrnet_w32", prefix="", train=False, amp=False):
    """HRNet.forward_feature hrnet.py:559-605.  ``amp``: op-by-op fp16 emulation of amp level O2."""
    cfg = HRNET_CFG[name]
    p = _P(params, prefix, train, amp)
    x = torch.as_tensor(x) if train else torch.as_tensor(x, dtype=torch.float32)  # train: keep dtype (fp64 oracle runs)
    x = _r(p, x)
    x = F.relu(_bn(p.sub("bn1"), _conv(p.sub("conv1"), x, 2, 1)))
    x = F.relu(_bn(p.sub("bn2"), _conv(p.sub("conv2"), x, 2, 1)))
    for b in range(cfg["stage1"]["num_blocks"][0]):
        x = _bottleneck(p.sub(f"layer1.{b}"), x)

    def transition(tname, prev, n_cur):
        """_make_transition_layer hrnet.py:440-496; new branches read the LAST previous output."""
        tp = p.sub(tname)
        outs = []
        for i in range(n_cur):
            if i < len(prev):
                if tp.has(f"{i}.0.weight"):
                    s = tp.sub(f"{i}")
                    outs.append(F.relu(_bn(s.sub("1"), _conv(s.sub("0"), prev[i], 1, 1))))
                else:
                    outs.append(prev[i])
            else:
                t = prev[-1]
                for j in range(i + 1 - len(prev)):
                    s = tp.sub(f"{i}.{j}")
                    t = F.relu(_bn(s.sub("1"), _conv(s.sub("0"), t, 2, 1)))
                outs.append(t)
        return outs

    ys = [x]
    for si, sname in enumerate(["stage2", "stage3", "stage4"], start=1):
        scfg = cfg[sname]
        xs = transition(f"transition{si}", ys, scfg["num_branches"])
        mso = scfg.get("multiscale_output", True)
        for m in range(scfg["num_modules"]):
            last = m == scfg["num_modules"] - 1
            xs = _hr_module(p.sub(f"{sname}.{m}"), xs, scfg["num_branches"], scfg["num_blocks"],
                            multi_scale_output=not (last and not mso))
        ys = xs
    return ys[0]


def maxpool3x3s2_same(x):
    """nn.MaxPool2d(3, 2, pad_mode="same") resnet.py:190 - pad bottom/right only (even H, W)."""
    h, w = x.shape[2:]
    oh, ow = -(-h // 2), -(-w // 2)
    ph, pw = max((oh - 1) * 2 + 3 - h, 0), max((ow - 1) * 2 + 3 - w, 0)
    x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
    return F.max_pool2d(x, 3, 2)


def resnet_forward(params, x, name="resnet50", prefix="", amp=False, train=False):
    """ResNet.forward_feature resnet.py:247-264."""
    p = _P(params, prefix, train, amp)
    x = torch.as_tensor(x) if train else _r(p, torch.as_tensor(x, dtype=torch.float32))
    x = F.relu(_bn(p.sub("bn1"), _conv(p.sub("conv1"), x, 2, 3)))
    x = maxpool3x3s2_same(x)
    for li, nblocks in enumerate(RESNET_LAYERS[name], start=1):
        for b in range(nblocks):
            stride = 2 if (li > 1 and b == 0) else 1
            x = _bottleneck(p.sub(f"layer{li}.{b}"), x, stride)
    return x


def hrnet_head_forward(params, x, prefix="", train=False, amp=False):
    """HRNetHead.construct hrnet_head.py:47-49: 1x1 conv + bias."""
    return _conv(_P(params, prefix, train, amp).sub("head"), x)


def simple_baseline_head_forward(params, x, prefix="", num_deconv_layers=3, amp=False, train=False):
    """SimpleBaselineHead.construct simple_baseline_head.py:95-98.
    deconv_layer = SequentialCell(deconv, bn, relu, deconv, bn, relu, ...) -> indices 3i, 3i+1."""
    p = _P(params, prefix, train, amp)
    for i in range(num_deconv_layers):
        w = p[f"deconv_layer.{3 * i}.weight"]
        x = _r(p, F.conv_transpose2d(x, _r(p, w), None, stride=2, padding=1))
        x = F.relu(_bn(p.sub(f"deconv_layer.{3 * i + 1}"), x))
    return _conv(p.sub("final_layer"), x)


def net_forward_train(params, x, backbone="hrnet_w32", head="hrnet_head", amp=False):
    """Net.construct in training mode: batch-statistics BatchNorm, autograd graph kept.
    ``params`` must hold torch tensors (leaf tensors with requires_grad for the trainable ones).
    ``amp=True``: the reference's training recipe (amp_level O2, tools/train.py:176-181) emulated op by op - fp16 conv operands
    and cell outputs (round-to-nearest fp16 through a differentiable cast, so the activation gradients are rounded to fp16 at
    the same points on the way back), fp32 accumulation, BatchNorm statistics / affine in fp32."""
    if backbone.startswith("hrnet"):
        f = hrnet_forward(params, x, backbone, prefix="backbone.", train=True, amp=amp)
        return hrnet_head_forward(params, f, prefix="head.", train=True, amp=amp)
    f = resnet_forward(params, x, backbone, prefix="backbone.", train=True, amp=amp)
    return simple_baseline_head_forward(params, f, prefix="head.", train=True, amp=amp)


def net_forward(params, x, backbone="hrnet_w32", head="hrnet_head", amp=False):
    """Net.construct networks.py:39-44 (no neck exists in the reference).  ``amp=True``: HRNet + HRNetHead under the
    op-by-op fp16 emulation of amp level O2 (every cell output rounded to fp16, BatchNorm computed in fp32)."""
    with torch.no_grad():
        if amp and backbone.startswith("hrnet"):
            f = hrnet_forward(params, x, backbone, prefix="backbone.", amp=True)
            return hrnet_head_forward(params, f, prefix="head.", amp=True)
        if amp:
            f = resnet_forward(params, x, backbone, prefix="backbone.", amp=True)
            return simple_baseline_head_forward(params, f, prefix="head.", amp=True)
        if backbone.startswith("hrnet"):
            f = hrnet_forward(params, x, backbone, prefix="backbone.")
        else:
            f = resnet_forward(params, x, backbone, prefix="backbone.")
        if head in ("hrnet_head", "HRNetHead"):
            return hrnet_head_forward(params, f, prefix="head.")
        return simple_baseline_head_forward(params, f, prefix="head.")
