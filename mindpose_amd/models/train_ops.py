"""Training-mode building blocks: torch.autograd Functions whose forward AND backward are HIP kernels behind the
C ABI (conv forward / data-gradient on the direct MFMA conv, weight-gradient MFMA kernel, BatchNorm2d with batch
statistics, exchange-unit sum).  torch.autograd only records the graph and owns the tensors.

Reference semantics: the cells of mindpose/models/backbones/hrnet.py run by mindspore.Model.train
(tools/train.py:176-233); amp aside, everything here computes in fp32.
"""
import contextlib
import ctypes
import os
from typing import Optional

import numpy as np
import torch

from .. import _lib
from .layers import BN_EPS, F16_VARIANTS, F32_WINOGRAD, tune_conv_variant, winograd_enabled

BN_MOMENTUM = 0.9  # mindspore.nn.BatchNorm2d(momentum=0.9): moving = 0.9*moving + 0.1*batch [MS-knowledge]

_const_cache = {}


def _ones_zeros(c: int, device):
    key = (c, str(device))
    if key not in _const_cache:
        _const_cache[key] = (torch.ones(c, device=device), torch.zeros(c, device=device))
    return _const_cache[key]


def _desc(n, cin, h, w, cout, k, stride, pad_t, pad_l, conv_h, conv_w, out_h, out_w, out_mul=1, out_rep=1, off_y=0,
          off_x=0, flags=0):
    return _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=stride, pad_top=pad_t, pad_left=pad_l,
                         conv_h=conv_h, conv_w=conv_w, out_h=out_h, out_w=out_w, out_mul=out_mul, out_rep=out_rep,
                         out_off_y=off_y, out_off_x=off_x, relu=0, flags=_lib.MP_CONV_SHARES_CUS | flags)


def _conv_launch(lib, d, x, packed, scale, shift, out, what, packed_u=None, res1=None):
    """One forward-kernel launch (forward conv or a data-gradient conv) with the autotuned tile variant; ``packed_u``: the
    Winograd form of the same weights (3x3 stride 1), which then competes in the tuner; ``res1``: added in the epilogue."""
    v = tune_conv_variant(lib, d, x, packed, scale, shift, res1, None, out, packed_u=packed_u)
    if v == F32_WINOGRAD and packed_u is not None:
        _lib.check(lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(x), _lib.ptr(packed_u), _lib.ptr(scale), _lib.ptr(shift),
                                              _lib.ptr(res1), None, _lib.ptr(out), _lib.stream()), what)
        return
    _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(scale), _lib.ptr(shift),
                                         _lib.ptr(res1), None, _lib.ptr(out), _lib.stream()), what)


def _pack_winograd(lib, d, w, cout, cin, mode, owner):
    """Winograd form of a 3x3 weight (mode 5: forward, 6: data gradient) when the descriptor is inside that form, else None."""
    if not winograd_enabled() or os.environ.get("MINDPOSE_AUTOTUNE", "1") == "0" or lib.mp_conv_winograd_supported(ctypes.byref(d)) != 0:
        return None
    return _cached_pack(lib, w, owner, False, cout, cin, 3, mode, 0, 0)


def _pack(lib, w, cout, cin, k, mode, py=0, px=0, owner=None):
    return _cached_pack(lib, w, owner, False, cout, cin, k, mode, py, px)


class Conv2dFn(torch.autograd.Function):
    """z = conv2d(x, weight) (+ bias); k in {1, 3}, stride in {1, 2}, padding = k // 2."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        lib = _lib.load()
        x = _lib.require_cuda_f32(x, "x")
        w = weight.detach().contiguous()
        n, cin, h, wd = x.shape
        cout, _, k, _ = w.shape
        if padding != k // 2 or k not in (1, 3) or stride not in (1, 2):
            raise NotImplementedError("training path covers k in {1,3}, stride in {1,2}, padding = k//2")
        ho, wo = (h + 2 * padding - k) // stride + 1, (wd + 2 * padding - k) // stride + 1
        ones, zeros = _ones_zeros(cout, x.device)
        shift = bias.detach().contiguous() if bias is not None else zeros
        z = torch.empty(n, cout, ho, wo, device=x.device, dtype=torch.float32)
        d = _desc(n, cin, h, wd, cout, k, stride, padding, padding, ho, wo, ho, wo)
        packed = _pack(lib, w, cout, cin, k, 0, owner=weight)
        _conv_launch(lib, d, x, packed, ones, shift, z, "mp_conv2d_fwd", packed_u=_pack_winograd(lib, d, w, cout, cin, 5, weight))
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.padding, ctx.has_bias = stride, padding, bias is not None
        ctx.weight_param = weight
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        n, cin, h, wd = x.shape
        cout, _, k, _ = w.shape
        s, pad = ctx.stride, ctx.padding
        ho, wo = dz.shape[2], dz.shape[3]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            ones, zeros = _ones_zeros(cin, x.device)
            if s == 1:
                dx = torch.empty_like(x)
                d = _desc(n, cout, ho, wo, cin, k, 1, k - 1 - pad, k - 1 - pad, h, wd, h, wd)
                packed = _pack(lib, w, cin, cout, k, 2, owner=ctx.weight_param)
                # ResidualBlockFn: the gradient that reaches x through the identity is added in this launch's epilogue
                _conv_launch(lib, d, dz, packed, ones, zeros, dx, "conv dgrad",
                             packed_u=_pack_winograd(lib, d, w, cin, cout, 6, ctx.weight_param) if k == 3 else None,
                             res1=getattr(ctx, "dx_residual", None))
            else:
                if h != 2 * ho or wd != 2 * wo:
                    raise NotImplementedError("stride-2 data gradient needs even input extents")
                if k == 3:
                    dx = torch.empty_like(x)
                    for py in (0, 1):
                        for px in (0, 1):
                            d = _desc(n, cout, ho, wo, cin, 2, 1, 0, 0, ho, wo, h, wd, out_mul=2, off_y=py, off_x=px)
                            packed = _pack(lib, w, cin, cout, 2, 3, py, px, owner=ctx.weight_param)
                            _conv_launch(lib, d, dz, packed, ones, zeros, dx, "conv dgrad phase")
                else:  # 1x1 stride 2: only even positions receive gradient
                    dx = torch.zeros_like(x)
                    d = _desc(n, cout, ho, wo, cin, 1, 1, 0, 0, ho, wo, h, wd, out_mul=2)
                    packed = _pack(lib, w, cin, cout, 1, 2, owner=ctx.weight_param)
                    _conv_launch(lib, d, dz, packed, ones, zeros, dx, "conv dgrad 1x1s2")
        if ctx.needs_input_grad[1]:
            direct = _direct_grad(ctx.weight_param)  # add straight into the gradient arena: no AccumulateGrad launch
            dw = direct if direct is not None else torch.empty_like(w)
            d = _desc(n, cin, h, wd, cout, k, s, pad, pad, ho, wo, ho, wo)
            ws_bytes = lib.mp_conv_wgrad_workspace_bytes(ctypes.byref(d))
            ws = torch.empty(max(ws_bytes // 4, 1), device=x.device, dtype=torch.float32)
            _lib.check(lib.mp_conv_wgrad(ctypes.byref(d), _lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), int(direct is not None), _lib.ptr(ws),
                                         ws_bytes, _lib.stream()), "mp_conv_wgrad")
            if direct is not None:
                dw = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dz.sum(dim=(0, 2, 3))  # [Cout] reduction of the head conv's bias gradient (17 values)
        return dx, dw, db, None, None


class BatchNormActFn(torch.autograd.Function):
    """y = act(BN_train(z) (+ res)); updates the moving statistics in place (mindspore.nn.BatchNorm2d training)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, res, moving_mean, moving_var, relu):
        lib = _lib.load()
        z = _lib.require_cuda_f32(z, "z")
        n, c, h, w = z.shape
        y = torch.empty_like(z)
        mean = torch.empty(c, device=z.device)
        invstd = torch.empty(c, device=z.device)
        ws_bytes = lib.mp_bn_workspace_bytes(c)
        ws = torch.empty(ws_bytes // 4 + 1, device=z.device, dtype=torch.float32)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        r = res.contiguous() if res is not None else None
        _lib.check(lib.mp_bn_train_fwd(_lib.ptr(z), _lib.ptr(g), _lib.ptr(b), _lib.ptr(r), _lib.ptr(y), _lib.ptr(mean),
                                       _lib.ptr(invstd), _lib.ptr(moving_mean), _lib.ptr(moving_var), n, c, h * w, BN_EPS,
                                       BN_MOMENTUM, int(relu), _lib.ptr(ws), ws_bytes, _lib.stream()), "mp_bn_train_fwd")
        ctx.save_for_backward(z, y, g, mean, invstd)
        ctx.relu, ctx.has_res = relu, res is not None
        ctx.gamma_param, ctx.beta_param = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        z, y, g, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous()
        n, c, h, w = z.shape
        dz = torch.empty_like(z)
        dres = torch.empty_like(z) if ctx.has_res else None
        dgamma = torch.empty(c, device=z.device)
        dbeta = torch.empty(c, device=z.device)
        ws_bytes = lib.mp_bn_workspace_bytes(c)
        ws = torch.empty(ws_bytes // 4 + 1, device=z.device, dtype=torch.float32)
        ga, ba = _direct_grad(ctx.gamma_param), _direct_grad(ctx.beta_param)  # + straight into the gradient arena
        if ga is None or ba is None:
            ga = ba = None
        _lib.check(lib.mp_bn_train_bwd_acc(_lib.ptr(dy), _lib.ptr(z), _lib.ptr(y), _lib.ptr(g), _lib.ptr(mean), _lib.ptr(invstd),
                                           _lib.ptr(dz), _lib.ptr(dres), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ga), _lib.ptr(ba),
                                           n, c, h * w, int(ctx.relu), _lib.ptr(ws), ws_bytes, _lib.stream()), "mp_bn_train_bwd")
        if ga is not None:
            return dz, None, None, dres, None, None, None
        return dz, dgamma, dbeta, dres, None, None, None


class FuseSumFn(torch.autograd.Function):
    """out = relu(((base + up(t1)) + up(t2)) + up(t3)); terms at scale 1 are plain adds (hrnet.py:327-339)."""

    @staticmethod
    def forward(ctx, base, scales, *terms):
        lib = _lib.load()
        n, c, h, w = base.shape
        base = base.contiguous()
        ts = [t.contiguous() for t in terms]
        out = torch.empty_like(base)
        args = []
        for i in range(3):
            if i < len(ts):
                args += [_lib.ptr(ts[i]), int(scales[i])]
            else:
                args += [None, 1]
        _lib.check(lib.mp_fuse_upsample_sum(_lib.ptr(base), *args, _lib.ptr(out), n, c, h, w, 1, _lib.stream()),
                   "mp_fuse_upsample_sum")
        ctx.save_for_backward(out)
        ctx.scales = [int(s) for s in scales[:len(ts)]]
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (out,) = ctx.saved_tensors
        dy = dy.contiguous()
        n, c, h, w = out.shape
        dbase = torch.empty_like(out)
        dts = [torch.empty(n, c, h // s, w // s, device=out.device) for s in ctx.scales]
        args = []
        for i in range(3):
            if i < len(dts):
                args += [_lib.ptr(dts[i]), ctx.scales[i]]
            else:
                args += [None, 1]
        _lib.check(lib.mp_fuse_upsample_sum_bwd(_lib.ptr(dy), _lib.ptr(out), _lib.ptr(dbase), *args, n, c, h, w, 1,
                                                _lib.stream()), "mp_fuse_upsample_sum_bwd")
        return (dbase, None, *dts)


# ---- amp O2: the same graph on channel-blocked fp16 activations ([N, ceil(C/8), H, W, 8] half tensors) -------------------
# fp16 conv operands / activations / activation gradients, fp32 accumulation, BatchNorm statistics and every PARAMETER
# gradient in fp32 (the parameters themselves stay fp32 "master" weights: the pack kernel rounds them to fp16 per step).


def _direct_grad(param):
    """The parameter's slot in a flat gradient arena when the owner allows kernels to accumulate into it directly
    (``GradientAverager`` without overlap hooks sets ``_mp_grad_direct``): saves one AccumulateGrad add launch per parameter
    (878 per HRNet-W32 step, 8 % of the graphed O2 step)."""
    if getattr(param, "_mp_grad_direct", False) and param.grad is not None and param.grad.is_contiguous():
        return param.grad
    return None


def _c8_shape(n, c, h, w):
    return (n, (c + 7) // 8, h, w, 8)


def _c8_alloc(n, c, h, w, device):
    """Uninitialised when every channel block is full; zero-filled when the last block has padding channels (the kernels
    keep them zero, but a fresh buffer must start that way)."""
    make = torch.empty if c % 8 == 0 else torch.zeros
    return make(_c8_shape(n, c, h, w), device=device, dtype=torch.float16)


_BN16_WS = {}


def _bn16_workspace(lib, c, device):
    """ONE persistent BatchNorm workspace per (device, stream): no allocation per layer."""
    need = lib.mp_bn_workspace_bytes(c)
    key = (device, torch.cuda.current_stream(device).cuda_stream)  # calls on one stream are ordered; streams do not share
    ws = _BN16_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.zeros(max(need, lib.mp_bn_workspace_bytes(2048)) // 4 + 1, device=device, dtype=torch.float32)
        _BN16_WS[key] = ws
    return ws, ws.numel() * 4


# ---- fp16 weight packings ------------------------------------------------------------------------------------------------
# The fp32 master weights change every step, so every step re-packs them (forward form + data-gradient forms: 700 packings
# per HRNet-W32 step).  Each parameter remembers its packed buffers (``param._mp_packs``); ``repack_weights(net)`` - called
# by ``Net.train_forward`` - refreshes ALL of them with ONE launch (mp_f16_pack_weight_batch) and the per-layer code then
# finds its packing fresh.  A packing is used once per refresh; anything else (first step, direct calls of the functions,
# a weight changed since the refresh) packs individually, as before.
_PACK_GEN = [0]


def invalidate_packs():
    """Master weights were written behind torch's back (optimizer kernels update through raw pointers): nothing packed
    before this call may be reused."""
    _PACK_GEN[0] += 1


class _PackEntry:
    __slots__ = ("buf", "fresh", "gen", "ptr", "version")

    def __init__(self, buf):
        self.buf, self.fresh, self.gen, self.ptr, self.version = buf, False, -1, 0, -1


class _PackJob(ctypes.Structure):  # mp_f16_pack_job
    _fields_ = [("w", ctypes.c_void_p), ("packed", ctypes.c_void_p)] + [(k, ctypes.c_int) for k in
                ("cout", "cin", "kh", "kw", "transposed", "phase_y", "phase_x", "reserved")]


def _cached_pack(lib, w, owner, half, cout, cin, k, mode, py, px):
    """Packed form of ``w`` (fp16 or fp32 kernels); with ``owner`` (the Parameter ``w`` was detached from) the buffer is
    persistent and a packing refreshed by ``repack_weights`` is handed out without a launch."""
    if half:
        numel, dtype = lib.mp_f16_packed_weight_bytes(cout, cin, k, k) // 2, torch.float16
    elif mode in (5, 6):  # Winograd forms of a 3x3 weight
        numel, dtype = lib.mp_conv_winograd_packed_weight_bytes(cout, cin) // 4, torch.float32
    else:
        numel, dtype = lib.mp_conv_packed_weight_bytes(cout, cin, k, k) // 4, torch.float32
    entry = None
    if owner is not None:
        packs = owner.__dict__.setdefault("_mp_packs", {})
        key = (half, cout, cin, k, mode, py, px)
        entry = packs.get(key)
        if entry is None:
            if half and mode == 3:
                # the four sub-pixel phase packings of one weight are slices (2 py + px) of ONE buffer: MP_CONV_PHASES4 launches
                # take the whole of it, per-phase launches their slice
                parents = owner.__dict__.setdefault("_mp_pack_parents", {})
                parent = parents.get(key[:5])
                if parent is None:
                    parent = parents[key[:5]] = torch.empty(4 * numel, device=w.device, dtype=dtype)
                entry = packs[key] = _PackEntry(parent[(2 * py + px) * numel:(2 * py + px + 1) * numel])
            else:
                entry = packs[key] = _PackEntry(torch.empty(numel, device=w.device, dtype=dtype))
        if (entry.fresh and entry.gen == _PACK_GEN[0] and entry.ptr == w.data_ptr() and entry.version == owner._version):
            entry.fresh = False
            return entry.buf
        entry.fresh = False
    packed = entry.buf if entry is not None else torch.empty(numel, device=w.device, dtype=dtype)
    fn = lib.mp_f16_pack_weight if half else lib.mp_conv_pack_weight
    _lib.check(fn(_lib.ptr(w), _lib.ptr(packed), cout, cin, k, k, mode, py, px, _lib.stream()), "pack_weight")
    return packed


def _pack16(lib, w, cout, cin, k, mode, py=0, px=0, owner=None):
    return _cached_pack(lib, w, owner, True, cout, cin, k, mode, py, px)


def phases4_enabled() -> bool:
    """``MINDPOSE_DGRAD_PHASES4=0``: the stride-2 3x3 data gradient as four phase launches (A/B; the results are bit-identical)."""
    return os.environ.get("MINDPOSE_DGRAD_PHASES4", "1") != "0"


def _dgrad16_stride2(lib, w, owner, dz, dx, n, cin, cout, h, wd, ho, wo, ones, zeros, below=None):
    """Data gradient of a 3x3 stride-2 padding-1 conv: four 2x2 sub-pixel phase convs over dz, each writing every second pixel
    of dx - as ONE launch (MP_CONV_PHASES4; phase = second grid dimension) when the weight has an owner whose four packings share
    a buffer, else as four launches.  ``below`` = (z, y, relu) of the BatchNorm whose output this conv read: the one launch then
    also masks the gradient and leaves that BatchNorm's backward sums (returns (partials, n_parts), else (None, 0))."""
    if owner is not None and phases4_enabled():
        for py in (0, 1):
            for px in (0, 1):
                _pack16(lib, w, cin, cout, 2, 3, py, px, owner=owner)  # fresh slices (no launch after repack_weights)
        parent = owner.__dict__["_mp_pack_parents"][(True, cin, cout, 2, 3)]
        d = _desc(n, cout, ho, wo, cin, 2, 1, 0, 0, ho, wo, h, wd, out_mul=2, flags=_lib.MP_CONV_PHASES4)
        if below is not None:
            return _conv16_stats_launch(lib, d, dz, parent, ones, zeros, dx, None, 2, z=below[0], y=below[1], relu=below[2])
        _conv16_launch(lib, d, dz, parent, ones, zeros, dx, "conv dgrad phases")
        return None, 0
    for py in (0, 1):
        for px in (0, 1):
            d = _desc(n, cout, ho, wo, cin, 2, 1, 0, 0, ho, wo, h, wd, out_mul=2, off_y=py, off_x=px)
            _conv16_launch(lib, d, dz, _pack16(lib, w, cin, cout, 2, 3, py, px, owner=owner), ones, zeros, dx, "conv dgrad phase")
    return None, 0


def repack_weights(module):
    """Refresh every remembered packing of ``module``'s parameters: one launch for the fp16 forms, one for the fp32 forms
    (no-op until a first step has run)."""
    jobs = {True: [], False: []}
    for prm in module.parameters():
        packs = prm.__dict__.get("_mp_packs")
        if packs and prm.is_cuda and prm.dtype == torch.float32 and prm.is_contiguous():
            for key, entry in packs.items():
                jobs[key[0]].append((prm, key, entry))
    total = 0
    for half, group in jobs.items():
        if not group:
            continue
        lib = _lib.load()
        sig = tuple((prm.data_ptr(), entry.buf.data_ptr()) for prm, _, entry in group)
        tables = module.__dict__.setdefault("_mp_pack_tables", {})
        table = tables.get(half)
        if table is None or table[0] != sig:
            arr = (_PackJob * len(group))()
            first = np.zeros(len(group) + 1, dtype=np.uint32)
            for i, (prm, (_, cout, cin, k, mode, py, px), entry) in enumerate(group):
                arr[i] = _PackJob(prm.data_ptr(), entry.buf.data_ptr(), cout, cin, k, k, mode, py, px, 0)
                cp = (cout + 15) // 16 * 16
                # fp16: one thread per (group of 8 input channels, cout) pair - it walks the taps itself
                units = (cin + 31) // 32 * 4 * cp if half else (cin + 3) // 4 * 4 * k * k * (cp // 4)
                if not half and mode in (5, 6):  # Winograd forms: one thread per (cin, cout) pair
                    units = (cin + 3) // 4 * 4 * cp
                first[i + 1] = first[i] + (units + 255) // 256
            dev = group[0][0].device
            jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            first_dev = torch.from_numpy(first.view(np.int32)).to(dev)
            table = tables[half] = (sig, jobs_dev, first_dev, int(first[-1]))
        _, jobs_dev, first_dev, blocks = table
        fn = lib.mp_f16_pack_weight_batch if half else lib.mp_conv_pack_weight_batch
        _lib.check(fn(_lib.ptr(jobs_dev), _lib.ptr(first_dev), len(group), blocks, _lib.stream()), "pack_weight_batch")
        gen = _PACK_GEN[0]
        for prm, _, entry in group:
            entry.fresh, entry.gen, entry.ptr, entry.version = True, gen, prm.data_ptr(), prm._version
        total += len(group)
    return total


def _ones_zeros16(c: int, device):
    return _ones_zeros((c + 15) // 16 * 16, device)


def _conv16_launch(lib, d, x, packed, scale, shift, out, what, res1=None):
    v = tune_conv_variant(lib, d, x, packed, scale, shift, res1, None, out, half=True)  # one-tile / multi-tile, cached per shape
    _lib.check(lib.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(res1),
                                     None, _lib.ptr(out), _lib.stream()), what)


class ToC8Fn(torch.autograd.Function):
    """fp32 NCHW -> channel-blocked fp16 (the network input, or the output of ResNet's fp32 stem: then the gradient flows back
    as fp32 NCHW)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _lib.require_cuda_f32(x, "x")
        n, c, h, w = x.shape
        out = _c8_alloc(n, c, h, w, x.device)
        _lib.check(lib.mp_f16_to_c8(_lib.ptr(x), _lib.ptr(out), n, c, h, w, _lib.stream()), "mp_f16_to_c8")
        ctx.c = c
        return out

    @staticmethod
    def backward(ctx, dy):
        if not ctx.needs_input_grad[0]:
            return None
        lib = _lib.load()
        dy = dy.contiguous()
        n, _, h, w, _ = dy.shape
        dx = torch.empty(n, ctx.c, h, w, device=dy.device, dtype=torch.float32)
        _lib.check(lib.mp_f16_from_c8(_lib.ptr(dy), _lib.ptr(dx), n, ctx.c, h, w, _lib.stream()), "mp_f16_from_c8")
        return dx


class FromC8Fn(torch.autograd.Function):
    """channel-blocked fp16 -> fp32 NCHW (network output handed to the fp32 loss); backward rounds the loss gradient to fp16."""

    @staticmethod
    def forward(ctx, x, c):
        lib = _lib.load()
        n, _, h, w, _ = x.shape
        out = torch.empty(n, c, h, w, device=x.device, dtype=torch.float32)
        _lib.check(lib.mp_f16_from_c8(_lib.ptr(x), _lib.ptr(out), n, c, h, w, _lib.stream()), "mp_f16_from_c8")
        ctx.c = c
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = dy.contiguous()
        n, c, h, w = dy.shape
        out = _c8_alloc(n, c, h, w, dy.device)
        _lib.check(lib.mp_f16_to_c8(_lib.ptr(dy), _lib.ptr(out), n, c, h, w, _lib.stream()), "mp_f16_to_c8")
        return out, None


class Conv16Fn(torch.autograd.Function):
    """z = conv2d(x, fp16(weight)) (+ bias) on the fp16 matrix cores; k in {1, 3}, stride in {1, 2}, padding = k // 2."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        lib = _lib.load()
        w = weight.detach().contiguous()
        n, _, h, wd, _ = x.shape
        cout, cin, k, _ = w.shape
        if padding != k // 2 or k not in (1, 3) or stride not in (1, 2):
            raise NotImplementedError("training path covers k in {1,3}, stride in {1,2}, padding = k//2")
        ho, wo = (h + 2 * padding - k) // stride + 1, (wd + 2 * padding - k) // stride + 1
        ones, zeros = _ones_zeros16(cout, x.device)
        shift = zeros
        if bias is not None:
            shift = torch.zeros_like(zeros)
            shift[:cout] = bias.detach()
        z = _c8_alloc(n, cout, ho, wo, x.device)
        d = _desc(n, cin, h, wd, cout, k, stride, padding, padding, ho, wo, ho, wo)
        packed = _pack16(lib, w, cout, cin, k, 0, owner=weight)
        _conv16_launch(lib, d, x, packed, ones, shift, z, "mp_f16_conv2d_fwd")
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.padding, ctx.has_bias = stride, padding, bias is not None
        ctx.weight_param = weight
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        n, _, h, wd, _ = x.shape
        cout, cin, k, _ = w.shape
        s, pad = ctx.stride, ctx.padding
        ho, wo = dz.shape[2], dz.shape[3]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            ones, zeros = _ones_zeros16(cin, x.device)
            dx = _c8_alloc(n, cin, h, wd, x.device)
            if s == 1:
                d = _desc(n, cout, ho, wo, cin, k, 1, k - 1 - pad, k - 1 - pad, h, wd, h, wd)
                # ResidualBlock16Fn: the gradient that reaches x through the identity is added in this launch's epilogue
                _conv16_launch(lib, d, dz, _pack16(lib, w, cin, cout, k, 2, owner=ctx.weight_param), ones, zeros, dx, "conv dgrad",
                               res1=getattr(ctx, "dx_residual", None))
            else:
                if h != 2 * ho or wd != 2 * wo:
                    raise NotImplementedError("stride-2 data gradient needs even input extents")
                if k == 3:
                    _dgrad16_stride2(lib, w, ctx.weight_param, dz, dx, n, cin, cout, h, wd, ho, wo, ones, zeros)
                else:  # 1x1 stride 2 (ResNet down-sample): only the even positions receive gradient
                    dx.zero_()
                    d = _desc(n, cout, ho, wo, cin, 1, 1, 0, 0, ho, wo, h, wd, out_mul=2)
                    _conv16_launch(lib, d, dz, _pack16(lib, w, cin, cout, 1, 2, owner=ctx.weight_param), ones, zeros, dx, "conv dgrad 1x1s2")
        if ctx.needs_input_grad[1]:
            direct = _direct_grad(ctx.weight_param)  # add straight into the gradient arena: no AccumulateGrad launch
            dw = direct if direct is not None else torch.empty_like(w)
            d = _desc(n, cin, h, wd, cout, k, s, pad, pad, ho, wo, ho, wo)
            ws_bytes = lib.mp_f16_conv_wgrad_workspace_bytes(ctypes.byref(d))
            ws = torch.empty(max(ws_bytes // 4, 1), device=x.device, dtype=torch.float32)
            _lib.check(lib.mp_f16_conv_wgrad(ctypes.byref(d), _lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), 1.0, int(direct is not None),
                                             _lib.ptr(ws), ws_bytes, _lib.stream()), "mp_f16_conv_wgrad")
            if direct is not None:
                dw = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dz.float().sum(dim=(0, 2, 3)).reshape(-1)[:cout]  # head conv bias (17 values)
        return dx, dw, db, None, None


class BatchNormAct16Fn(torch.autograd.Function):
    """y = act(BN_train(z) (+ res)) on channel-blocked fp16; statistics / gamma / beta (and their gradients) in fp32."""

    @staticmethod
    def forward(ctx, z, gamma, beta, res, moving_mean, moving_var, relu):
        lib = _lib.load()
        n, _, h, w, _ = z.shape
        c = gamma.numel()
        y = torch.empty_like(z)
        mean = torch.empty(c, device=z.device)
        invstd = torch.empty(c, device=z.device)
        ws, ws_bytes = _bn16_workspace(lib, c, z.device)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        r = res.contiguous() if res is not None else None
        _lib.check(lib.mp_f16_bn_train_fwd(_lib.ptr(z), _lib.ptr(g), _lib.ptr(b), _lib.ptr(r), _lib.ptr(y), _lib.ptr(mean),
                                           _lib.ptr(invstd), _lib.ptr(moving_mean), _lib.ptr(moving_var), n, c, h * w, BN_EPS,
                                           BN_MOMENTUM, int(relu), _lib.ptr(ws), ws_bytes, _lib.stream()), "mp_f16_bn_train_fwd")
        ctx.save_for_backward(z, y, g, b, mean, invstd)
        ctx.relu, ctx.has_res = relu, res is not None
        ctx.gamma_param, ctx.beta_param = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        z, y, g, b, mean, invstd = ctx.saved_tensors
        if ctx.relu and not ctx.has_res and os.environ.get("MINDPOSE_BN16_MASK_FROM_Z", "1") != "0":
            y = None  # no residual: the ReLU mask is re-derived from z (forward arithmetic), y is not read
        dy = dy.contiguous()
        n, _, h, w, _ = z.shape
        c = g.numel()
        dz = torch.empty_like(z)
        dres = torch.empty_like(z) if ctx.has_res else None
        dgamma = torch.empty(c, device=z.device)
        dbeta = torch.empty(c, device=z.device)
        ws, ws_bytes = _bn16_workspace(lib, c, z.device)
        ga, ba = _direct_grad(ctx.gamma_param), _direct_grad(ctx.beta_param)
        if ga is None or ba is None:
            ga = ba = None
        _lib.check(lib.mp_f16_bn_train_bwd(_lib.ptr(dy), _lib.ptr(z), _lib.ptr(y), _lib.ptr(g), _lib.ptr(b), _lib.ptr(mean),
                                           _lib.ptr(invstd), _lib.ptr(dz), _lib.ptr(dres), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ga),
                                           _lib.ptr(ba), n, c, h * w, int(ctx.relu), _lib.ptr(ws), ws_bytes, _lib.stream()),
                   "mp_f16_bn_train_bwd")
        if ga is not None:
            return dz, None, None, dres, None, None, None
        return dz, dgamma, dbeta, dres, None, None, None


class FuseSum16Fn(torch.autograd.Function):
    """``links``: per operand (base, terms...) the BatchNorm link of the chain that produced it, or None - an operand whose only
    consumer is this sum gets its gradient masked and reduced by the backward kernel itself (``mp_f16_fuse_sum_bwd_term_stats``)."""

    @staticmethod
    def forward(ctx, base, scales, links, *terms):
        lib = _lib.load()
        n, c8, h, w, _ = base.shape
        base = base.contiguous()
        ts = [t.contiguous() for t in terms]
        out = torch.empty_like(base)
        args = []
        for i in range(3):
            args += [_lib.ptr(ts[i]), int(scales[i])] if i < len(ts) else [None, 1]
        _lib.check(lib.mp_f16_fuse_upsample_sum(_lib.ptr(base), *args, _lib.ptr(out), n, c8 * 8, h, w, 1, _lib.stream()),
                   "mp_f16_fuse_upsample_sum")
        ctx.save_for_backward(out)
        ctx.scales = [int(s) for s in scales[:len(ts)]]
        ctx.links = list(links) if links is not None else [None] * (1 + len(ts))
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (out,) = ctx.saved_tensors
        dy = dy.contiguous()
        n, c8, h, w, _ = out.shape
        dbase = torch.empty_like(out)
        dts = [torch.empty(n, c8, h // s, w // s, 8, device=out.device, dtype=torch.float16) for s in ctx.scales]
        outs, scs = [dbase] + dts, [1] + ctx.scales
        plain = [None, None, None, None]
        for k, (dst, sc, link) in enumerate(zip(outs, scs, ctx.links)):
            if (link is not None and link.claimed == 1 and (_bn_fuse_parts() & 8) and tuple(link.z.shape) == tuple(dst.shape)):
                n_parts = lib.mp_f16_fuse_term_stats_parts(n, c8 * 8, h, w, sc)
                part = torch.empty(c8 * n_parts * 16, device=out.device, dtype=torch.float32)
                _lib.check(lib.mp_f16_fuse_sum_bwd_term_stats(_lib.ptr(dy), _lib.ptr(out), _lib.ptr(dst), sc, n, c8 * 8, h, w, 1,
                                                              _lib.ptr(link.z), _lib.ptr(link.y) if link.relu else None, int(link.relu),
                                                              _lib.ptr(part), part.numel() * 4, _lib.stream()),
                           "mp_f16_fuse_sum_bwd_term_stats")
                link.hand_over(part, n_parts, dst)
            else:
                plain[k] = dst
        if any(p is not None for p in plain):
            args = []
            for i in range(3):
                args += [_lib.ptr(plain[i + 1]), ctx.scales[i]] if i < len(dts) else [None, 1]
            _lib.check(lib.mp_f16_fuse_upsample_sum_bwd(_lib.ptr(dy), _lib.ptr(out), _lib.ptr(plain[0]), *args, n, c8 * 8, h, w, 1,
                                                        _lib.stream()), "mp_f16_fuse_upsample_sum_bwd")
        return (dbase, None, None, *dts)


class StemConvFn(torch.autograd.Function):
    """ResNet stem: 7x7 stride-2 conv of the (3-channel, fp32 NCHW) image - forward on the fp32 MFMA kernel, weight gradient by
    the direct-reduction kernel (3 input channels leave the matrix cores nothing to do); the image gets no gradient."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        lib = _lib.load()
        x = _lib.require_cuda_f32(x, "x")
        w = weight.detach().contiguous()
        n, cin, h, wd = x.shape
        cout, _, k, _ = w.shape
        if stride != 2 or padding != k // 2 or cin > 4:
            raise NotImplementedError("stem conv: stride 2, padding k//2, <= 4 input channels")
        ho, wo = (h + 2 * padding - k) // 2 + 1, (wd + 2 * padding - k) // 2 + 1
        ones, zeros = _ones_zeros(cout, x.device)
        z = torch.empty(n, cout, ho, wo, device=x.device, dtype=torch.float32)
        d = _desc(n, cin, h, wd, cout, k, 2, padding, padding, ho, wo, ho, wo)
        _conv_launch(lib, d, x, _pack(lib, w, cout, cin, k, 0), ones, zeros, z, "stem conv")
        ctx.save_for_backward(x, w)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        n, cin, h, wd = x.shape
        cout, _, k, _ = w.shape
        dw = torch.empty_like(w)
        _lib.check(lib.mp_stem_conv_wgrad(_lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), n, cin, h, wd, cout, k, _lib.stream()),
                   "mp_stem_conv_wgrad")
        return None, dw, None, None


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, pad_mode="same") on fp32 NCHW (resnet.py:190)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _lib.require_cuda_f32(x, "x")
        n, c, h, w = x.shape
        out = torch.empty(n, c, (h + 1) // 2, (w + 1) // 2, device=x.device, dtype=torch.float32)
        _lib.check(lib.mp_maxpool3x3s2_same(_lib.ptr(x), _lib.ptr(out), n, c, h, w, _lib.stream()), "mp_maxpool3x3s2_same")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        _lib.check(lib.mp_maxpool3x3s2_same_bwd(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dx), n, c, h, w, _lib.stream()),
                   "mp_maxpool3x3s2_same_bwd")
        return dx


class Deconv16Fn(torch.autograd.Function):
    """Conv2dTranspose(k=4, s=2, p=1) on channel-blocked fp16 (simple_baseline_head.py:80-88), weight [Cin, Cout, 4, 4].
    Forward: four 2x2 sub-pixel phase convs.  Data gradient: per phase, the strided gather of dy followed by the 2x2 conv with
    roles swapped (running sum through the epilogue).  Weight gradient: the 4x4 stride-2 weight-gradient GEMM with the roles of
    input and output gradient exchanged."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = _lib.load()
        w = weight.detach().contiguous()
        n, _, h, wd, _ = x.shape
        cin, cout = w.shape[0], w.shape[1]
        ones, zeros = _ones_zeros16(cout, x.device)
        y = _c8_alloc(n, cout, 2 * h, 2 * wd, x.device)
        for py in (0, 1):
            for px in (0, 1):
                d = _desc(n, cin, h, wd, cout, 2, 1, 1 - py, 1 - px, h, wd, 2 * h, 2 * wd, out_mul=2, off_y=py, off_x=px)
                _conv16_launch(lib, d, x, _pack16(lib, w, cout, cin, 2, 1, py, px, owner=weight), ones, zeros, y, "deconv phase")
        ctx.save_for_backward(x, w)
        ctx.weight_param = weight
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        n, _, h, wd, _ = x.shape
        cin, cout = w.shape[0], w.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            ones, zeros = _ones_zeros16(cin, x.device)
            dx = _c8_alloc(n, cin, h, wd, x.device)
            phase = _c8_alloc(n, cout, h, wd, x.device)
            first = True
            for py in (0, 1):
                for px in (0, 1):
                    _lib.check(lib.mp_f16_gather_phase(_lib.ptr(dy), _lib.ptr(phase), n, cout, h, wd, py, px, _lib.stream()),
                               "mp_f16_gather_phase")
                    # y_phase[m] = sum_t x[m - (1-p) + t] wp[t]  =>  dx[j] = sum_t dy_phase[j + (1-p) - t] wp[t]:
                    # a 2x2 conv over dy_phase with mirrored taps and padding p
                    d = _desc(n, cout, h, wd, cin, 2, 1, py, px, h, wd, h, wd)
                    packed = _pack16(lib, w, cin, cout, 2, 4, py, px, owner=ctx.weight_param)
                    v = tune_conv_variant(lib, d, phase, packed, ones, zeros, None if first else dx, None, dx, half=True)
                    _lib.check(lib.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(phase), _lib.ptr(packed), _lib.ptr(ones),
                                                     _lib.ptr(zeros), None if first else _lib.ptr(dx), None, _lib.ptr(dx),
                                                     _lib.stream()), "deconv dgrad phase")
                    first = False
        if ctx.needs_input_grad[1]:
            # dW[ci][co][ky][kx] = sum_m x[m][ci] * dy[2m + ky - 1][co]: weight gradient of a 4x4 s2 p1 conv whose input is dy
            # and whose output gradient is x -> result laid out [ci][co][4][4] = the transposed conv's own weight layout
            direct = _direct_grad(ctx.weight_param)
            dw = direct if direct is not None else torch.empty_like(w)
            d = _desc(n, cout, 2 * h, 2 * wd, cin, 4, 2, 1, 1, h, wd, h, wd)
            ws_bytes = lib.mp_f16_conv_wgrad_workspace_bytes(ctypes.byref(d))
            if ws_bytes == 0:
                raise _lib.MindposeHipError("transposed-conv weight gradient: unsupported shape")
            ws = torch.empty(ws_bytes // 4, device=x.device, dtype=torch.float32)
            _lib.check(lib.mp_f16_conv_wgrad(ctypes.byref(d), _lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), 1.0, int(direct is not None),
                                             _lib.ptr(ws), ws_bytes, _lib.stream()), "deconv wgrad")
            if direct is not None:
                dw = None
        return dx, dw


def stem_conv_bn_relu(x, conv, bn):
    """ResNet stem group conv7x7 s2 + BatchNorm(train) + ReLU on the fp32 image."""
    z = StemConvFn.apply(x, conv.weight, conv.stride, conv.padding)
    return BatchNormActFn.apply(z, bn.gamma, bn.beta, None, bn.moving_mean, bn.moving_variance, True)


def maxpool3x3s2_same(x):
    return MaxPoolFn.apply(x)


class DeconvFn(torch.autograd.Function):
    """fp32 NCHW form of ``Deconv16Fn``: four 2x2 phase convs forward, gathered-phase 2x2 convs for the data gradient, the 4x4
    stride-2 weight-gradient GEMM (roles exchanged) for the weight gradient."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = _lib.load()
        x = _lib.require_cuda_f32(x, "x")
        w = weight.detach().contiguous()
        n, cin, h, wd = x.shape
        cout = w.shape[1]
        ones, zeros = _ones_zeros(cout, x.device)
        y = torch.empty(n, cout, 2 * h, 2 * wd, device=x.device, dtype=torch.float32)
        for py in (0, 1):
            for px in (0, 1):
                d = _desc(n, cin, h, wd, cout, 2, 1, 1 - py, 1 - px, h, wd, 2 * h, 2 * wd, out_mul=2, off_y=py, off_x=px)
                _conv_launch(lib, d, x, _pack(lib, w, cout, cin, 2, 1, py, px), ones, zeros, y, "deconv phase")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        n, cin, h, wd = x.shape
        cout = w.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            ones, zeros = _ones_zeros(cin, x.device)
            dx = torch.empty_like(x)
            phase = torch.empty(n, cout, h, wd, device=x.device, dtype=torch.float32)
            first = True
            for py in (0, 1):
                for px in (0, 1):
                    _lib.check(lib.mp_gather_phase(_lib.ptr(dy), _lib.ptr(phase), n, cout, h, wd, py, px, _lib.stream()), "mp_gather_phase")
                    d = _desc(n, cout, h, wd, cin, 2, 1, py, px, h, wd, h, wd)
                    packed = _pack(lib, w, cin, cout, 2, 4, py, px)
                    _lib.check(lib.mp_conv2d_fwd(ctypes.byref(d), _lib.ptr(phase), _lib.ptr(packed), _lib.ptr(ones), _lib.ptr(zeros),
                                                 None if first else _lib.ptr(dx), None, _lib.ptr(dx), _lib.stream()), "deconv dgrad phase")
                    first = False
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            d = _desc(n, cout, 2 * h, 2 * wd, cin, 4, 2, 1, 1, h, wd, h, wd)
            ws_bytes = lib.mp_conv_wgrad_workspace_bytes(ctypes.byref(d))
            if ws_bytes == 0:
                raise _lib.MindposeHipError("transposed-conv weight gradient: unsupported shape")
            ws = torch.empty(ws_bytes // 4, device=x.device, dtype=torch.float32)
            _lib.check(lib.mp_conv_wgrad(ctypes.byref(d), _lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), 0, _lib.ptr(ws), ws_bytes,
                                         _lib.stream()), "deconv wgrad")
        return dx, dw


def deconv_bn_relu(x, deconv, bn):
    """Conv2dTranspose(4, 2, 1) + BatchNorm(train) + ReLU of the SimpleBaseline head."""
    _claim(x)
    if _is_c8(x):
        y = Deconv16Fn.apply(x, deconv.weight)
        return BatchNormAct16Fn.apply(y, bn.gamma, bn.beta, None, bn.moving_mean, bn.moving_variance, True)
    y = DeconvFn.apply(x, deconv.weight)
    return BatchNormActFn.apply(y, bn.gamma, bn.beta, None, bn.moving_mean, bn.moving_variance, True)


def to_c8(x: torch.Tensor) -> torch.Tensor:
    return ToC8Fn.apply(x)


def from_c8(x: torch.Tensor, channels: int) -> torch.Tensor:
    _claim(x)
    return FromC8Fn.apply(x, channels)


def _is_c8(x) -> bool:
    return x.dtype == torch.float16 and x.dim() == 5


class _SubCtx:
    """The slice of an autograd context the per-cell Functions use, so that a composite Function can drive their static
    forward / backward methods itself."""

    def __init__(self, n_inputs):
        self.needs_input_grad = (True,) * n_inputs
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class _ResidualBlockFn(torch.autograd.Function):
    """A whole residual block - conv+BN+ReLU groups whose last BatchNorm adds the block input and applies the ReLU (BasicBlock: two
    groups, Bottleneck without down-sample: three) - as ONE autograd node.  Same kernels as the per-cell Functions; what it
    saves is autograd's separate launch that sums the two gradients reaching the block input: the identity's gradient is added
    in the epilogue of the first conv's data-gradient launch (104 launches per HRNet-W32 step).  ``CONV`` / ``BN``: the per-cell
    Functions of the activation type (channel-blocked fp16, or fp32 NCHW)."""
    CONV = BN = None

    @classmethod
    def forward(cls, ctx, x, meta, *params):
        groups = []
        y = x
        n_groups = len(meta)
        for gi, (stride, padding, mm, mv) in enumerate(meta):
            w, gamma, beta = params[3 * gi: 3 * gi + 3]
            cc, bc = _SubCtx(5), _SubCtx(7)
            z = cls.CONV.forward(cc, y, w, None, stride, padding)
            last = gi == n_groups - 1
            y = cls.BN.forward(bc, z, gamma, beta, x if last else None, mm, mv, True)
            groups.append((cc, bc))
        ctx.groups = groups
        return y

    @classmethod
    def backward(cls, ctx, dy):
        grads = []
        g = dy
        dres = None
        for gi in range(len(ctx.groups) - 1, -1, -1):
            cc, bc = ctx.groups[gi]
            dz, dgamma, dbeta, dr, _, _, _ = cls.BN.backward(bc, g)
            if dr is not None:
                dres = dr
            if gi == 0:
                cc.dx_residual = dres
            g, dw, _, _, _ = cls.CONV.backward(cc, dz)
            grads = [dw, dgamma, dbeta] + grads
        ctx.groups = None
        return (g, None, *grads)


class ResidualBlock16Fn(_ResidualBlockFn):
    CONV, BN = Conv16Fn, BatchNormAct16Fn

    @staticmethod
    def forward(ctx, x, meta, *params):
        return _ResidualBlockFn.forward.__func__(ResidualBlock16Fn, ctx, x, meta, *params)

    @staticmethod
    def backward(ctx, dy):
        return _ResidualBlockFn.backward.__func__(ResidualBlock16Fn, ctx, dy)


class ResidualBlock32Fn(_ResidualBlockFn):
    CONV, BN = Conv2dFn, BatchNormActFn

    @staticmethod
    def forward(ctx, x, meta, *params):
        return _ResidualBlockFn.forward.__func__(ResidualBlock32Fn, ctx, x, meta, *params)

    @staticmethod
    def backward(ctx, dy):
        return _ResidualBlockFn.backward.__func__(ResidualBlock32Fn, ctx, dy)


# ---- amp O2: BatchNorm fused into the neighbouring convs (round 3) ----------------------------------------------------------------
# A chain  conv -> BN (-> ReLU) [-> conv -> BN ...] (+ chain input on the last BN)  is ONE autograd node whose conv launches also do
# the BatchNorm REDUCTIONS (csrc/conv_f16_dev.h):
#   forward   conv_i's epilogue leaves the partial sums of z_i, z_i^2        -> BN_i runs its apply pass only
#   backward  the data-gradient launch that produces the gradient reaching BN_i's OUTPUT (conv_{i+1}'s, or - across nodes - the
#             first conv of the NEXT chain, residual gradient included) masks it with y_i > 0 and leaves the partial sums of g, g z_i
#             -> BN_i's backward runs its apply pass only, on the pre-masked gradient (which is also the residual branch's gradient)
# The cross-node hand-over goes through a `_BnLink` the producing chain hangs on its output tensor.  It is used only when the
# consuming chain is the tensor's ONLY consumer: every function of this module that takes a channel-blocked activation counts
# itself on the link (`_claim`), so a second consumer - autograd then SUMS gradients, and the sum is not what the conv's epilogue
# saw - switches the hand-over off for that tensor and the BatchNorm falls back to its own reduction launch.
# MINDPOSE_BN_FUSE=0 keeps the per-cell functions everywhere (A/B, parity tests of the two paths against each other).


def bn_fuse_enabled() -> bool:
    return os.environ.get("MINDPOSE_BN_FUSE", "1") != "0"


def _bn_fuse_parts() -> int:
    """Bit mask of the fused pieces (debugging / A-B): 1 forward statistics, 2 backward statistics inside a chain, 4 across chains,
    8 from the element-wise gradient producers (fan-out sum, exchange-unit backward), 16 from the merged-phase launch of a
    stride-2 data gradient."""
    return int(os.environ.get("MINDPOSE_BN_FUSE_PARTS", "31"))


class _BnLink:
    """The last BatchNorm of a chain, as seen by whoever consumes the chain's output: its input z, its output y (mask), whether it
    has a ReLU.  ``claimed`` counts the consumers of y that know about links; ``partials`` is filled by the consumer's data-gradient
    launch together with ``g``, the pre-masked gradient tensor that launch wrote.  The producer trusts the partial sums only for a
    gradient that IS that tensor (`_link_pre`): a consumer outside this module (a plain torch op, a hook, an auxiliary loss) never
    calls `_claim`, but autograd then hands the producer the SUM of two gradients in a fresh tensor - the check fails and the
    BatchNorm falls back to its own reduce-and-apply pass."""
    __slots__ = ("z", "y", "relu", "claimed", "partials", "n_parts", "g")

    def __init__(self, z, y, relu):
        self.z, self.y, self.relu, self.claimed, self.partials, self.n_parts, self.g = z, y, relu, 0, None, 0, None

    def hand_over(self, part, n_parts, g):
        self.partials, self.n_parts, self.g = part, n_parts, g


def _link_pre(link, dy):
    """(partials, n_parts) when ``dy`` is the pre-masked gradient the link's consumer wrote, else None."""
    if link is None or link.partials is None or link.claimed != 1 or link.g is None:
        return None
    if link.g.data_ptr() != dy.data_ptr() or tuple(link.g.shape) != tuple(dy.shape) or link.g.dtype != dy.dtype:
        return None
    return link.partials, link.n_parts


def _claim(t):
    """A consumer of the activation ``t`` announces itself (see above); returns the link or None."""
    link = getattr(t, "_mp_bn_link", None)
    if link is not None:
        link.claimed += 1
    return link


def _stats_alloc(lib, d, v, c8out, device):
    """Partial-sum buffer of a conv launch with epilogue statistics: (tensor, n_parts) or (None, 0) when variant ``v`` has no
    statistics build for this shape."""
    n_parts = lib.mp_f16_conv_stats_parts(ctypes.byref(d), v)  # v = -1 (tiny layers, tuner off): the library's own choice
    if n_parts <= 0:
        return None, 0
    return torch.empty(c8out * n_parts * 16, device=device, dtype=torch.float32), n_parts


def bn_pre_enabled() -> bool:
    """``MINDPOSE_BN_PRE=1``: inside a chain, conv -> BatchNorm -> ReLU -> 3x3 conv runs WITHOUT an apply pass over the first conv's
    output: the statistics are finalised by a one-block launch and the second conv applies them while it stages its operand,
    writing the activation for the backward pass on the way (bit-identical to the apply pass at launch level,
    tests/test_gpu_bn_fuse.py).  Off by default: measured on HRNet-W32 N = 128 the eager step's kernel time drops 2 % (27.2 ->
    26.6 ms) but the graph-replayed multi-lane step gets 0.8 % SLOWER (5.59 -> 5.55 k img/s, four interleaved pairs on two boxes) -
    the apply passes it removes are HBM-bound launches that were already hidden under the other lanes' convs, the time it adds sits
    in CU-filling conv launches."""
    return os.environ.get("MINDPOSE_BN_PRE", "0") == "1"


_PRE_CAPABLE = {}


def _pre_capable(lib, d) -> bool:
    """Can a conv launch of shape ``d`` apply the BatchNorm of the layer below on its own operand?  (some variant of the
    weights-in-registers family takes it, and the layer is large enough for the tuner to pick variants at all)"""
    key = tuple(getattr(d, f) for f, _ in d._fields_)
    hit = _PRE_CAPABLE.get(key)
    if hit is None:
        macs = d.n * d.conv_h * d.conv_w * d.cout * d.cin * d.kh * d.kw
        hit = (os.environ.get("MINDPOSE_AUTOTUNE", "1") != "0" and macs >= (1 << 26)
               and any(lib.mp_f16_conv_pre_supported(ctypes.byref(d), v) for v in range(F16_VARIANTS)))
        _PRE_CAPABLE[key] = hit
    return hit


def _conv16_stats_launch(lib, d, x, packed, scale, shift, out, res1, mode, z=None, y=None, relu=0, pre=None):
    """The tuned conv launch with epilogue statistics (mode 1 forward / 2 backward); returns (partials, n_parts) or (None, 0) after
    a PLAIN launch when the tuned variant has no statistics build.  ``pre`` = dict(scale, shift, y, relu): ``x`` is the RAW output of
    the conv below and the launch applies that layer's BatchNorm on its operand, writing the activation to ``pre["y"]`` (the caller
    checked `_pre_capable`)."""
    stats = dict(mode=mode, z=z, y=y, relu=relu)
    if pre is not None:
        stats["pre"] = pre
    v = tune_conv_variant(lib, d, x, packed, scale, shift, res1, None, out, half=True, stats=stats)
    part, n_parts = _stats_alloc(lib, d, v, (d.cout + 7) // 8, out.device)
    if pre is not None and (part is None or v < 0):
        raise _lib.MindposeHipError("no conv variant applies the BatchNorm on its operand for a shape _pre_capable admitted")
    if part is None:
        _lib.check(lib.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(scale), _lib.ptr(shift),
                                         _lib.ptr(res1), None, _lib.ptr(out), _lib.stream()), "mp_f16_conv2d_fwd")
        return None, 0
    st = _lib.ConvStats(mode=mode, relu=int(relu), partials=part.data_ptr(), partials_bytes=part.numel() * 4,
                        z=_lib.ptr(z), y=_lib.ptr(y) if relu else None)
    if pre is not None:
        st.pre_scale, st.pre_shift, st.pre_out, st.pre_relu = _lib.ptr(pre["scale"]), _lib.ptr(pre["shift"]), _lib.ptr(pre["y"]), int(pre["relu"])
    _lib.check(lib.mp_f16_conv2d_fwd_stats(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(scale), _lib.ptr(shift),
                                           _lib.ptr(res1), _lib.ptr(out), ctypes.byref(st), _lib.stream()), "mp_f16_conv2d_fwd_stats")
    return part, n_parts


# ---- deferred, grouped weight gradients ------------------------------------------------------------------------------------------------
# A weight gradient is a leaf of the backward pass: nothing reads it before the optimizer.  The fused chains therefore do not launch
# it per layer: the (input, output-gradient, arena slot) triple is queued under its launch shape and stream, and every
# ``WGRAD_GROUP`` layers of one shape - the eight 3x3 convs of an HRNet branch (hrnet.py:202-241) - run as ONE
# ``mp_f16_conv_wgrad_grouped`` launch pair: per layer fewer and longer pixel slabs (an eighth of the split-K traffic), an eighth of
# the launches.  ``flush_wgrad_jobs()`` runs the remainders; whoever reads the gradient arena calls it first (the arena optimizers'
# ``step``, ``GraphedTrainStep`` before its capture ends).  MINDPOSE_WGRAD_GROUP=1 launches every layer on its own as before.
WGRAD_GROUP = 8
_WGRAD_PENDING = {}


def _wgrad_group_size() -> int:
    return max(1, min(WGRAD_GROUP, int(os.environ.get("MINDPOSE_WGRAD_GROUP", str(WGRAD_GROUP)))))


def _wgrad_enqueue(lib, d, x, dz, dw):
    """Queue dW += x (*) dz (accumulating into the arena slot ``dw``); launches the group when it is full."""
    dev = x.device
    # one queue per launch shape, whatever stream a layer ran on: the groups - and with them the split-K partition, i.e. the bits -
    # are the same for a single-stream step and for one whose branches / exchange-unit rows run on side streams
    key = (tuple(getattr(d, f) for f, _ in d._fields_), dev)
    jobs = _WGRAD_PENDING.setdefault(key, [])
    jobs.append((d, x, dz, dw, torch.cuda.current_stream(dev).cuda_stream))
    if len(jobs) >= _wgrad_group_size():
        # (keeping a full group of the branch-0 chain queued for the side stream at the module boundary - instead of 140 us of
        # bandwidth-bound launches at the end of the longest chain - was measured SLOWER: 23.7 vs 22.0 ms per step; the concurrent
        # stream takes HBM from the chain it was meant to shorten)
        _wgrad_flush_key(lib, key)
    elif not _WGRAD_CALLBACK[0]:
        # the remainders are launched when THIS backward pass ends (autograd's final callbacks run after the engine has joined the
        # streams of the pass with the caller's stream): whoever looks at the gradient arena after backward() sees them
        _WGRAD_CALLBACK[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(_wgrad_backward_done)


def _wgrad_flush_key(lib, key, joined: bool = False):
    jobs = _WGRAD_PENDING.pop(key, [])
    if not jobs:
        return
    d, dev = jobs[0][0], key[1]
    n = len(jobs)
    # the group runs on the current stream, behind the producers of every operand (layers that ran on other streams); ``joined``: the
    # caller's stream is already ordered behind all of them (the early flush below forks from such a point)
    cur = torch.cuda.current_stream(dev)
    for sid in sorted({j[4] for j in jobs}):
        if sid != cur.cuda_stream and not joined:
            cur.wait_stream(torch.cuda.ExternalStream(sid, device=dev))
    for _, x, dz, _, sid in jobs:
        if sid != cur.cuda_stream:
            # operands allocated on another stream's pool are read by a launch on THIS stream: tell the caching allocator, or the
            # blocks could be handed out again on their origin stream while the grouped kernel still reads them
            x.record_stream(cur)
            dz.record_stream(cur)
    with torch.cuda.device(dev):
        if n == 1:
            _, x, dz, dw, _ = jobs[0]
            wsb = lib.mp_f16_conv_wgrad_workspace_bytes(ctypes.byref(d))
            ws = torch.empty(max(wsb // 4, 1), device=dev, dtype=torch.float32)
            _lib.check(lib.mp_f16_conv_wgrad(ctypes.byref(d), _lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), 1.0, 1, _lib.ptr(ws), wsb,
                                             _lib.stream()), "mp_f16_conv_wgrad")
            return
        wsb = lib.mp_f16_conv_wgrad_grouped_workspace_bytes(ctypes.byref(d), n)
        ws = torch.empty(max(wsb // 4, 1), device=dev, dtype=torch.float32)
        arr = ctypes.c_void_p * n
        _lib.check(lib.mp_f16_conv_wgrad_grouped(ctypes.byref(d), arr(*[_lib.ptr(j[1]) for j in jobs]), arr(*[_lib.ptr(j[2]) for j in jobs]),
                                                 arr(*[_lib.ptr(j[3]) for j in jobs]), n, 1.0, 1, _lib.ptr(ws), wsb, _lib.stream()),
                   "mp_f16_conv_wgrad_grouped")


_WGRAD_CALLBACK = [False]
_WGRAD_AUTOFLUSH = [True]


def set_wgrad_autoflush(on: bool) -> bool:
    """Whether the end of EVERY backward() call launches the queued remainders (default).  A segmented step (utils/graph_step.py)
    runs one backward pass as several autograd calls and turns this off in between, so that the weight-gradient groups - and with
    them the split-K partitions, i.e. the bits - are those of the single-call pass; it flushes itself after the last segment.
    Returns the previous setting."""
    prev = _WGRAD_AUTOFLUSH[0]
    _WGRAD_AUTOFLUSH[0] = bool(on)
    return prev


def pending_wgrad_slots():
    """data_ptr of every gradient-arena slot that still has a queued (not yet launched) weight gradient."""
    return {j[3].data_ptr() for jobs in _WGRAD_PENDING.values() for j in jobs}


def _wgrad_backward_done() -> None:
    _WGRAD_CALLBACK[0] = False
    if _WGRAD_AUTOFLUSH[0]:
        flush_wgrad_jobs()


def drop_wgrad_jobs(lo: int = 0, hi: int = 1 << 63) -> None:
    """Forget queued weight gradients WITHOUT launching them (a backward pass that raised leaves jobs holding that pass's
    activations; the gradient arena's ``begin_step`` calls this so they can never be accumulated into the next step).  Only jobs
    whose destination slot lies in ``[lo, hi)`` - the caller's own arena - are dropped: a second model / optimizer in the process
    keeps its queue."""
    for key in list(_WGRAD_PENDING):
        keep = [j for j in _WGRAD_PENDING[key] if not (lo <= j[3].data_ptr() < hi)]
        if keep:
            _WGRAD_PENDING[key] = keep
        else:
            del _WGRAD_PENDING[key]
    if not _WGRAD_PENDING:
        _WGRAD_CALLBACK[0] = False


_WGRAD_SIDE = {}          # device -> the stream of the early flush
_WGRAD_SIDE_BUSY = set()  # devices whose early flush has not been joined yet


def flush_wgrad_jobs_early() -> None:
    """Launch what is queued NOW on a side stream forked from the current one, to be joined by the next `flush_wgrad_jobs`.  HRNet
    calls it (a gradient hook on the stage-1 output) when the backward pass reaches stage 1: the remainders of stages 2 - 4 - ~1.5 ms
    of bandwidth-bound launches that used to run alone behind the backward pass - then stream beside the branch-less stage-1 /
    stem backward (also bandwidth-bound, on ONE queue: two such streams together get more of the HBM than either alone).  The jobs
    and their groups are those the end-of-pass flush would have formed (no later layer has the shapes of stages 2 - 4): same bits.
    Only with the automatic end-of-pass flush on (a segmented capture carries its groups across the cuts itself)."""
    if not _WGRAD_PENDING or not _WGRAD_AUTOFLUSH[0] or os.environ.get("MINDPOSE_WGRAD_EARLY_FLUSH", "1") == "0":
        return
    dev = next(iter(_WGRAD_PENDING))[1]
    cur = torch.cuda.current_stream(dev)
    side = _WGRAD_SIDE.get(dev)
    if side is None:
        side = _WGRAD_SIDE[dev] = torch.cuda.Stream(device=dev)
    # every queued job was produced by a node upstream of the caller's gradient: the current stream is ordered behind all of them
    keys = [k for k in _WGRAD_PENDING if k[1] == dev]
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        lib = _lib.load()
        for key in keys:
            _wgrad_flush_key(lib, key, joined=True)
    _WGRAD_SIDE_BUSY.add(dev)


def flush_wgrad_jobs() -> None:
    """Launch every queued weight gradient on the current stream (each group behind the producers of its operands): the arena is
    complete for whatever the caller enqueues next."""
    _WGRAD_CALLBACK[0] = False
    for dev in list(_WGRAD_SIDE_BUSY):  # an early flush of this pass: its launches are part of "the arena is complete"
        torch.cuda.current_stream(dev).wait_stream(_WGRAD_SIDE[dev])
        _WGRAD_SIDE_BUSY.discard(dev)
    if not _WGRAD_PENDING:
        return
    lib = _lib.load()
    for key in list(_WGRAD_PENDING):
        _wgrad_flush_key(lib, key)


def _run_bn_fwd_job(lib, j):
    """The BatchNorm forward pass a chain yielded: apply-only when the statistics came with the conv, reduce + apply otherwise;
    statistics only (``pre``) when the next conv of the chain applies them on its operand."""
    if j.get("pre") is not None:
        _lib.check(lib.mp_f16_bn_train_finalize(_lib.ptr(j["g"]), _lib.ptr(j["b"]), _lib.ptr(j["mean"]), _lib.ptr(j["invstd"]), _lib.ptr(j["mm"]),
                                                _lib.ptr(j["mv"]), j["n"], j["c"], j["hw"], BN_EPS, BN_MOMENTUM, _lib.ptr(j["part"]),
                                                j["n_parts"], _lib.ptr(j["pre"]["scale"]), _lib.ptr(j["pre"]["shift"]), _lib.ptr(j["ws"]),
                                                j["ws_bytes"], _lib.stream()), "mp_f16_bn_train_finalize")
        return
    if j["part"] is None:
        _lib.check(lib.mp_f16_bn_train_fwd(_lib.ptr(j["z"]), _lib.ptr(j["g"]), _lib.ptr(j["b"]), _lib.ptr(j["res"]), _lib.ptr(j["y"]),
                                           _lib.ptr(j["mean"]), _lib.ptr(j["invstd"]), _lib.ptr(j["mm"]), _lib.ptr(j["mv"]), j["n"], j["c"],
                                           j["hw"], BN_EPS, BN_MOMENTUM, j["relu"], _lib.ptr(j["ws"]), j["ws_bytes"], _lib.stream()),
                   "mp_f16_bn_train_fwd")
    else:
        _lib.check(lib.mp_f16_bn_train_fwd_stats(_lib.ptr(j["z"]), _lib.ptr(j["g"]), _lib.ptr(j["b"]), _lib.ptr(j["res"]), _lib.ptr(j["y"]),
                                                 _lib.ptr(j["mean"]), _lib.ptr(j["invstd"]), _lib.ptr(j["mm"]), _lib.ptr(j["mv"]), j["n"],
                                                 j["c"], j["hw"], BN_EPS, BN_MOMENTUM, j["relu"], _lib.ptr(j["part"]), j["n_parts"],
                                                 _lib.ptr(j["ws"]), j["ws_bytes"], _lib.stream()), "mp_f16_bn_train_fwd_stats")


def _run_bn_bwd_job(lib, j):
    """The backward counterpart: apply-only on a pre-masked gradient whose sums its producer delivered, else reduce + apply."""
    if j["pre"] is None:
        _lib.check(lib.mp_f16_bn_train_bwd(_lib.ptr(j["dy"]), _lib.ptr(j["z"]), _lib.ptr(j["yy"]), _lib.ptr(j["g"]), _lib.ptr(j["b"]),
                                           _lib.ptr(j["mean"]), _lib.ptr(j["invstd"]), _lib.ptr(j["dz"]), _lib.ptr(j["dr"]),
                                           _lib.ptr(j["dgamma"]), _lib.ptr(j["dbeta"]), _lib.ptr(j["ga"]), _lib.ptr(j["ba"]), j["n"], j["c"],
                                           j["hw"], j["relu"], _lib.ptr(j["ws"]), j["ws_bytes"], _lib.stream()), "mp_f16_bn_train_bwd")
    else:
        _lib.check(lib.mp_f16_bn_train_bwd_stats(_lib.ptr(j["dy"]), _lib.ptr(j["z"]), _lib.ptr(j["g"]), _lib.ptr(j["mean"]),
                                                 _lib.ptr(j["invstd"]), _lib.ptr(j["dz"]), _lib.ptr(j["dgamma"]), _lib.ptr(j["dbeta"]),
                                                 _lib.ptr(j["ga"]), _lib.ptr(j["ba"]), j["n"], j["c"], j["hw"], _lib.ptr(j["pre"][0]),
                                                 j["pre"][1], _lib.ptr(j["ws"]), j["ws_bytes"], _lib.stream()), "mp_f16_bn_train_bwd_stats")


def _chain16_fwd_steps(lib, x, meta, residual, res_ext, params):
    """Forward of one conv-BatchNorm chain as a generator: launches a group's conv, YIELDS the BatchNorm job (the driver runs it)
    and goes on; returns (groups, output)."""
    groups = []
    a = x
    n_groups = len(meta)
    pre = None  # the previous group's BatchNorm, left to THIS group's conv: dict(scale, shift, y, relu, z)
    for gi, (stride, padding, mm, mv, relu) in enumerate(meta):
        weight, gamma, beta = params[3 * gi: 3 * gi + 3]
        w = weight.detach().contiguous()
        n, _, h, wd, _ = a.shape
        cout, cin, k, _ = w.shape
        if padding != k // 2 or k not in (1, 3) or stride not in (1, 2):
            raise NotImplementedError("training path covers k in {1,3}, stride in {1,2}, padding = k//2")
        ho, wo = (h + 2 * padding - k) // stride + 1, (wd + 2 * padding - k) // stride + 1
        ones, zeros = _ones_zeros16(cout, a.device)
        z = _c8_alloc(n, cout, ho, wo, a.device)
        d = _desc(n, cin, h, wd, cout, k, stride, padding, padding, ho, wo, ho, wo)
        packed = _pack16(lib, w, cout, cin, k, 0, owner=weight)
        if pre is not None:  # operand = the raw output below; this launch applies its BatchNorm and writes the activation `a`
            part, n_parts = _conv16_stats_launch(lib, d, pre["z"], packed, ones, zeros, z, None, 1, pre=pre)
            pre = None
        elif _bn_fuse_parts() & 1:
            part, n_parts = _conv16_stats_launch(lib, d, a, packed, ones, zeros, z, None, 1)
        else:
            part, n_parts = None, 0
            _conv16_launch(lib, d, a, packed, ones, zeros, z, "mp_f16_conv2d_fwd")
        last = gi == n_groups - 1
        res = (x if residual else res_ext) if last else None
        y = torch.empty_like(z)
        mean = torch.empty(cout, device=z.device)
        invstd = torch.empty(cout, device=z.device)
        ws, ws_bytes = _bn16_workspace(lib, cout, z.device)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        job = dict(z=z, g=g, b=b, res=res, y=y, mean=mean, invstd=invstd, mm=mm, mv=mv, n=n, c=cout, hw=ho * wo, relu=int(relu),
                   part=part, n_parts=n_parts, ws=ws, ws_bytes=ws_bytes)
        if not last and part is not None and bn_pre_enabled() and cout % 8 == 0:
            # conv -> BatchNorm -> ReLU -> 3x3 conv (hrnet.py:67-72): when the next conv can take the raw z, only the statistics
            # are finalised here - no pass over z / y; the next launch writes y (the backward pass reads it) on the way
            ns, npad = meta[gi + 1][0], meta[gi + 1][1]
            nw = params[3 * (gi + 1)]
            if nw.shape[2] == 3 and ns == 1 and npad == 1:
                dn = _desc(n, cout, ho, wo, nw.shape[0], 3, 1, 1, 1, ho, wo, ho, wo)
                only = os.environ.get("MINDPOSE_BN_PRE_CH")  # kernel work: the operand route for these channel counts only
                min_hw = int(os.environ.get("MINDPOSE_BN_PRE_MINHW", "0"))  # ... and maps of at least this many pixels
                if _pre_capable(lib, dn) and (not only or str(cout) in only.split(",")) and ho * wo >= min_hw:
                    pre = dict(scale=torch.empty((cout + 7) // 8 * 8, device=z.device), shift=torch.empty((cout + 7) // 8 * 8, device=z.device),
                               y=y, relu=int(relu), z=z)
                    job["pre"] = pre
        yield job
        groups.append(dict(a=a, w=w, weight=weight, gamma=gamma, beta=beta, g=g, b=b, z=z, y=y, mean=mean, invstd=invstd,
                           stride=stride, padding=padding, relu=bool(relu), res=res is not None))
        a = y
    return groups, a


def _chain16_bwd_steps(lib, groups, dy, out_link, in_link, needs_dx, res_is_input):
    """Backward of one chain as a generator: YIELDS each group's BatchNorm backward job, then queues the weight gradient and
    launches the data gradient (which masks / reduces for the BatchNorm below); returns (dx, dres, grads)."""
    dy = dy.contiguous()
    grads = []
    # the gradient reaching the LAST BatchNorm: pre-masked with partial sums when the (only) consumer's data gradient made them
    link = out_link
    pre = _link_pre(link, dy)
    dres = None
    for gi in range(len(groups) - 1, -1, -1):
        G = groups[gi]
        z, y, a, w = G["z"], G["y"], G["a"], G["w"]
        n, _, ho, wo, _ = z.shape
        cout, cin, k, _ = w.shape
        h, wd = a.shape[2], a.shape[3]
        dz = torch.empty_like(z)
        dgamma = torch.empty(cout, device=z.device)
        dbeta = torch.empty(cout, device=z.device)
        ws, ws_bytes = _bn16_workspace(lib, cout, z.device)
        ga, ba = _direct_grad(G["gamma"]), _direct_grad(G["beta"])
        if ga is None or ba is None:
            ga = ba = None
        job = dict(dy=dy, z=z, g=G["g"], b=G["b"], mean=G["mean"], invstd=G["invstd"], dz=dz, dgamma=dgamma, dbeta=dbeta, ga=ga, ba=ba,
                   n=n, c=cout, hw=ho * wo, relu=int(G["relu"]), ws=ws, ws_bytes=ws_bytes, pre=pre, yy=None, dr=None)
        if pre is not None:  # apply pass only: dy is g = dy * mask, its sums came with it
            if G["res"]:
                dres = dy
        else:
            yy = y if (G["relu"] and G["res"]) else None  # no residual: the mask is re-derived from z, y is not read
            if G["relu"] and not G["res"] and os.environ.get("MINDPOSE_BN16_MASK_FROM_Z", "1") == "0":
                yy = y
            job["yy"] = yy
            job["dr"] = torch.empty_like(z) if G["res"] else None
            if G["res"]:
                dres = job["dr"]
        yield job
        if ga is not None:
            dgamma = dbeta = None
        # ---- conv: weight gradient (a leaf), then the data gradient that feeds the BatchNorm below
        s, pad = G["stride"], G["padding"]
        direct = _direct_grad(G["weight"])
        d = _desc(n, cin, h, wd, cout, k, s, pad, pad, ho, wo, ho, wo)
        if direct is not None:  # into the gradient arena: queued, launched with the other layers of this shape
            _wgrad_enqueue(lib, d, a, dz, direct)
            dw = None
        else:
            dw = torch.empty_like(w)
            wsb = lib.mp_f16_conv_wgrad_workspace_bytes(ctypes.byref(d))
            wws = torch.empty(max(wsb // 4, 1), device=z.device, dtype=torch.float32)
            _lib.check(lib.mp_f16_conv_wgrad(ctypes.byref(d), _lib.ptr(a), _lib.ptr(dz), _lib.ptr(dw), 1.0, 0, _lib.ptr(wws), wsb,
                                             _lib.stream()), "mp_f16_conv_wgrad")
        pre = None
        dx = None
        if gi > 0 or needs_dx:
            ones, zeros = _ones_zeros16(cin, z.device)
            dx = _c8_alloc(n, cin, h, wd, z.device)
            res1 = dres if (gi == 0 and res_is_input) else None  # the chain input's second path: added in this launch's epilogue
            # which BatchNorm does this gradient reach?  the previous group's - or, across nodes, the producer of the chain input
            below = None
            if gi > 0:
                if _bn_fuse_parts() & 2:
                    P = groups[gi - 1]
                    below = (P["z"], P["y"], P["relu"], None)
            elif in_link is not None and in_link.claimed == 1 and (_bn_fuse_parts() & 4):
                below = (in_link.z, in_link.y, in_link.relu, in_link)
            if s == 1:
                dd = _desc(n, cout, ho, wo, cin, k, 1, k - 1 - pad, k - 1 - pad, h, wd, h, wd)
                packed = _pack16(lib, w, cin, cout, k, 2, owner=G["weight"])
                if below is not None and tuple(below[0].shape) == tuple(dx.shape):
                    part, n_parts = _conv16_stats_launch(lib, dd, dz, packed, ones, zeros, dx, res1, 2, z=below[0], y=below[1],
                                                         relu=below[2])
                    if part is not None:
                        if below[3] is not None:
                            below[3].hand_over(part, n_parts, dx)
                        else:
                            pre = (part, n_parts)
                else:
                    _conv16_launch(lib, dd, dz, packed, ones, zeros, dx, "conv dgrad", res1=res1)
            else:
                if h != 2 * ho or wd != 2 * wo:
                    raise NotImplementedError("stride-2 data gradient needs even input extents")
                if res1 is not None:
                    raise NotImplementedError("a residual chain starts with a stride-1 conv")
                if k == 3:
                    use = below if (below is not None and tuple(below[0].shape) == tuple(dx.shape) and phases4_enabled()
                                    and (_bn_fuse_parts() & 16)) else None
                    part, n_parts = _dgrad16_stride2(lib, w, G["weight"], dz, dx, n, cin, cout, h, wd, ho, wo, ones, zeros,
                                                     below=use[:3] if use is not None else None)
                    if part is not None:
                        if use[3] is not None:
                            use[3].hand_over(part, n_parts, dx)
                        else:
                            pre = (part, n_parts)
                else:
                    dx.zero_()
                    dd = _desc(n, cout, ho, wo, cin, 1, 1, 0, 0, ho, wo, h, wd, out_mul=2)
                    _conv16_launch(lib, dd, dz, _pack16(lib, w, cin, cout, 1, 2, owner=G["weight"]), ones, zeros, dx, "conv dgrad 1x1s2")
        elif gi == 0 and dres is not None and res_is_input:
            dx = dres
        grads = [dw, dgamma, dbeta] + grads
        dy = dx
    return dy, dres, grads


def _drive(gen, run_job, lib):
    """Run a chain generator to its end, executing the BatchNorm job it yields at every group; returns its return value.
    (The chains are generators so that a driver CAN interleave several of them - the lockstep experiment of DESIGN 4.9.)"""
    while True:
        try:
            job = next(gen)
        except StopIteration as stop:
            return stop.value
        run_job(lib, job)


class Chain16Fn(torch.autograd.Function):
    """``meta`` = per group (stride, padding, moving_mean, moving_var, relu); ``residual``: the chain input is added before the last
    group's activation (BasicBlock / Bottleneck without down-sample, hrnet.py:66-83, 126-146); ``params`` = (weight, gamma, beta)
    per group.  k in {1, 3}, stride in {1, 2}, padding = k // 2, no conv bias."""

    @staticmethod
    def forward(ctx, x, meta, residual, in_link, res_ext, *params):
        """``res_ext``: a second tensor added before the last group's activation instead of the chain input (the block with a
        down-sample path, hrnet.py:74-81: identity = bn(conv1x1(x)))."""
        lib = _lib.load()
        groups, a = _drive(_chain16_fwd_steps(lib, x, meta, residual, res_ext, params), _run_bn_fwd_job, lib)
        ctx.groups = groups
        ctx.in_link = in_link
        ctx.out_link = _BnLink(groups[-1]["z"], a, groups[-1]["relu"])
        ctx.needs_dx = x.requires_grad
        ctx.res_is_input = bool(residual)
        return a

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        groups, ctx.groups = ctx.groups, None
        dx, dres, grads = _drive(_chain16_bwd_steps(lib, groups, dy, ctx.out_link, ctx.in_link, ctx.needs_dx, ctx.res_is_input),
                                 _run_bn_bwd_job, lib)
        return (dx, None, None, None, None if ctx.res_is_input else dres, *grads)


def _chain16(x, groups, relus, residual, res_ext=None):
    """groups = [(conv, bn), ...] on a channel-blocked activation, as ONE fused node."""
    link = _claim(x)  # a residual chain's second use of x (the identity) is inside the node: still ONE consumer
    if res_ext is not None:
        _claim(res_ext)
    meta = tuple((cv.stride, cv.padding, bn.moving_mean, bn.moving_variance, bool(r)) for (cv, bn), r in zip(groups, relus))
    params = [t for cv, bn in groups for t in (cv.weight, bn.gamma, bn.beta)]
    y = Chain16Fn.apply(x, meta, bool(residual), link, res_ext, *params)
    out_link = getattr(y.grad_fn, "out_link", None)  # the node object IS the ctx of forward / backward
    if out_link is not None:
        y._mp_bn_link = out_link
    return y


def residual_block(x, groups):
    """``groups`` = [(conv, bn), ...]; relu(bn_k(conv_k(... relu(bn_1(conv_1 x)) ...)) + x).  Blocks with a stride-1 first conv take the
    one-node form above (both activation types); anything else composes the per-cell functions."""
    first = groups[0][0]
    if _is_c8(x) and bn_fuse_enabled() and first.stride == 1 and all(cv.bias is None for cv, _ in groups):
        return _chain16(x, groups, [True] * len(groups), residual=True)
    _claim(x)
    if first.stride == 1 and all(cv.bias is None for cv, _ in groups) and os.environ.get("MINDPOSE_FUSE_RESIDUAL", "1") != "0":
        meta = tuple((cv.stride, cv.padding, bn.moving_mean, bn.moving_variance) for cv, bn in groups)
        params = [t for cv, bn in groups for t in (cv.weight, bn.gamma, bn.beta)]
        return (ResidualBlock16Fn if _is_c8(x) else ResidualBlock32Fn).apply(x, meta, *params)
    y = x
    for i, (cv, bn) in enumerate(groups):
        y = conv_bn_act(y, cv, bn, relu=True, res=x if i == len(groups) - 1 else None)
    return y


def conv_bn_act(x, conv, bn, relu: bool, res: Optional[torch.Tensor] = None):
    """One conv + BatchNorm(train) (+ residual) (+ ReLU) group of the reference's cells; the kernel family follows the
    activation type (fp32 NCHW, or channel-blocked fp16 under amp O2)."""
    if _is_c8(x):
        if bn is not None and conv.bias is None and bn_fuse_enabled():
            return _chain16(x, [(conv, bn)], [relu], residual=False, res_ext=res)
        _claim(x)
        if res is not None:
            _claim(res)
        z = Conv16Fn.apply(x, conv.weight, conv.bias, conv.stride, conv.padding)
        if bn is None:
            return z
        return BatchNormAct16Fn.apply(z, bn.gamma, bn.beta, res, bn.moving_mean, bn.moving_variance, relu)
    z = Conv2dFn.apply(x, conv.weight, conv.bias, conv.stride, conv.padding)
    if bn is None:
        return z
    return BatchNormActFn.apply(z, bn.gamma, bn.beta, res, bn.moving_mean, bn.moving_variance, relu)


class FanOutFn(torch.autograd.Function):
    """``k`` handles on one tensor for ``k`` consumers; the backward pass sums the ``k`` gradients in ONE launch
    (``mp_sum_tensors``: fp32 arithmetic, one rounding) instead of autograd's ``k - 1`` pairwise add kernels.  ``link``: the tensor
    is the output of a fused chain whose only consumer is this fan-out - the sum kernel then also masks the gradient and leaves the
    last BatchNorm's backward sums (``mp_f16_sum_tensors_stats``)."""

    @staticmethod
    def forward(ctx, x, k, link):
        ctx.set_materialize_grads(False)
        ctx.link = link
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *grads):
        return _fan_in(grads, ctx.link), None, None


def _fan_in(grads, link):
    """The sum of one tensor's consumer gradients (None entries skipped) in one launch - with ``link`` (the tensor is the output of a
    fused chain whose only consumer is the fan-out) also the mask and the last BatchNorm's backward sums."""
    gs = [g.contiguous() for g in grads if g is not None]
    if not gs:
        return None
    if len(gs) == 1:
        return gs[0]
    lib = _lib.load()
    half = gs[0].dtype == torch.float16
    if (half and link is not None and link.claimed == 1 and len(gs) <= 4 and (_bn_fuse_parts() & 8)
            and tuple(link.z.shape) == tuple(gs[0].shape)):
        n, c8, h, w, _ = gs[0].shape
        n_parts = lib.mp_f16_ew_stats_parts(n, c8 * 8, h * w)
        part = torch.empty(c8 * n_parts * 16, device=gs[0].device, dtype=torch.float32)
        dst = torch.empty_like(gs[0])
        ops = gs + [None] * (4 - len(gs))
        _lib.check(lib.mp_f16_sum_tensors_stats(_lib.ptr(ops[0]), _lib.ptr(ops[1]), _lib.ptr(ops[2]), _lib.ptr(ops[3]), _lib.ptr(dst),
                                                _lib.ptr(link.z), _lib.ptr(link.y) if link.relu else None, int(link.relu), n, c8 * 8,
                                                h * w, _lib.ptr(part), part.numel() * 4, _lib.stream()), "mp_f16_sum_tensors_stats")
        link.hand_over(part, n_parts, dst)
        return dst
    out = gs[0]
    for i in range(1, len(gs), 3):  # up to four operands per launch (the running sum + three more)
        ops = gs[i:i + 3]
        dst = torch.empty_like(gs[0])
        _lib.check(lib.mp_sum_tensors(_lib.ptr(out), _lib.ptr(ops[0]), _lib.ptr(ops[1]) if len(ops) > 1 else None,
                                      _lib.ptr(ops[2]) if len(ops) > 2 else None, _lib.ptr(dst), dst.numel() * dst.element_size(),
                                      int(half), _lib.stream()), "mp_sum_tensors")
        out = dst
    return out


class FanOutManyFn(torch.autograd.Function):
    """`FanOutFn` for SEVERAL tensors as ONE autograd node: handles ``ks[j]`` on ``xs[j]``; the backward pass runs the tensors' fan-in
    sums one after the other inside one node.  Why one node: the branch chains of an HRModule's backward each start at "their"
    fan-in; as separate nodes on the current stream the fan-ins form a spine, every chain is released by an event behind a
    DIFFERENT kernel of it - and the hipGraph runtime then replays the chains almost one after the other (tools/probes/
    graph_spine_probe.py: 2217 us against 790 us concurrent / 2839 us serial), while chains forked from ONE point run concurrently
    (846 us).  As one node the engine records its consumers' events behind the LAST fan-in kernel: a single fork point."""

    @staticmethod
    def forward(ctx, ks, links, *xs):
        ctx.set_materialize_grads(False)
        ctx.ks, ctx.links = ks, links
        outs = []
        for x, k in zip(xs, ks):
            outs += [x.view_as(x) for _ in range(k)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        res, i = [], 0
        for k, link in zip(ctx.ks, ctx.links):
            res.append(_fan_in(grads[i:i + k], link))
            i += k
        return (None, None, *res)


class GradJoinFn(torch.autograd.Function):
    """Identity over several tensors as ONE autograd node: in the backward pass their gradients - produced at different points of
    the current stream - leave through one node, so the chains that consume them on side streams fork from a SINGLE point (see
    `FanOutManyFn`: chains released behind different kernels of a spine replay one after the other)."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *grads):
        return grads  # the very tensors that came in: a BatchNorm hand-over (`_link_pre`) still recognises its gradient


def grad_join(xs):
    """``xs`` routed through `GradJoinFn` (the BatchNorm links travel with the tensors); a list of one, CPU tensors or tensors without
    gradient pass through."""
    xs = list(xs)
    if len(xs) < 2 or not all(x.is_cuda and x.requires_grad for x in xs) or os.environ.get("MINDPOSE_FAN_OUT_ONE_NODE", "1") == "0":
        return xs
    outs = GradJoinFn.apply(*xs)
    for o, x in zip(outs, xs):
        link = getattr(x, "_mp_bn_link", None)
        if link is not None:
            o._mp_bn_link = link
    return list(outs)


def fan_out_many(xs, ks):
    """``[fan_out(x, k) for x, k in zip(xs, ks)]`` with the fan-ins of the backward pass in ONE autograd node (`FanOutManyFn`);
    tensors with at most one consumer (or that `fan_out` would pass through) stay outside it."""
    outs = [None] * len(xs)
    pick = []
    for j, (x, k) in enumerate(zip(xs, ks)):
        if k <= 1 or not x.is_cuda or not x.requires_grad or (x.numel() * x.element_size()) % 16 or os.environ.get("MINDPOSE_FAN_OUT", "1") == "0":
            outs[j] = (x,) * k
        else:
            pick.append(j)
    if len(pick) == 1 or os.environ.get("MINDPOSE_FAN_OUT_ONE_NODE", "1") == "0":
        for j in pick:
            outs[j] = fan_out(xs[j], ks[j])
    elif pick:
        flat = FanOutManyFn.apply([ks[j] for j in pick], [_claim(xs[j]) for j in pick], *[xs[j] for j in pick])
        i = 0
        for j in pick:
            outs[j] = tuple(flat[i:i + ks[j]])
            i += ks[j]
    return outs


def fan_out(x, k: int):
    """``k`` handles on ``x`` whose gradients are summed by one kernel; needs 16-byte multiples (every activation here is)."""
    if k <= 1 or not x.is_cuda or not x.requires_grad or (x.numel() * x.element_size()) % 16 or os.environ.get("MINDPOSE_FAN_OUT", "1") == "0":
        return (x,) * k  # the consumers of the k handles count themselves on x's BatchNorm link (the same tensor object)
    return FanOutFn.apply(x, k, _claim(x))


def fuse_sum(base, terms):
    """terms = [(tensor, integer scale), ...] (1-3 entries)."""
    links = [_claim(base)] + [_claim(t) for t, _ in terms]
    if _is_c8(base):
        return FuseSum16Fn.apply(base, [s for _, s in terms], links, *[t for t, _ in terms])
    return FuseSumFn.apply(base, [s for _, s in terms], *[t for t, _ in terms])
