#!/usr/bin/env python3
"""Micro-benchmark of the fp16 BatchNorm training passes at the HRNet-W32 branch shapes (N=128)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for c, h, w in ((32, 64, 48), (64, 32, 24), (128, 16, 12), (256, 8, 6), (64, 64, 48), (256, 64, 48)):
    n = 128
    shape = (n, c // 8, h, w, 8)
    z = torch.randn(shape, device=dev).half(); y = torch.empty_like(z); res = torch.randn(shape, device=dev).half()
    dy = torch.randn(shape, device=dev).half(); dz = torch.empty_like(z); dres = torch.empty_like(z)
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    mean, invstd, dg, db = (torch.empty(c, device=dev) for _ in range(4))
    mm, mv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    nb = lib.mp_bn_workspace_bytes(c); ws = torch.zeros(nb // 4 + 1, device=dev)
    def fwd():
        _lib.check(lib.mp_f16_bn_train_fwd(_lib.ptr(z), _lib.ptr(g), _lib.ptr(b), _lib.ptr(res), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(invstd),
                                           _lib.ptr(mm), _lib.ptr(mv), n, c, h * w, 1e-5, 0.9, 1, _lib.ptr(ws), nb, _lib.stream()), "f")
    def bwd():
        _lib.check(lib.mp_f16_bn_train_bwd(_lib.ptr(dy), _lib.ptr(z), _lib.ptr(y), _lib.ptr(g), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(dz),
                                           _lib.ptr(dres), _lib.ptr(dg), _lib.ptr(db), None, None, n, c, h * w, 1, _lib.ptr(ws), nb, _lib.stream()), "b")
    mb = z.numel() * 2 / 1e6
    for name, fn, passes in (("fwd", fwd, 4), ("bwd", bwd, 8)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"C={c:3d} {h}x{w} tensor {mb:6.1f} MB  {name}: {us:7.1f} us  = {passes * mb / us * 1e-3 * 1e3:6.2f} GB/ms -> {passes * mb / us:5.2f} TB/s over {passes} tensor passes")
