#!/usr/bin/env python3
"""fp16 training BatchNorm passes (mp_f16_bn_train_fwd / _bwd) on the HRNet-W32 layer shapes: time per call against the
algorithmic bytes (forward: z twice + y [+ res]; backward: dy, z twice + dz, with a residual also y twice + dres).
   python tools/bench_bn16.py [N]"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(32, 64, 48), (64, 32, 24), (128, 16, 12), (256, 8, 6), (64, 64, 48), (256, 64, 48), (32, 32, 24), (32, 16, 12)]
st = _lib.stream()


def timed(fn, reps=40):
    ts = []
    for _ in range(5):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


for c, h, w in SHAPES:
    c8, hw = (c + 7) // 8, h * w
    mk = lambda: torch.randn(n, c8, h, w, 8, device=dev).half()
    z, res, y, dy, dz, dres = mk(), mk(), mk(), mk(), mk(), mk()
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    mean, invstd, mm, mv = (torch.zeros(c, device=dev) for _ in range(4))
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    nb = lib.mp_bn_workspace_bytes(c)
    ws = torch.zeros(nb // 4 + 1, device=dev)
    a = z.numel() * 2
    for relu, with_res in ((1, False), (1, True)):
        r = res if with_res else None
        f = lambda: _lib.check(lib.mp_f16_bn_train_fwd(_lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(r), _lib.ptr(y), _lib.ptr(mean),
                                                       _lib.ptr(invstd), _lib.ptr(mm), _lib.ptr(mv), n, c, hw, 1e-5, 0.9, relu, _lib.ptr(ws), nb, st), "fwd")
        # a layer without residual input re-derives its ReLU mask from z (y not passed), as the training path does
        b = lambda: _lib.check(lib.mp_f16_bn_train_bwd(_lib.ptr(dy), _lib.ptr(z), _lib.ptr(y if with_res else None), _lib.ptr(gamma),
                                                       _lib.ptr(beta), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(dz),
                                                       _lib.ptr(dres if with_res else None), _lib.ptr(dg), _lib.ptr(db), None, None,
                                                       n, c, hw, relu, _lib.ptr(ws), nb, st), "bwd")
        tf, tb = timed(f), timed(b)
        bf, bb = a * (3 + with_res), a * (5 + 3 * with_res)
        print(f"C={c:3d} {h}x{w} N={n} ({a / 1e6:6.1f} MB) relu res={int(with_res)}: fwd {tf:6.1f} us ({bf / tf / 1e6:5.2f} TB/s)   "
              f"bwd {tb:6.1f} us ({bb / tb / 1e6:5.2f} TB/s)", flush=True)
