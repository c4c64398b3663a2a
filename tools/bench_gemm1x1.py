#!/usr/bin/env python3
"""fp32 convolution: the blocked-GEMM kernel (variant 10) against every other form the tuner knows, on the ResNet-50 / HRNet
pointwise shapes and (round 4, `s2` argument) the 3x3 stride-2 convolutions of HRNet-W32's transitions and exchange units.
   python tools/bench_gemm1x1.py [N] [s2]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(256, 128, 64, 48, 1), (128, 512, 32, 24, 1), (512, 128, 32, 24, 1), (512, 256, 32, 24, 1), (256, 1024, 16, 12, 1), (1024, 256, 16, 12, 1),
          (1024, 512, 16, 12, 1), (512, 2048, 8, 6, 1), (2048, 512, 8, 6, 1), (256, 512, 64, 48, 2), (512, 1024, 32, 24, 2), (1024, 2048, 16, 12, 2),
          (64, 256, 64, 48, 1)]
K = 1
if len(sys.argv) > 2 and sys.argv[2] == "s2":  # (cin, cout, h, w, stride) of the 3x3 stride-2 layers, HRNet-W32 at 256x192
    K = 3
    SHAPES = [(64, 64, 128, 96, 2), (256, 64, 64, 48, 2), (32, 64, 64, 48, 2), (64, 128, 32, 24, 2), (128, 256, 16, 12, 2), (32, 128, 32, 24, 2),
              (64, 256, 16, 12, 2), (32, 256, 16, 12, 2), (64, 64, 32, 24, 2), (32, 32, 64, 48, 2), (32, 32, 32, 24, 2)]
st = _lib.stream()


def timed(fn, reps=20):
    ts = []
    for _ in range(5):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


for cin, cout, h, w, s in SHAPES:
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    x = torch.randn(n, cin, h, w, device=dev); wt = torch.randn(cout, cin, K, K, device=dev) * (2.0 / (cin * K * K)) ** 0.5
    scale = torch.rand(cout, device=dev) + 0.5; shift = torch.randn(cout, device=dev); res = torch.randn(n, cout, ho, wo, device=dev)
    out = torch.empty(n, cout, ho, wo, device=dev)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=K, kw=K, stride=s, pad_top=K // 2, pad_left=K // 2, conv_h=ho, conv_w=wo, out_h=ho, out_w=wo,
                      out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
    pk = torch.empty(lib.mp_conv_packed_weight_bytes(cout, cin, K, K) // 4, device=dev)
    _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wt), _lib.ptr(pk), cout, cin, K, K, 0, 0, 0, st), "pack")
    gf = 2 * n * ho * wo * cin * cout * K * K / 1e9
    best, tg = None, float("nan")
    for v in list(range(9)) + [10]:
        call = lambda: lib.mp_conv2d_fwd_variant(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(pk), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(res), None,
                                                 _lib.ptr(out), st)
        if call() != 0:
            continue
        t = timed(call)
        if v == 10:
            tg = t
        elif best is None or t < best[1]:
            best = (v, t)
    print(f"{cin:4d}->{cout:4d} {h}x{w} s{s} N={n}: gemm {tg:7.1f} us ({gf / tg * 1e3:6.1f} TF) | best other v{best[0]} {best[1]:7.1f} us ({gf / best[1] * 1e3:6.1f} TF) | x{best[1] / tg:.2f}",
          flush=True)
