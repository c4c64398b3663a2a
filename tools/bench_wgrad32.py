#!/usr/bin/env python3
"""fp32 weight-gradient entry (mp_conv_wgrad = split-K MFMA kernel + slab reduction) on the HRNet layer shapes.
   python tools/bench_wgrad32.py [N]      (under rocprofv3 --kernel-trace --stats the two kernels are listed separately)"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
only = sys.argv[2] if len(sys.argv) > 2 else ""
SHAPES = [(32, 32, 64, 48, 3, 1), (64, 64, 32, 24, 3, 1), (128, 128, 16, 12, 3, 1), (256, 256, 8, 6, 3, 1), (64, 256, 64, 48, 1, 1), (32, 64, 64, 48, 3, 2)]
for cin, cout, h, w, k, st in SHAPES:
    if only and only != f"{cin}x{cout}":
        continue
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // st + 1, (w + 2 * pad - k) // st + 1
    x, dz = torch.randn(n, cin, h, w, device=dev), torch.randn(n, cout, ho, wo, device=dev)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=st, pad_top=pad, pad_left=pad, conv_h=ho, conv_w=wo, out_h=ho,
                      out_w=wo, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0)
    nb = lib.mp_conv_wgrad_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nb // 4, device=dev); dw = torch.empty(cout, cin, k, k, device=dev)
    args = (ctypes.byref(d), _lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), 0, _lib.ptr(ws), nb, _lib.stream())
    _lib.check(lib.mp_conv_wgrad(*args), "wgrad")
    ts = []
    for _ in range(5):
        for _ in range(3): lib.mp_conv_wgrad(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.mp_conv_wgrad(*args)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    t = statistics.median(ts)
    gf = 2 * n * ho * wo * cout * cin * k * k / 1e9
    print(f"{cin:3d}->{cout:3d} k{k} s{st} {h}x{w} N={n}: {t:7.1f} us per call ({gf / t * 1e3:5.1f} TFLOP/s), slabs {nb / 1e6:.1f} MB", flush=True)
