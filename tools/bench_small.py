#!/usr/bin/env python3
"""The small-problem fp32 conv kernel (tuner ids 11 / 12, csrc/conv_small_f32.hip) against Winograd (9) and the direct tile builds
(0 - 7) on HRNet's deep-branch shapes at small batch sizes; device time per launch from a captured hipGraph of 40 launches.
   python tools/bench_small.py [N ...]        (default 1 8 32)"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
batches = [int(a) for a in sys.argv[1:]] or [1, 8, 32]
# (cin, cout, h, w, k, stride): h x w = the INPUT map
SHAPES = [(128, 128, 16, 12, 3, 1), (256, 256, 8, 6, 3, 1), (64, 64, 32, 24, 3, 1), (128, 256, 16, 12, 3, 2), (64, 128, 32, 24, 3, 2),
          (256, 128, 8, 6, 1, 1), (256, 32, 8, 6, 1, 1), (128, 32, 16, 12, 1, 1)]
REPS = 40


def graph_time(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn(side.cuda_stream); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(REPS):
                fn(side.cuda_stream)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / REPS * 1e3)
    return statistics.median(ts)


for n in batches:
    for cin, cout, h, w, k, s in SHAPES:
        pad = k // 2
        oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        x = torch.randn(n, cin, h, w, device=dev); wt = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
        scale = torch.rand(cout, device=dev) + 0.5; shift = torch.randn(cout, device=dev)
        out = torch.empty(n, cout, oh, ow, device=dev)
        d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=s, pad_top=pad, pad_left=pad, conv_h=oh, conv_w=ow, out_h=oh,
                          out_w=ow, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
        pd = torch.empty(lib.mp_conv_packed_weight_bytes(cout, cin, k, k) // 4, device=dev)
        _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wt), _lib.ptr(pd), cout, cin, k, k, 0, 0, 0, _lib.stream()), "pack")
        res = {}
        for v in list(range(8)) + [8, 10, 11, 12]:
            def fn(st, v=v):
                return lib.mp_conv2d_fwd_variant(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(pd), _lib.ptr(scale), _lib.ptr(shift), None, None,
                                                 _lib.ptr(out), ctypes.c_void_p(st))
            if fn(torch.cuda.current_stream().cuda_stream) != 0:
                continue
            torch.cuda.synchronize()
            res[v] = graph_time(lambda st: fn(st))
        if k == 3 and s == 1 and lib.mp_conv_winograd_supported(ctypes.byref(d)) == 0:
            pu = torch.empty(lib.mp_conv_winograd_packed_weight_bytes(cout, cin) // 4, device=dev)
            _lib.check(lib.mp_conv_winograd_pack_weight(_lib.ptr(wt), _lib.ptr(pu), cout, cin, _lib.stream()), "pack u")
            torch.cuda.synchronize()
            res[9] = graph_time(lambda st: lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(x), _lib.ptr(pu), _lib.ptr(scale), _lib.ptr(shift),
                                                                       None, None, _lib.ptr(out), ctypes.c_void_p(st)))
        others = {a: b for a, b in res.items() if a not in (11, 12)}
        bo = min(others, key=others.get)
        gf = 2 * n * oh * ow * cin * cout * k * k / 1e9
        fmt = lambda v: f"{res[v]:6.1f}" if v in res else "   -  "
        print(f"N={n:2d} {cin:3d}->{cout:3d} {h}x{w} k{k}s{s}: small {fmt(11)} us  wide {fmt(12)} us | best other = {bo:2d} {others[bo]:6.1f} us "
              f"| {gf / min(res.values()) * 1e3:5.1f} TF", flush=True)
