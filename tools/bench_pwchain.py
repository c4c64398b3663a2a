#!/usr/bin/env python3
"""Stage-1 1x1 chain (expand conv of Bottleneck i + reduce conv of Bottleneck i + 1): the one-launch kernel (pwchain_f16.hip) against
the best variant of each of the two stand-alone launches; device time from hipGraph replays of 20 launches:
   python tools/bench_pwchain.py [N]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mindpose_amd import _lib
from mindpose_amd.models.layers import ActC8, F16_VARIANTS
lib = _lib.load(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for h, w in [(64, 48), (96, 72)]:
    mid, res, y, z = ActC8(n, 64, h, w, dev), ActC8(n, 256, h, w, dev), ActC8(n, 256, h, w, dev), ActC8(n, 64, h, w, dev)
    mid.c8_tensor.normal_(); res.c8_tensor.normal_()
    def pack(cout, cin):
        wt = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
        pk = torch.empty(lib.mp_f16_packed_weight_bytes(cout, cin, 1, 1) // 2, device=dev, dtype=torch.float16)
        _lib.check(lib.mp_f16_pack_weight(_lib.ptr(wt), _lib.ptr(pk), cout, cin, 1, 1, 0, 0, 0, _lib.stream()), "pack")
        return pk
    pk3, pk1 = pack(256, 64), pack(64, 256)
    sc3, sh3, sc1, sh1 = torch.ones(256, device=dev), torch.zeros(256, device=dev), torch.ones(64, device=dev), torch.zeros(64, device=dev)
    def desc(cin, cout):
        return _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=1, kw=1, stride=1, pad_top=0, pad_left=0, conv_h=h, conv_w=w, out_h=h,
                             out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
    d3, d1 = desc(64, 256), desc(256, 64)
    def best(d, x, pk, sc, sh, r, out):
        t = {}
        for v in range(F16_VARIANTS):
            call = lambda: lib.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r) if r is not None else None,
                                                 None, _lib.ptr(out), _lib.stream())
            if call() != 0:
                continue
            t[v] = bench.graph_time(call, dev, reps=20, warm=2) * 1e6
        v = min(t, key=t.get)
        return v, t[v]
    v3, t3 = best(d3, mid, pk3, sc3, sh3, res, y)
    v1, t1 = best(d1, y, pk1, sc1, sh1, None, z)
    fused = lambda: lib.mp_f16_expand_reduce_fwd(_lib.ptr(mid), _lib.ptr(res), _lib.ptr(pk3), _lib.ptr(sc3), _lib.ptr(sh3), 1, _lib.ptr(pk1), _lib.ptr(sc1),
                                                 _lib.ptr(sh1), 1, _lib.ptr(y), _lib.ptr(z), n, 64, 256, 64, h, w, _lib.stream())
    _lib.check(fused(), "fused")
    tf = bench.graph_time(fused, dev, reps=20, warm=2) * 1e6
    mb = n * h * w * 2 * (64 + 256 + 256 + 64) / 1e6
    print(f"{h}x{w} N={n}: expand v{v3} {t3:6.1f} us + reduce v{v1} {t1:6.1f} us = {t3 + t1:6.1f} us | one launch {tf:6.1f} us "
          f"({mb:.0f} MB -> {mb / tf:.2f} TB/s)", flush=True)
