#!/usr/bin/env python3
"""First conv of the network under amp O2: the one-launch kernel reading the fp32 image (stem_f16.hip) against layout pass + general
fp16 conv (best variant); device time from hipGraph replays:   python tools/bench_stem.py [N]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mindpose_amd import _lib
from mindpose_amd.models.layers import ActC8, F16_VARIANTS
lib = _lib.load(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for h, w in [(256, 192), (384, 288)]:
    x = torch.randn(n, 3, h, w, device=dev)
    wt = torch.randn(64, 3, 3, 3, device=dev) / 27 ** 0.5
    sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    xa, out = ActC8(n, 3, h, w, dev), ActC8(n, 64, h // 2, w // 2, dev)
    pk = torch.empty(lib.mp_f16_packed_weight_bytes(64, 3, 3, 3) // 2, device=dev, dtype=torch.float16)
    _lib.check(lib.mp_f16_pack_weight(_lib.ptr(wt), _lib.ptr(pk), 64, 3, 3, 3, 0, 0, 0, _lib.stream()), "pack")
    d = _lib.ConvDesc(n=n, cin=3, h=h, w=w, cout=64, kh=3, kw=3, stride=2, pad_top=1, pad_left=1, conv_h=h // 2, conv_w=w // 2, out_h=h // 2,
                      out_w=w // 2, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
    t_layout = bench.graph_time(lambda: lib.mp_f16_to_c8(_lib.ptr(x), _lib.ptr(xa), n, 3, h, w, _lib.stream()), dev, reps=20, warm=2) * 1e6
    best = {}
    for v in range(F16_VARIANTS):
        call = lambda: lib.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(xa), _lib.ptr(pk), _lib.ptr(sc), _lib.ptr(sh), None, None, _lib.ptr(out), _lib.stream())
        if call() == 0:
            best[v] = bench.graph_time(call, dev, reps=20, warm=2) * 1e6
    v = min(best, key=best.get)
    fused = lambda: lib.mp_f16_stem_conv_fwd(_lib.ptr(x), _lib.ptr(wt), _lib.ptr(sc), _lib.ptr(sh), 1, _lib.ptr(out), n, h, w, _lib.stream())
    _lib.check(fused(), "stem")
    tf = bench.graph_time(fused, dev, reps=20, warm=2) * 1e6
    mb = n * (3 * h * w * 4 + 64 * (h // 2) * (w // 2) * 2) / 1e6
    print(f"{h}x{w} N={n}: layout pass {t_layout:6.1f} us + conv v{v} {best[v]:6.1f} us = {t_layout + best[v]:6.1f} us | one launch {tf:6.1f} us "
          f"({mb:.0f} MB -> {mb / tf:.2f} TB/s)", flush=True)
