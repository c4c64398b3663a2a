#!/usr/bin/env python3
"""Time one fp16 conv layer shape across batch sizes and variants (fixed vs proportional cost)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
from mindpose_amd.models.layers import ActC8
lib = _lib.load(); dev = torch.device("cuda:0")
c, h, w = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 16, 12)))
variants = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1, 3, 5, 19]
wt = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
nb = lib.mp_f16_packed_weight_bytes(c, c, 3, 3); packed = torch.empty(nb // 2, device=dev, dtype=torch.float16)
_lib.check(lib.mp_f16_pack_weight(_lib.ptr(wt), _lib.ptr(packed), c, c, 3, 3, 0, 0, 0, _lib.stream()), "pack")
sc, sh = torch.ones((c + 15) // 16 * 16, device=dev), torch.zeros((c + 15) // 16 * 16, device=dev)
for n in (8, 32, 128, 256):
    x, out = ActC8(n, c, h, w, dev), ActC8(n, c, h, w, dev)
    x.c8_tensor.normal_()
    d = _lib.ConvDesc(n=n, cin=c, h=h, w=w, cout=c, kh=3, kw=3, stride=1, pad_top=1, pad_left=1, conv_h=h, conv_w=w, out_h=h, out_w=w,
                      out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
    row = []
    for v in variants:
        args = (ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh), None, None, _lib.ptr(out), _lib.stream())
        if lib.mp_f16_conv2d_fwd(*args) != 0:
            row.append("   -  "); continue
        for _ in range(5): lib.mp_f16_conv2d_fwd(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): lib.mp_f16_conv2d_fwd(*args)
        e1.record(); e1.synchronize()
        row.append(f"{e0.elapsed_time(e1) / 50 * 1e3:6.1f}")
    gf = 2 * n * h * w * c * c * 9 / 1e9
    print(f"C={c} {h}x{w} N={n:4d} ({gf:6.2f} GF) us per variant {variants}: " + " ".join(row))
