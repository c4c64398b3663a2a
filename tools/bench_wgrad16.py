#!/usr/bin/env python3
"""fp16 weight-gradient kernel: LDS-DMA form vs the register-staged form on the HRNet layer shapes.
   python tools/bench_wgrad16.py [N]"""
import ctypes, os, statistics, sys
os.environ.setdefault("MINDPOSE_EXPERIMENT_KNOBS", "1")  # the MP_* knobs below are honoured only then
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
from mindpose_amd.models.layers import ActC8
lib = _lib.load(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(32, 32, 64, 48, 3, 1), (64, 64, 32, 24, 3, 1), (128, 128, 16, 12, 3, 1), (256, 256, 8, 6, 3, 1), (64, 256, 64, 48, 1, 1),
          (256, 64, 64, 48, 1, 1), (64, 64, 64, 48, 3, 1), (32, 64, 64, 48, 3, 2), (64, 128, 32, 24, 3, 2)]
for cin, cout, h, w, k, st in SHAPES:
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // st + 1, (w + 2 * pad - k) // st + 1
    x, dz = ActC8(n, cin, h, w, dev), ActC8(n, cout, ho, wo, dev)
    x.c8_tensor.normal_(); dz.c8_tensor.normal_()
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=st, pad_top=pad, pad_left=pad, conv_h=ho, conv_w=wo, out_h=ho,
                      out_w=wo, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=0, flags=0)
    res, outs = {}, {}
    for mode in ("1", "0"):
        os.environ["MP_WGRAD16_DMA"] = mode
        nb = lib.mp_f16_conv_wgrad_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(nb // 4, device=dev)
        dw = torch.empty(cout, cin, k, k, device=dev)
        args = (ctypes.byref(d), _lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), 1.0, 0, _lib.ptr(ws), nb, _lib.stream())
        _lib.check(lib.mp_f16_conv_wgrad(*args), "wgrad")
        ts = []
        for _ in range(5):
            for _ in range(3): lib.mp_f16_conv_wgrad(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): lib.mp_f16_conv_wgrad(*args)
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[mode] = statistics.median(ts); outs[mode] = dw.clone()
    gf = 2 * n * ho * wo * cout * cin * k * k / 1e9
    err = float((outs["1"] - outs["0"]).abs().max() / outs["0"].abs().max())
    print(f"{cin:3d}->{cout:3d} k{k} s{st} {h}x{w} N={n}: staged {res['0']:6.1f} us ({gf / res['0'] * 1e3:5.0f} TF)  LDS-DMA {res['1']:6.1f} us "
          f"({gf / res['1'] * 1e3:5.0f} TF)  rel diff {err:.1e}", flush=True)
