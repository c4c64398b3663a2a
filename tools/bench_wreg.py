#!/usr/bin/env python3
"""A/B of the fp16 conv variants on the HRNet branch shapes (interleaved rounds in one process, median of rounds):
   python tools/bench_wreg.py [N]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
from mindpose_amd.models.layers import ActC8
lib = _lib.load(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(32, 32, 64, 48, 3, 1), (64, 64, 32, 24, 3, 1), (128, 128, 16, 12, 3, 1), (256, 256, 8, 6, 3, 1), (96, 96, 48, 36, 3, 1),
          (192, 192, 24, 18, 3, 1), (384, 384, 12, 9, 3, 1), (32, 64, 64, 48, 3, 2), (64, 128, 32, 24, 3, 2), (32, 32, 64, 48, 3, 2),
          (128, 256, 16, 12, 3, 2), (64, 64, 128, 96, 3, 2), (256, 64, 64, 48, 3, 2)]
if len(sys.argv) > 2:
    SHAPES = [sh for sh in SHAPES if sh[5] == int(sys.argv[2])]
for cin, cout, h, w, k, st in SHAPES:
    nn = n if h * w <= 3072 else max(1, n // 2)
    wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    nb = lib.mp_f16_packed_weight_bytes(cout, cin, k, k); packed = torch.empty(nb // 2, device=dev, dtype=torch.float16)
    _lib.check(lib.mp_f16_pack_weight(_lib.ptr(wt), _lib.ptr(packed), cout, cin, k, k, 0, 0, 0, _lib.stream()), "pack")
    cp = (cout + 15) // 16 * 16
    sc, sh = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
    ho, wo = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
    x, out, res = ActC8(nn, cin, h, w, dev), ActC8(nn, cout, ho, wo, dev), ActC8(nn, cout, ho, wo, dev)
    x.c8_tensor.normal_(); res.c8_tensor.normal_()
    d = _lib.ConvDesc(n=nn, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=st, pad_top=k // 2, pad_left=k // 2, conv_h=ho, conv_w=wo, out_h=ho,
                      out_w=wo, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
    def args(v):
        return (ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res), None, _lib.ptr(out), _lib.stream())
    ok = [v for v in range(37) if lib.mp_f16_conv2d_fwd(*args(v)) == 0]
    torch.cuda.synchronize()
    # one native launch plan per variant (40 back-to-back launches enqueued by ONE C call: no Python / ctypes cost per launch)
    plans = {}
    for v in ok:
        h_ = ctypes.c_void_p(lib.mp_plan_create())
        for _ in range(40):
            _lib.check(lib.mp_plan_add_conv_f16(h_, ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res), None,
                                                _lib.ptr(out)), "plan add")
        plans[v] = h_
    times = {v: [] for v in ok}
    for rnd in range(5):
        for v in ok:
            lib.mp_plan_run_range(plans[v], 0, 5, _lib.stream())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.mp_plan_run_range(plans[v], 0, 40, _lib.stream())
            e1.record(); e1.synchronize()
            times[v].append(e0.elapsed_time(e1) / 40 * 1e3)
    for h_ in plans.values():
        lib.mp_plan_destroy(h_)
    med = {v: statistics.median(t) for v, t in times.items()}
    gf = 2 * nn * ho * wo * cout * cin * k * k / 1e9
    old = min((t, v) for v, t in med.items() if v < 25)
    new = min(((t, v) for v, t in med.items() if v >= 25), default=(float("nan"), -1))
    print(f"{cin:3d}->{cout:3d} k{k} s{st} {h}x{w} N={nn}: best tile kernel v{old[1]} {old[0]:6.1f} us ({gf / old[0] * 1e-3:6.1f} TF) | best wreg v{new[1]} {new[0]:6.1f} us "
          f"({gf / new[0] * 1e-3:6.1f} TF) | all: " + " ".join(f"v{v}:{t:.1f}" for v, t in sorted(med.items())), flush=True)
