#!/usr/bin/env python3
"""fp16 3x3 stride-1 branch convs: the weight-stationary persistent kernel (variants 37..44, conv_f16_ws.hip) against the best of
the other forms, interleaved rounds in one process, median of rounds, 40 back-to-back launches of a native plan per sample:
   python tools/bench_ws.py [w32|w48|all]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
from mindpose_amd.models.layers import ActC8, F16_VARIANTS
lib = _lib.load(); dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
W32 = [(128, 32, 32, 64, 48), (128, 64, 64, 32, 24), (128, 128, 128, 16, 12)]
W48 = [(64, 48, 48, 96, 72), (64, 96, 96, 48, 36), (64, 192, 192, 24, 18), (64, 384, 384, 12, 9), (128, 48, 48, 64, 48), (128, 96, 96, 32, 24),
       (128, 192, 192, 16, 12), (128, 256, 256, 8, 6)]
SHAPES = W32 if which == "w32" else W48 if which == "w48" else W32 + W48
k = 3
for nn, cin, cout, h, w in SHAPES:
    wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    nb = lib.mp_f16_packed_weight_bytes(cout, cin, k, k); packed = torch.empty(nb // 2, device=dev, dtype=torch.float16)
    _lib.check(lib.mp_f16_pack_weight(_lib.ptr(wt), _lib.ptr(packed), cout, cin, k, k, 0, 0, 0, _lib.stream()), "pack")
    cp = (cout + 15) // 16 * 16
    sc, sh = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
    x, out, res = ActC8(nn, cin, h, w, dev), ActC8(nn, cout, h, w, dev), ActC8(nn, cout, h, w, dev)
    x.c8_tensor.normal_(); res.c8_tensor.normal_()
    d = _lib.ConvDesc(n=nn, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=1, pad_top=1, pad_left=1, conv_h=h, conv_w=w, out_h=h,
                      out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
    def args(v):
        return (ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res), None, _lib.ptr(out), _lib.stream())
    ok = [v for v in range(F16_VARIANTS) if lib.mp_f16_conv2d_fwd(*args(v)) == 0]
    torch.cuda.synchronize()
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
    gf = 2 * nn * h * w * cout * cin * k * k / 1e9
    mb = nn * h * w * 2 * ((cin + 7) // 8 * 8 + 2 * ((cout + 7) // 8 * 8)) / 1e6
    old = min((t, v) for v, t in med.items() if v < 37)
    new = min(((t, v) for v, t in med.items() if 37 <= v < 45), default=(float("nan"), -1))
    print(f"{cin:3d}->{cout:3d} {h}x{w} N={nn} ({gf:.2f} GFLOP, {mb:.0f} MB): best other v{old[1]} {old[0]:6.1f} us ({gf / old[0] * 1e3:6.1f} TF) | "
          f"weight-stationary v{new[1]} {new[0]:6.1f} us ({gf / new[0] * 1e3:6.1f} TF, {mb / new[0]:.2f} TB/s) | ws: "
          + " ".join(f"v{v}:{t:.1f}" for v, t in sorted(med.items()) if 37 <= v < 45) + " | wreg round 4: "
          + " ".join(f"v{v}:{t:.1f}" for v, t in sorted(med.items()) if v >= 45), flush=True)
