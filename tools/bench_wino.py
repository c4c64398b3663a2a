#!/usr/bin/env python3
"""fp32 Winograd F(2x2,3x3) kernel against the direct MFMA kernel and torch on the HRNet branch shapes.
   python tools/bench_wino.py [N]"""
import ctypes, os, statistics, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
torch.backends.cudnn.allow_tf32 = False
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(32, 32, 64, 48), (64, 64, 32, 24), (128, 128, 16, 12), (64, 64, 64, 48), (256, 32, 64, 48), (32, 48, 20, 16), (256, 256, 8, 6)]
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


for cin, cout, h, w in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(cin * 1000 + h)
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev)
    scale = (torch.rand(cout, generator=g) + 0.5).to(dev); shift = torch.randn(cout, generator=g).to(dev)
    res = torch.randn(n, cout, h, w, generator=g).to(dev)
    d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=3, kw=3, stride=1, pad_top=1, pad_left=1, conv_h=h, conv_w=w, out_h=h,
                      out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
    rc = lib.mp_conv_winograd_supported(ctypes.byref(d))
    if rc != 0:
        print(f"{cin}->{cout} {h}x{w}: not supported ({rc})"); continue
    pu = torch.empty(lib.mp_conv_winograd_packed_weight_bytes(cout, cin) // 4, device=dev)
    _lib.check(lib.mp_conv_winograd_pack_weight(_lib.ptr(wt), _lib.ptr(pu), cout, cin, st), "pack u")
    pd = torch.empty(lib.mp_conv_packed_weight_bytes(cout, cin, 3, 3) // 4, device=dev)
    _lib.check(lib.mp_conv_pack_weight(_lib.ptr(wt), _lib.ptr(pd), cout, cin, 3, 3, 0, 0, 0, st), "pack d")
    ow, od = torch.full((n, cout, h, w), float("nan"), device=dev), torch.empty(n, cout, h, w, device=dev)
    fw = lambda: _lib.check(lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(x), _lib.ptr(pu), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(res),
                                                       None, _lib.ptr(ow), st), "wino")
    fd = lambda: _lib.check(lib.mp_conv2d_fwd(ctypes.byref(d), _lib.ptr(x), _lib.ptr(pd), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(res), None,
                                              _lib.ptr(od), st), "direct")
    fw(); fd(); torch.cuda.synchronize()
    ref = torch.relu(F.conv2d(x.double(), wt.double(), padding=1) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
                     + res.double())
    ew = float((ow.double() - ref).abs().max()); ed = float((od.double() - ref).abs().max())
    tw, td = timed(fw), timed(fd)
    gf = 2 * n * h * w * cin * cout * 9 / 1e9
    print(f"{cin:3d}->{cout:3d} {h}x{w} N={n}: winograd {tw:7.1f} us ({gf / tw * 1e3:6.1f} TF alg) err {ew:.2e} | direct {td:7.1f} us ({gf / td * 1e3:6.1f} TF) "
          f"err {ed:.2e} | x{td / tw:.2f}", flush=True)
# phase breakdown (diagnostic library: tools/build_stamps.sh, MINDPOSE_HIP_LIB=build/stamps/libmindpose_hip.so)
import numpy as np
dbg = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
if lib.mp_debug_set_stamp_buffer(dbg.data_ptr(), dbg.numel() * 8) == 0:
    for cin, cout, h, w in SHAPES[:4]:
        x = torch.randn(n, cin, h, w, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev)
        scale = torch.ones(cout, device=dev); shift = torch.zeros(cout, device=dev); ow = torch.empty(n, cout, h, w, device=dev)
        d = _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=3, kw=3, stride=1, pad_top=1, pad_left=1, conv_h=h, conv_w=w, out_h=h,
                          out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
        pu = torch.empty(lib.mp_conv_winograd_packed_weight_bytes(cout, cin) // 4, device=dev)
        lib.mp_conv_winograd_pack_weight(_lib.ptr(wt), _lib.ptr(pu), cout, cin, st)
        call = lambda: lib.mp_conv2d_winograd_fwd(ctypes.byref(d), _lib.ptr(x), _lib.ptr(pu), _lib.ptr(scale), _lib.ptr(shift), None, None, _lib.ptr(ow), st)
        call(); call(); torch.cuda.synchronize(); dbg.zero_(); call(); torch.cuda.synchronize()
        wgs = n * ((h + 1) // 2 * (w // 2) + 47) // 48 * ((cout + 31) // 32)  # upper bound (two-team launches have half of it)
        a = dbg[: wgs * 8].reshape(-1, 8).cpu().numpy().astype(np.float64)
        a = a[a[:, 0] > 0]
        names = ["total", "prologue", "transform", "bar(V)", "mfma", "store+wait", "bar(raw)", "epilogue"]
        print(f"{cin}->{cout} {h}x{w}: {len(a)} WGs; ticks/WG: " + "  ".join(f"{nm}={a[:, i].mean():7.0f}" for i, nm in enumerate(names)))
