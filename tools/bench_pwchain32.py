#!/usr/bin/env python3
"""The fp32 expand + reduce chain launch (mp_expand_reduce_fwd, csrc/pwchain_f32.hip) against the two launches it replaces, on
HRNet's stage-1 shape: device time per launch from a captured hipGraph of 20.   python tools/bench_pwchain32.py [N] [h w]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 48)
REPS = 20
mid, res = torch.randn(n, 64, h, w, device=dev), torch.randn(n, 256, h, w, device=dev)
w3, w1 = torch.randn(256, 64, 1, 1, device=dev) * 0.17, torch.randn(64, 256, 1, 1, device=dev) * 0.09
s3, b3, s1, b1 = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev), torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
y, z = torch.empty(n, 256, h, w, device=dev), torch.empty(n, 64, h, w, device=dev)
pk3 = torch.empty(lib.mp_conv_packed_weight_bytes(256, 64, 1, 1) // 4, device=dev)
pk1 = torch.empty(lib.mp_conv_packed_weight_bytes(64, 256, 1, 1) // 4, device=dev)
_lib.check(lib.mp_conv_pack_weight(_lib.ptr(w3), _lib.ptr(pk3), 256, 64, 1, 1, 0, 0, 0, _lib.stream()), "pack")
_lib.check(lib.mp_conv_pack_weight(_lib.ptr(w1), _lib.ptr(pk1), 64, 256, 1, 1, 0, 0, 0, _lib.stream()), "pack")


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


def chain(st):
    _lib.check(lib.mp_expand_reduce_fwd(_lib.ptr(mid), _lib.ptr(res), None, None, None, None, _lib.ptr(pk3), _lib.ptr(s3), _lib.ptr(b3), _lib.ptr(pk1),
                                        _lib.ptr(s1), _lib.ptr(b1), _lib.ptr(y), _lib.ptr(z), n, 64, 256, 64, h, w, ctypes.c_void_p(st)), "chain")


def chain_ds(st):  # the first block: down-sample conv of x0 (= mid here) inside the launch
    _lib.check(lib.mp_expand_reduce_fwd(_lib.ptr(mid), None, _lib.ptr(mid), _lib.ptr(pk3), _lib.ptr(s3), _lib.ptr(b3), _lib.ptr(pk3), _lib.ptr(s3),
                                        _lib.ptr(b3), _lib.ptr(pk1), _lib.ptr(s1), _lib.ptr(b1), _lib.ptr(y), _lib.ptr(z), n, 64, 256, 64, h, w,
                                        ctypes.c_void_p(st)), "chain ds")


def expand_only(st):  # the last block
    _lib.check(lib.mp_expand_reduce_fwd(_lib.ptr(mid), _lib.ptr(res), None, None, None, None, _lib.ptr(pk3), _lib.ptr(s3), _lib.ptr(b3), None, None,
                                        None, _lib.ptr(y), None, n, 64, 256, 64, h, w, ctypes.c_void_p(st)), "expand only")


def desc(cin, cout):
    return _lib.ConvDesc(n=n, cin=cin, h=h, w=w, cout=cout, kh=1, kw=1, stride=1, pad_top=0, pad_left=0, conv_h=h, conv_w=w, out_h=h, out_w=w,
                         out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)


d3, d1 = desc(64, 256), desc(256, 64)


def two(st, v3=10, v1=8):
    _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d3), v3, _lib.ptr(mid), _lib.ptr(pk3), _lib.ptr(s3), _lib.ptr(b3), _lib.ptr(res), None, _lib.ptr(y),
                                         ctypes.c_void_p(st)), "expand")
    _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d1), v1, _lib.ptr(y), _lib.ptr(pk1), _lib.ptr(s1), _lib.ptr(b1), None, None, _lib.ptr(z),
                                         ctypes.c_void_p(st)), "reduce")


def three(st):  # what the down-sample form replaces: down-sample conv, expand conv + residual, reduce conv
    _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d3), 10, _lib.ptr(mid), _lib.ptr(pk3), _lib.ptr(s3), _lib.ptr(b3), None, None, _lib.ptr(res),
                                         ctypes.c_void_p(st)), "down-sample")
    two(st)


def expand_gemm(st):
    _lib.check(lib.mp_conv2d_fwd_variant(ctypes.byref(d3), 10, _lib.ptr(mid), _lib.ptr(pk3), _lib.ptr(s3), _lib.ptr(b3), _lib.ptr(res), None, _lib.ptr(y),
                                         ctypes.c_void_p(st)), "expand")


tc = graph_time(chain)
t2 = graph_time(two)
print(f"N={n} {h}x{w}: down-sample form {graph_time(chain_ds):7.1f} us | three launches {graph_time(three):7.1f} us || expand only "
      f"{graph_time(expand_only):7.1f} us | blocked GEMM {graph_time(expand_gemm):7.1f} us", flush=True)
gf = 2 * n * h * w * (64 * 256 + 256 * 64) / 1e9
mb = n * h * w * 4 * (64 + 256 + 256 + 64) / 1e6
print(f"N={n} {h}x{w}: chain {tc:7.1f} us ({gf / tc * 1e3:5.1f} TF, {mb / tc:6.0f} GB/s... {mb:.0f} MB) | two launches (gemm + stream) {t2:7.1f} us", flush=True)
