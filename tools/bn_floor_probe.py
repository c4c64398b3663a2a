#!/usr/bin/env python3
"""Where does an amp-O2 BatchNorm apply pass spend its time?  Device time per launch (200 launches replayed as one hipGraph)
of mp_f16_bn_train_fwd_stats / _bwd_stats on the HRNet-W32 N = 128 maps as a function of the number of partial-sum slots every
workgroup folds in its prologue, beside a plain streaming kernel (torch copy_) over the same bytes:
   python tools/bn_floor_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MINDPOSE_EXPERIMENT_KNOBS", "1")
from mindpose_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
REPS = 200


def timed(fn):
    """REPS launches captured in one hipGraph (the host's ~9 us per ctypes call would otherwise be the floor)"""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(REPS):
                fn()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


for c, hw in [(32, 48), (256, 48), (128, 192), (64, 768), (32, 3072), (64, 3072)]:
    n = 128
    c8 = (c + 7) // 8
    z = torch.randn(n, c8, hw, 8, device=dev).half()
    g = torch.randn_like(z); y = torch.empty_like(z); res = torch.randn_like(z)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    mean, invstd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    mm, mv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    wsb = lib.mp_bn_workspace_bytes(2048); ws = torch.zeros(wsb // 4 + 1, device=dev)
    mb = z.numel() * 2 / 1e6
    row = [f"C={c:3d} HW={hw:4d} ({mb:5.1f} MB/tensor)"]
    row.append(f"copy {timed(lambda: y.copy_(z)):5.1f}")
    for parts in (1, 64, 256, 512):
        part = torch.rand(c8 * parts * 16, device=dev)
        f = timed(lambda: lib.mp_f16_bn_train_fwd_stats(_lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), None, _lib.ptr(y), _lib.ptr(mean),
                                                        _lib.ptr(invstd), _lib.ptr(mm), _lib.ptr(mv), n, c, hw, 1e-5, 0.9, 1, _lib.ptr(part),
                                                        parts, _lib.ptr(ws), wsb, _lib.stream()))
        fr = timed(lambda: lib.mp_f16_bn_train_fwd_stats(_lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(res), _lib.ptr(y),
                                                         _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(mm), _lib.ptr(mv), n, c, hw, 1e-5, 0.9, 1,
                                                         _lib.ptr(part), parts, _lib.ptr(ws), wsb, _lib.stream()))
        b = timed(lambda: lib.mp_f16_bn_train_bwd_stats(_lib.ptr(g), _lib.ptr(z), _lib.ptr(gamma), _lib.ptr(mean), _lib.ptr(invstd),
                                                        _lib.ptr(y), _lib.ptr(dg), _lib.ptr(db), None, None, n, c, hw, _lib.ptr(part), parts,
                                                        _lib.ptr(ws), wsb, _lib.stream()))
        row.append(f"parts {parts:3d}: fwd {f:5.1f} fwd+res {fr:5.1f} bwd {b:5.1f}")
    print(" | ".join(row), flush=True)
