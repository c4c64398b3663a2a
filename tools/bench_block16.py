"""Fused fp16 BasicBlock vs two conv launches:  python tools/bench_block16.py [N ...]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from mindpose_amd import _lib  # noqa: E402
from mindpose_amd.models.layers import ActC8  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
c, h, w = 32, 64, 48
for n in [int(a) for a in sys.argv[1:]] or [8, 32, 128, 256]:
    x, mid, out = (ActC8(n, c, h, w, dev) for _ in range(3))
    x.c8_tensor.normal_()
    pk = []
    for _ in range(2):
        wt = torch.randn(c, c, 3, 3, device=dev) / 17
        p = torch.empty(lib.mp_f16_packed_weight_bytes(c, c, 3, 3) // 2, device=dev, dtype=torch.float16)
        _lib.check(lib.mp_f16_pack_weight(_lib.ptr(wt), _lib.ptr(p), c, c, 3, 3, 0, 0, 0, _lib.stream()), "pack")
        pk.append(p)
    sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    d = _lib.ConvDesc(n=n, cin=c, h=h, w=w, cout=c, kh=3, kw=3, stride=1, pad_top=1, pad_left=1, conv_h=h, conv_w=w, out_h=h,
                      out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)

    def two(v):
        lib.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(pk[0]), _lib.ptr(sc), _lib.ptr(sh), None, None, _lib.ptr(mid), _lib.stream())
        lib.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(mid), _lib.ptr(pk[1]), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(x), None, _lib.ptr(out), _lib.stream())

    def fused(rows):
        _lib.check(lib.mp_f16_basicblock_fwd(_lib.ptr(x), _lib.ptr(pk[0]), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(pk[1]), _lib.ptr(sc), _lib.ptr(sh),
                                             _lib.ptr(out), n, c, h, w, rows, _lib.stream()), "fused")

    def timeit(fn, *a):
        for _ in range(5):
            fn(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn(*a)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 50

    import os
    os.environ.setdefault("MINDPOSE_EXPERIMENT_KNOBS", "1")  # the MP_* knobs below are honoured only then
    res = {f"two v{v}": timeit(two, v) for v in (10,)}
    os.environ["MP_F16_BLOCK_V2"] = "0"
    res.update({f"v1 R{r}": timeit(fused, r) for r in (6, 5)})
    os.environ["MP_F16_BLOCK_V2"] = "1"
    res.update({f"v2/8w R{r}": timeit(fused, r) for r in (8, 6)})
    print(f"N={n:4d} us: " + "  ".join(f"{k} {v:6.1f}" for k, v in res.items()), flush=True)
