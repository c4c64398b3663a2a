#!/usr/bin/env python3
"""One fp16 3x3 stride-1 layer, one variant, N launches (the target of a rocprofv3 --pmc pass: tools/pmc_one.sh):
   python tools/ws_one.py <variant> <cin> <cout> <h> <w> <n> [launches]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
from mindpose_amd.models.layers import ActC8
lib = _lib.load(); dev = torch.device("cuda:0")
v, cin, cout, h, w, nn = [int(a) for a in sys.argv[1:7]]
launches = int(sys.argv[7]) if len(sys.argv) > 7 else 20
k = 3
wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
nb = lib.mp_f16_packed_weight_bytes(cout, cin, k, k); packed = torch.empty(nb // 2, device=dev, dtype=torch.float16)
_lib.check(lib.mp_f16_pack_weight(_lib.ptr(wt), _lib.ptr(packed), cout, cin, k, k, 0, 0, 0, _lib.stream()), "pack")
cp = (cout + 15) // 16 * 16
sc, sh = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
x, out, res = ActC8(nn, cin, h, w, dev), ActC8(nn, cout, h, w, dev), ActC8(nn, cout, h, w, dev)
x.c8_tensor.normal_(); res.c8_tensor.normal_()
d = _lib.ConvDesc(n=nn, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=1, pad_top=1, pad_left=1, conv_h=h, conv_w=w, out_h=h,
                  out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
for _ in range(launches):
    _lib.check(lib.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res), None,
                                     _lib.ptr(out), _lib.stream()), "conv")
torch.cuda.synchronize()
print("done")
