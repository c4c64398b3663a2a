"""Phase stamps of the weight-stationary fp16 conv kernel (diagnostic library, tools/build_stamps.sh):
    MINDPOSE_HIP_LIB=build/stamps/libmindpose_hip.so python tools/ws_probe.py [variant cin cout h w n]
Prints, for wave 0 of every workgroup (median over workgroups): cycles from kernel start to 'loads issued', to 'prologue done',
then per tile: wait + barrier, DMA issue, MFMA loop, epilogue."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
from mindpose_amd.models.layers import ActC8
lib = _lib.load(); dev = torch.device("cuda:0")
a = [int(v) for v in sys.argv[1:]]
CASES = [tuple(a)] if len(a) == 6 else [(38, 48, 48, 96, 72, 64), (41, 96, 96, 48, 36, 64), (42, 96, 96, 48, 36, 64), (40, 64, 64, 32, 24, 128),
                                        (43, 128, 128, 16, 12, 128), (45, 192, 192, 24, 18, 64), (46, 192, 192, 24, 18, 64)]
fn = getattr(lib, "mp_debug_set_ws_stamp_buffer")
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
dbg = torch.zeros(8192 * 64, dtype=torch.int64, device=dev)
k = 3
for v, cin, cout, h, w, nn in CASES:
    wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    nb = lib.mp_f16_packed_weight_bytes(cout, cin, k, k); packed = torch.empty(nb // 2, device=dev, dtype=torch.float16)
    _lib.check(lib.mp_f16_pack_weight(_lib.ptr(wt), _lib.ptr(packed), cout, cin, k, k, 0, 0, 0, _lib.stream()), "pack")
    cp = (cout + 15) // 16 * 16
    sc, sh = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
    x, out, res = ActC8(nn, cin, h, w, dev), ActC8(nn, cout, h, w, dev), ActC8(nn, cout, h, w, dev)
    x.c8_tensor.normal_(); res.c8_tensor.normal_()
    d = _lib.ConvDesc(n=nn, cin=cin, h=h, w=w, cout=cout, kh=k, kw=k, stride=1, pad_top=1, pad_left=1, conv_h=h, conv_w=w, out_h=h,
                      out_w=w, out_mul=1, out_rep=1, out_off_y=0, out_off_x=0, relu=1, flags=0)
    def run():
        return lib.mp_f16_conv2d_fwd(ctypes.byref(d), v, _lib.ptr(x), _lib.ptr(packed), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res), None,
                                     _lib.ptr(out), _lib.stream())
    assert fn(None, 0) == 0
    if run() != 0:
        print(f"v{v} {cin}->{cout} {h}x{w} N={nn}: not covered"); continue
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    dbg.zero_()
    assert fn(dbg.data_ptr(), dbg.numel() * 8) == 0
    run(); torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(8192, 64).astype(np.float64)
    ok = t[:, 0] > 0
    t = t[ok]
    nblk = len(t)
    start = t[:, 0].min()
    last = np.array([row[row > 0].max() for row in t])
    print(f"v{v} {cin}->{cout} {h}x{w} N={nn}: {us:.1f} us/launch; {nblk} workgroups; s_memtime ticks, kernel span = {(last.max() - start) / us:.0f} ticks/us")
    print(f"   workgroup start spread {np.median(t[:, 0] - start):.0f} (median) {np.max(t[:, 0] - start):.0f} (max); life median {np.median(last - t[:, 0]):.0f}, "
          f"kernel span {last.max() - start:.0f}")
    if v < 37 or v >= 45:  # weights-in-registers kernel: start, loads issued, barrier, one stamp per k-step, end
        ks = [i for i in range(3, 63) if (t[:, i] > 0).all()]
        if (t[:, 60] > 0).all():
            print(f"   prologue: weight loads issued {np.median(t[:, 60] - t[:, 0]):.0f}, DMA issued {np.median(t[:, 61] - t[:, 60]):.0f}, "
                  f"address tables {np.median(t[:, 1] - t[:, 61]):.0f}")
        print(f"   start -> loads issued {np.median(t[:, 1] - t[:, 0]):.0f}; -> all landed + barrier {np.median(t[:, 2] - t[:, 1]):.0f}; k-steps: "
              + " ".join(f"{np.median(t[:, b] - t[:, a]):.0f}" for a, b in zip([2] + ks[:-1], ks)) + f"; epilogue {np.median(t[:, 63] - t[:, ks[-1]]):.0f}")
        continue
    print(f"   start -> loads issued {np.median(t[:, 1] - t[:, 0]):.0f}; -> prologue done {np.median(t[:, 2] - t[:, 1]):.0f}")
    ntile = 0
    while ntile < 6 and (t[:, 4 + 2 * ntile] > 0).all():
        ntile += 1
    for i in range(ntile):
        b = 3 + 2 * i
        print(f"   tile {i}: wait+barrier {np.median(t[:, b] - t[:, b - 1]):.0f}  mfma loop + side jobs + accumulator copy {np.median(t[:, b + 1] - t[:, b]):.0f}")
    print(f"   last stamped tile -> end {np.median(t[:, 15] - t[:, 2 + 2 * ntile]):.0f}")
    taps = [i for i in range(16, 64) if (t[:, i] > 0).all()]
    if len(taps) > 1:
        print("   second tile, cycles per tap: " + " ".join(f"{np.median(t[:, b] - t[:, a]):.0f}" for a, b in zip(taps[:-1], taps[1:])))
