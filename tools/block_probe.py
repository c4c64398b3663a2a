"""Phase stamps of the fused fp16 BasicBlock kernel, second structure (diagnostic library):
    make -C build/stamps ... EXTRA=-DMP_BLOCK_STAMPS=1  (tools/build_stamps.sh builds with both stamp switches)
    MINDPOSE_HIP_LIB=build/stamps/libmindpose_hip.so python tools/block_probe.py [N]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindpose_amd import _lib
from mindpose_amd.models.layers import ActC8
lib = _lib.load(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
c, h, w = 32, 64, 48
dbg = torch.zeros(256 * 2 * 16, dtype=torch.int64, device=dev)
fn = getattr(lib, "mp_debug_set_block_stamp_buffer")
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(dbg.data_ptr(), dbg.numel() * 8) == 0
x, out = ActC8(n, c, h, w, dev), ActC8(n, c, h, w, dev)
x.c8_tensor.normal_()
pk = []
for _ in range(2):
    wt = torch.randn(c, c, 3, 3, device=dev) / 17
    p = torch.empty(lib.mp_f16_packed_weight_bytes(c, c, 3, 3) // 2, device=dev, dtype=torch.float16)
    _lib.check(lib.mp_f16_pack_weight(_lib.ptr(wt), _lib.ptr(p), c, c, 3, 3, 0, 0, 0, _lib.stream()), "pack"); pk.append(p)
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
for _ in range(20):
    _lib.check(lib.mp_f16_basicblock_fwd(_lib.ptr(x), _lib.ptr(pk[0]), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(pk[1]), _lib.ptr(sc), _lib.ptr(sh),
                                         _lib.ptr(out), n, c, h, w, 0, _lib.stream()), "fused")
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(256, 2, 16).astype(np.float64)
names = ["weights+first band landed", "(bands before the stamped one)", "issue next band's DMA", "conv1 MFMA loop", "epilogue 1 (LDS writes)",
         "barrier 1", "conv2 MFMA loop", "epilogue 2 (stores issued)", "wait for next band", "barrier 2", "(rest of the run)"]
for half in (0, 1):
    d = np.diff(t[:, half, :12], axis=1)
    ok = t[:, half, 11] > 0
    print(f"waves {4 * half}..: total {np.median((t[ok, half, 11] - t[ok, half, 0])):.0f} cycles; median cycles per phase:")
    for i, nm in enumerate(names):
        print(f"   {nm:32s} {np.median(d[ok, i]):8.0f}")
